/*
 * oracle/cloud_oracle.c -- the arithmetic circuits of
 * /root/reference/Cloud/cloud.c restated gate by gate, in the reference's
 * own sequential order, on top of the gate oracle in tfhe_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h).  Each function cites the
 * cloud.c lines it follows.  Arrays are `count` LWE samples of stride n+1.
 */
#include "tfhe_oracle.h"
#include <stdlib.h>
#include <string.h>

#define S(ck) ((size_t)orc_cloudkey_params(ck)->n + 1)

static int32_t *new_array(const orc_cloudkey *ck, size_t count)
{
    return (int32_t *)calloc(count * S(ck), sizeof(int32_t));
}

/* cloud.c:18-51 */
void orc_add(const orc_cloudkey *ck, int32_t *sum, int32_t *carryover, const int32_t *x,
             const int32_t *y, const int32_t *c, int32_t nb_bits)
{
    const size_t s = S(ck);
    int32_t *carry = new_array(ck, 1), *axc = new_array(ck, 1), *bxc = new_array(ck, 1);
    orc_gate_copy(ck, carry, c); /* :24 carry-in is c[0] only */
    for (int32_t i = 0; i < nb_bits; i++) {
        orc_gate_xor(ck, axc, x + i * s, carry);       /* :30 */
        orc_gate_xor(ck, bxc, y + i * s, carry);       /* :32 */
        orc_gate_xor(ck, sum + i * s, x + i * s, bxc); /* :38 */
        orc_gate_and(ck, axc, axc, bxc);               /* :40 */
        orc_gate_xor(ck, carry, carry, axc);           /* :43 */
    }
    orc_gate_copy(ck, carryover, carry); /* :46 carry-out to carryover[0] only */
    orc_scratch_free(ck, carry);
    orc_scratch_free(ck, axc);
    orc_scratch_free(ck, bxc);
}

/* cloud.c:53-57 */
void orc_zero(const orc_cloudkey *ck, int32_t *result, size_t size)
{
    for (size_t i = 0; i < size; i++) orc_gate_constant(ck, result + i * S(ck), 0);
}

/* cloud.c:59-63 */
void orc_NOT(const orc_cloudkey *ck, int32_t *result, const int32_t *x, size_t size)
{
    for (size_t i = 0; i < size; i++) orc_gate_not(ck, result + i * S(ck), x + i * S(ck));
}

static void copy_bits(const orc_cloudkey *ck, int32_t *dst, const int32_t *src, int32_t count)
{
    for (int32_t i = 0; i < count; i++) orc_gate_copy(ck, dst + i * S(ck), src + i * S(ck));
}

/* cloud.c:65-113 */
void orc_split(const orc_cloudkey *ck, int32_t *f1, int32_t *f2, int32_t *f3, const int32_t *a,
               const int32_t *b, const int32_t *c, const int32_t *d, const int32_t *e,
               const int32_t *carry, int32_t nb_bits)
{
    int32_t *sum = new_array(ck, 32), *sum2 = new_array(ck, 32), *sum3 = new_array(ck, 32);
    int32_t *co = new_array(ck, 32), *co2 = new_array(ck, 32), *co3 = new_array(ck, 32);
    orc_zero(ck, sum, nb_bits); /* :77-87 */
    orc_zero(ck, sum2, nb_bits);
    orc_zero(ck, sum3, nb_bits);
    orc_zero(ck, co, nb_bits);
    orc_zero(ck, co2, nb_bits);
    orc_zero(ck, co3, nb_bits);
    orc_add(ck, sum, co, e, b, carry, nb_bits);   /* :90 */
    orc_add(ck, sum2, co2, d, a, co, nb_bits);    /* :91 */
    orc_add(ck, sum3, co3, c, co2, carry, nb_bits); /* :92 y = carry-out word (bit0 + zeros) */
    copy_bits(ck, f1, sum3, nb_bits);             /* :94-105 */
    copy_bits(ck, f2, sum2, nb_bits);
    copy_bits(ck, f3, sum, nb_bits);
    orc_scratch_free(ck, sum);
    orc_scratch_free(ck, sum2);
    orc_scratch_free(ck, sum3);
    orc_scratch_free(ck, co);
    orc_scratch_free(ck, co2);
    orc_scratch_free(ck, co3);
}

/* Shared body of mul32/mul64/mul128 (cloud.c:115-218, 220-385, 387-647):
 * `words` input words (a,b,c,d) times the 32-bit multiplier `m`, accumulated
 * into words+1 sum words.  Per round: `words`x32 ANDs, shifted placement into
 * tmp3c1..tmp3c{words+1}, then words+1 chained in-place adds. */
static void mul_generic(const orc_cloudkey *ck, int32_t **results /*[words+1], high first*/,
                        const int32_t **in /*[words], low first*/, const int32_t *m,
                        const int32_t *carry, int32_t nb_bits, int words)
{
    const size_t s = S(ck);
    const int W1 = words + 1;
    int32_t *sum[5], *tmp[4], *t3c[5], *cy[5];
    for (int w = 0; w < W1; w++) {
        sum[w] = new_array(ck, 32);
        t3c[w] = new_array(ck, 32);
        cy[w] = new_array(ck, 32);
        orc_zero(ck, sum[w], nb_bits); /* :132-145 */
        orc_zero(ck, t3c[w], nb_bits);
        orc_zero(ck, cy[w], nb_bits);
    }
    for (int w = 0; w < words; w++) {
        tmp[w] = new_array(ck, 32);
        orc_zero(ck, tmp[w], nb_bits);
    }
    int round = 0;
    for (int32_t i = 0; i < nb_bits; ++i) {
        for (int32_t k = 0; k < nb_bits; ++k)
            for (int w = 0; w < words; w++)
                orc_gate_and(ck, tmp[w] + k * s, in[w] + k * s, m + i * s); /* :159, :275-279, :461-470 */
        int counter1 = 32 - round, counter2 = 32 - counter1;
        for (int32_t q = 0; q < round; ++q) orc_gate_constant(ck, t3c[0] + q * s, 0); /* :164-169 */
        /* tmp -> tmp3c1 shifted up by `round` (:171-181) */
        for (int32_t q = 0; q < counter1; ++q)
            orc_gate_copy(ck, t3c[0] + (q + round) * s, tmp[0] + q * s);
        for (int w = 0; w < words; w++) {
            /* high `round` bits of tmp[w] -> low bits of tmp3c{w+2} (:183-192, :316-326) */
            for (int32_t q = 0; q < counter2; ++q)
                orc_gate_copy(ck, t3c[w + 1] + q * s, tmp[w] + (q + counter1) * s);
            /* low bits of tmp[w+1] -> upper part of tmp3c{w+2} (:329-339) */
            if (w + 1 < words)
                for (int32_t q = 0; q < counter1; ++q)
                    orc_gate_copy(ck, t3c[w + 1] + (q + counter2) * s, tmp[w + 1] + q * s);
        }
        /* :194-195, :355-357, :604-608 */
        orc_add(ck, sum[0], cy[0], sum[0], t3c[0], carry, 32);
        for (int w = 1; w < W1; w++) orc_add(ck, sum[w], cy[w], sum[w], t3c[w], cy[w - 1], 32);
        round++;
    }
    for (int w = 0; w < W1; w++) copy_bits(ck, results[w], sum[W1 - 1 - w], 32); /* :200-204 */
    for (int w = 0; w < W1; w++) {
        orc_scratch_free(ck, sum[w]);
        orc_scratch_free(ck, t3c[w]);
        orc_scratch_free(ck, cy[w]);
    }
    for (int w = 0; w < words; w++) orc_scratch_free(ck, tmp[w]);
}

/* cloud.c:115-218 : result = high word, result2 = low word */
void orc_mul32(const orc_cloudkey *ck, int32_t *result, int32_t *result2, const int32_t *a,
               const int32_t *b, const int32_t *carry, int32_t nb_bits)
{
    int32_t *res[2] = {result, result2};
    const int32_t *in[1] = {a};
    mul_generic(ck, res, in, b, carry, nb_bits, 1);
}
/* cloud.c:220-385 : (b:a) x c -> result(high), result2, result3(low) */
void orc_mul64(const orc_cloudkey *ck, int32_t *result, int32_t *result2, int32_t *result3,
               const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *carry,
               int32_t nb_bits)
{
    int32_t *res[3] = {result, result2, result3};
    const int32_t *in[2] = {a, b};
    mul_generic(ck, res, in, c, carry, nb_bits, 2);
}
/* cloud.c:387-647 */
void orc_mul128(const orc_cloudkey *ck, int32_t *r1, int32_t *r2, int32_t *r3, int32_t *r4,
                int32_t *r5, const int32_t *a, const int32_t *b, const int32_t *c,
                const int32_t *d, const int32_t *e, const int32_t *carry, int32_t nb_bits)
{
    int32_t *res[5] = {r1, r2, r3, r4, r5};
    const int32_t *in[4] = {a, b, c, d};
    mul_generic(ck, res, in, e, carry, nb_bits, 4);
}

/* cloud.c:775-864, metadata arithmetic in the clear */
int orc_cloud_metadata(int32_t op, int32_t neg1, int32_t bit1, int32_t neg2, int32_t bit2,
                       int32_t *code_out, int32_t *bit_out, int32_t *neg_routing,
                       int32_t *int_bit)
{
    if (neg1 == 2) neg1 = 1; /* :787-789 */
    int32_t neg = neg1 + neg2; /* :804 */
    int32_t code = 0;          /* :812-821 */
    if (neg == 1) code = 1;
    if (neg == 2) code = 2;
    if (neg == 3) code = 4;
    int32_t ib;
    if (op == 4) { /* :833-844 */
        *bit_out = (bit1 >= bit2 ? bit1 : bit2) * 2;
        ib = bit1 >= bit2 ? bit1 : bit2;
    } else if (bit1 >= bit2) { /* :845-850 */
        ib = bit1;
        *bit_out = bit1;
    } else { /* :851-856 */
        ib = bit2;
        *bit_out = bit2;
    }
    *code_out = code;
    *neg_routing = neg;
    *int_bit = ib;
    if (op == 4 && ib >= 256) return 126; /* :860-864 */
    return 0;
}

/* cloud.c:870-2718, the value circuits.  opnd1 = ciphertext1..8,
 * opnd2 = ciphertext9..16 ([8][32] samples), carry1 = ciphertextcarry1. */
int orc_cloud_values(const orc_cloudkey *ck, int32_t op, int32_t neg, int32_t int_bit,
                     const int32_t *opnd1, const int32_t *opnd2, const int32_t *carry1,
                     int32_t *out)
{
    const size_t s = S(ck), word = 32 * s;
    if (op == 4 && int_bit >= 256) return 126;
    if (int_bit != 32 && int_bit != 64 && int_bit != 128 && int_bit != 256) return -1;
    const int W = int_bit / 32;
    int nwords_out = 0;
    int32_t *outw[9];
    for (int w = 0; w < 9; w++) outw[w] = out + (size_t)w * word;

    const int is_add = (op == 1 && (neg != 1 && neg != 2)) || (op == 2 && (neg == 1 || neg == 2)); /* :870 */
    const int is_sub = !is_add && (op == 2 || (op == 1 && (neg == 1 || neg == 2)));              /* :1194 */
    if (is_add) {
        /* :878-1189: W chained adds over word pairs */
        int32_t *cy_prev = NULL;
        for (int w = 0; w < W; w++) {
            int32_t *cy = new_array(ck, 32);
            orc_add(ck, outw[w], cy, opnd1 + w * word, opnd2 + w * word, w == 0 ? carry1 : cy_prev, 32);
            orc_scratch_free(ck, cy_prev);
            cy_prev = cy;
        }
        orc_scratch_free(ck, cy_prev);
        nwords_out = W;
    } else if (is_sub) {
        /* :1196 A-B / A+(-B): complement operand 2, add to operand 1.
         * else (:1809) (-A)+B: complement operand 1, add to operand 2. */
        const int second = (op == 2 && neg == 0) || (op == 1 && neg == 2);
        const int32_t *inv_src = second ? opnd2 : opnd1;
        const int32_t *keep = second ? opnd1 : opnd2;
        int32_t *temp = new_array(ck, 32);
        orc_zero(ck, temp, 32);            /* :1228 */
        orc_gate_constant(ck, temp, 1);    /* :1233 temp = 1 */
        int32_t *twos[8], *twoscarry[8];
        for (int w = 0; w < W; w++) {
            int32_t *inverse = new_array(ck, 32), *tempcarry = new_array(ck, 32);
            orc_NOT(ck, inverse, inv_src + w * word, 32); /* :1225 */
            orc_zero(ck, tempcarry, 32);                 /* :1229 */
            twos[w] = new_array(ck, 32);
            twoscarry[w] = new_array(ck, 32);
            if (w == 0)
                orc_add(ck, twos[0], twoscarry[0], inverse, temp, tempcarry, 32); /* :1236 */
            else
                orc_add(ck, twos[w], twoscarry[w], inverse, tempcarry, twoscarry[w - 1], 32); /* :1341 */
            orc_scratch_free(ck, inverse);
            orc_scratch_free(ck, tempcarry);
        }
        int32_t *cy_prev = NULL;
        for (int w = 0; w < W; w++) { /* :1245, :1352-1353 */
            int32_t *cy = new_array(ck, 32);
            orc_add(ck, outw[w], cy, keep + w * word, twos[w], w == 0 ? carry1 : cy_prev, 32);
            orc_scratch_free(ck, cy_prev);
            cy_prev = cy;
        }
        orc_scratch_free(ck, cy_prev);
        for (int w = 0; w < W; w++) {
            orc_scratch_free(ck, twos[w]);
            orc_scratch_free(ck, twoscarry[w]);
        }
        orc_scratch_free(ck, temp);
        nwords_out = W;
    } else if (op == 4) {
        if (int_bit == 32) { /* :2655-2718 */
            int32_t *r1 = new_array(ck, 32), *r2 = new_array(ck, 32);
            orc_mul32(ck, r1, r2, opnd1, opnd2, carry1, 32);
            copy_bits(ck, outw[0], r2, 32); /* low first :2683-2686 */
            copy_bits(ck, outw[1], r1, 32);
            orc_scratch_free(ck, r1);
            orc_scratch_free(ck, r2);
            nwords_out = 2;
        } else if (int_bit == 64) { /* :2568-2654 */
            int32_t *r[6], *f[3];
            for (int q = 0; q < 6; q++) r[q] = new_array(ck, 32);
            for (int q = 0; q < 3; q++) f[q] = new_array(ck, 32);
            orc_mul64(ck, r[0], r[1], r[2], opnd1, opnd1 + word, opnd2, carry1, 32);        /* :2589 */
            orc_mul64(ck, r[3], r[4], r[5], opnd1, opnd1 + word, opnd2 + word, carry1, 32); /* :2592 */
            orc_split(ck, f[0], f[1], f[2], r[0], r[1], r[3], r[4], r[5], carry1, 32);      /* :2594 */
            copy_bits(ck, outw[0], r[2], 32); /* :2609-2616 */
            copy_bits(ck, outw[1], f[2], 32);
            copy_bits(ck, outw[2], f[1], 32);
            copy_bits(ck, outw[3], f[0], 32);
            for (int q = 0; q < 6; q++) orc_scratch_free(ck, r[q]);
            for (int q = 0; q < 3; q++) orc_scratch_free(ck, f[q]);
            nwords_out = 4;
        } else if (int_bit == 128) { /* :2371-2567 */
            int32_t *r[21], *sm[16], *co[16];
            for (int q = 1; q <= 20; q++) r[q] = new_array(ck, 32);
            for (int q = 1; q <= 15; q++) {
                sm[q] = new_array(ck, 32);
                co[q] = new_array(ck, 32);
            }
            for (int q = 0; q < 4; q++) /* :2434-2443 */
                orc_mul128(ck, r[5 * q + 1], r[5 * q + 2], r[5 * q + 3], r[5 * q + 4], r[5 * q + 5],
                           opnd1, opnd1 + word, opnd1 + 2 * word, opnd1 + 3 * word,
                           opnd2 + q * word, carry1, 32);
            orc_add(ck, sm[1], co[1], r[10], r[4], carry1, 32); /* :2445-2449 */
            orc_add(ck, sm[2], co[2], r[9], r[3], co[1], 32);
            orc_add(ck, sm[3], co[3], r[8], r[2], co[2], 32);
            orc_add(ck, sm[4], co[4], r[7], r[1], co[3], 32);
            orc_add(ck, sm[5], co[5], r[6], carry1, co[4], 32);
            orc_add(ck, sm[6], co[6], sm[2], r[15], co[5], 32); /* :2451-2455 */
            orc_add(ck, sm[7], co[7], sm[3], r[14], co[6], 32);
            orc_add(ck, sm[8], co[8], sm[4], r[13], co[7], 32);
            orc_add(ck, sm[9], co[9], sm[5], r[12], co[8], 32);
            orc_add(ck, sm[10], co[10], r[11], carry1, co[9], 32);
            orc_add(ck, sm[11], co[11], sm[7], r[20], co[10], 32); /* :2457-2461 */
            orc_add(ck, sm[12], co[12], sm[8], r[19], co[11], 32);
            orc_add(ck, sm[13], co[13], sm[9], r[18], co[12], 32);
            orc_add(ck, sm[14], co[14], sm[10], r[17], co[13], 32);
            orc_add(ck, sm[15], co[15], r[16], carry1, co[14], 32);
            const int32_t *ex[8] = {r[5], sm[1], sm[6], sm[11], sm[12], sm[13], sm[14], sm[15]}; /* :2476-2491 */
            for (int q = 0; q < 8; q++) copy_bits(ck, outw[q], ex[q], 32);
            for (int q = 1; q <= 20; q++) orc_scratch_free(ck, r[q]);
            for (int q = 1; q <= 15; q++) {
                orc_scratch_free(ck, sm[q]);
                orc_scratch_free(ck, co[q]);
            }
            nwords_out = 8;
        } else
            return -1;
    } else
        return -1;
    /* unused words + the trailing carry word are operand 1's carry word (e.g. :901-916) */
    for (int w = nwords_out; w < 9; w++) copy_bits(ck, outw[w], carry1, 32);
    return 0;
}
