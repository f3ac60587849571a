/*
 * oracle/tfhe_oracle.h -- CPU restatement of the TFHE gate-bootstrapping path
 * that /root/reference/Cloud/cloud.c drives through libtfhe.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY STATUS: "parity unpinned" at ciphertext level.  libtfhe
 * (github.com/tfhe/tfhe, branch master, no pin; README.md:36-48) is an
 * un-vendored dependency that is absent from /root/reference, the reference
 * holds no golden ciphertext vectors, and its binaries are Mach-O.  The
 * algorithm below restates libtfhe's published gate bootstrapping
 *  with the external product evaluated in EXACT integer
 * arithmetic (libtfhe's own FP64 FFT is approximate and differs between its
 * FFT back-ends).  At plaintext level the oracle IS pinned: by the
 * reference's canned operands (Client1/process.c:94-99,122-129,152-163,
 * 185-204), the bit/word layout (Client1/alice.c:116-149), the output
 * interpretation (Output/verif.c:92-179) and integer arithmetic -- see
 * tests/test_oracle_*.py.
 *
 * An LWE sample is int32_t[n+1]: a[0..n-1] then b (Torus32 = int32 with
 * wraparound).  All keys are raw int32 arrays in the order libtfhe holds them:
 *   BK  [n][(k+1)*l][k+1][N]      TGSW rows, polys a_0..a_{k-1}, b
 *   KSK [k*N][t][base][n+1]       LWE samples (a.., b), entry d=0 unused
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t n;          /* LWE dimension (630) */
    int32_t N;          /* TLWE ring degree (1024), power of two */
    int32_t k;          /* TLWE mask polys; only k=1 is supported */
    int32_t l;          /* TGSW decomposition length (3) */
    int32_t Bgbit;      /* TGSW decomposition base bits (7) */
    int32_t ks_t;       /* key-switch length (8) */
    int32_t ks_basebit; /* key-switch base bits (2) */
} orc_params;

typedef struct orc_cloudkey orc_cloudkey;

/* polynomial-product back-ends, all computing the same thing:
 *   0 = exact NTT over p = 2^64-2^32+1   (default; unarguable integer math)
 *   1 = exact schoolbook with int32 wraparound (slow cross-check)
 *   2 = approximate FP64 FFT (what libtfhe does; NOT a parity target --
 *       used only as the speed-representative cpu_baseline and for
 *       decrypt-level comparisons) */
enum { ORC_POLYMUL_NTT = 0, ORC_POLYMUL_SCHOOLBOOK = 1, ORC_POLYMUL_FFT = 2 };

orc_cloudkey *orc_cloudkey_new(const orc_params *p, const int32_t *bk, const int32_t *ksk);
void orc_cloudkey_free(orc_cloudkey *ck);
void orc_cloudkey_set_polymul(orc_cloudkey *ck, int mode);
const orc_params *orc_cloudkey_params(const orc_cloudkey *ck);

/* ---- numeric helpers (libtfhe numeric-functions.cpp restated) ---- */
int32_t orc_modswitch_to_torus32(int32_t mu, int32_t Msize);
int32_t orc_modswitch_from_torus32(int32_t phase, int32_t Msize);

/* ---- pipeline stages, exposed one by one so tests can compare each GPU
 *      kernel against its own stage ---- */
/* bara[0..n-1], returns barb; x is an LWE sample under the n-key */
int32_t orc_modswitch_sample(const orc_cloudkey *ck, const int32_t *x, int32_t *bara);
/* acc = (0, X^{2N-barb} * (mu,...,mu)); acc is [(k+1)][N] */
void orc_blind_rotate_init(const orc_cloudkey *ck, int32_t *acc, int32_t barb, int32_t mu);
/* one CMux step: acc += BK_i (x) ((X^{barai}-1) * acc) */
void orc_blind_rotate_step(const orc_cloudkey *ck, int32_t *acc, int32_t i, int32_t barai);
/* all n steps */
void orc_blind_rotate(const orc_cloudkey *ck, int32_t *acc, const int32_t *bara);
/* u[0..kN-1], u[kN] = b : LWE sample under the extracted key */
void orc_sample_extract(const orc_cloudkey *ck, int32_t *u, const int32_t *acc);
/* out (n+1) = keyswitch(u) */
void orc_keyswitch(const orc_cloudkey *ck, int32_t *out, const int32_t *u);
/* tfhe_bootstrap_FFT(result, bk, MU=1/8, x) */
void orc_bootstrap(const orc_cloudkey *ck, int32_t *out, const int32_t *x);

/* ---- gates (libtfhe boot-gates.cpp restated; out may alias inputs) ---- */
void orc_gate_constant(const orc_cloudkey *ck, int32_t *out, int32_t value);
void orc_gate_not(const orc_cloudkey *ck, int32_t *out, const int32_t *ca);
void orc_gate_copy(const orc_cloudkey *ck, int32_t *out, const int32_t *ca);
void orc_gate_and(const orc_cloudkey *ck, int32_t *out, const int32_t *ca, const int32_t *cb);
void orc_gate_xor(const orc_cloudkey *ck, int32_t *out, const int32_t *ca, const int32_t *cb);
void orc_gate_or(const orc_cloudkey *ck, int32_t *out, const int32_t *ca, const int32_t *cb);
void orc_gate_nand(const orc_cloudkey *ck, int32_t *out, const int32_t *ca, const int32_t *cb);
/* bootsMUX(a,b,c) = a ? b : c (two bootstraps without key switch + one key switch) */
void orc_gate_mux(const orc_cloudkey *ck, int32_t *out, const int32_t *a, const int32_t *b,
                  const int32_t *c);
/* tfhe_bootstrap_woKS_FFT: u is [k*N+1] under the extracted key */
void orc_bootstrap_woks(const orc_cloudkey *ck, int32_t *u, const int32_t *x);
/* number of bootstraps performed through this key since creation */
uint64_t orc_cloudkey_bootstrap_count(const orc_cloudkey *ck);

/* ---- deferred (level-parallel) evaluation of the SAME sequential gate stream ----
 * Between orc_defer_begin() and orc_defer_run() every orc_gate_* call is only
 * recorded; orc_defer_run() evaluates the recorded gates level by level on
 * `nthreads` host threads (<= 0: all) and leaves every sample where the
 * sequential run would have left it.  Bit-identical to the immediate mode. */
void orc_defer_begin(orc_cloudkey *ck);
int64_t orc_defer_run(orc_cloudkey *ck, int nthreads);
void orc_scratch_free(const orc_cloudkey *ck, void *p);
/* `count` independent two-input gates (type: 0 AND, 1 XOR, 2 OR, 3 NAND) on nthreads threads */
void orc_gates_batch(const orc_cloudkey *ck, int32_t type, size_t count, int32_t *out,
                     const int32_t *a, const int32_t *b, int nthreads);
int orc_max_threads(void);

/* exact negacyclic product mod (X^N+1, 2^32), for cross-checking back-ends */
void orc_negacyclic_mul(int mode, int32_t N, int32_t *out, const int32_t *small, const int32_t *big);

/* ---- LWE phase (for decrypt-level checks): b - <a,s>, key bits s[n] ---- */
int32_t orc_lwe_phase(const int32_t *sample, const int32_t *key, int32_t n);

/* ---- circuits of Cloud/cloud.c, gate by gate, in the reference's order.
 *      Every array is `count` samples of stride (n+1).  See cloud_oracle.c. */
void orc_add(const orc_cloudkey *ck, int32_t *sum, int32_t *carryover, const int32_t *x,
             const int32_t *y, const int32_t *c, int32_t nb_bits);
void orc_zero(const orc_cloudkey *ck, int32_t *result, size_t size);
void orc_NOT(const orc_cloudkey *ck, int32_t *result, const int32_t *x, size_t size);
void orc_split(const orc_cloudkey *ck, int32_t *f1, int32_t *f2, int32_t *f3, const int32_t *a,
               const int32_t *b, const int32_t *c, const int32_t *d, const int32_t *e,
               const int32_t *carry, int32_t nb_bits);
void orc_mul32(const orc_cloudkey *ck, int32_t *result, int32_t *result2, const int32_t *a,
               const int32_t *b, const int32_t *carry, int32_t nb_bits);
void orc_mul64(const orc_cloudkey *ck, int32_t *result, int32_t *result2, int32_t *result3,
               const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *carry,
               int32_t nb_bits);
void orc_mul128(const orc_cloudkey *ck, int32_t *r1, int32_t *r2, int32_t *r3, int32_t *r4,
                int32_t *r5, const int32_t *a, const int32_t *b, const int32_t *c,
                const int32_t *d, const int32_t *e, const int32_t *carry, int32_t nb_bits);

/* Value part of cloud.c main(): operands are [8 words][32 samples] each plus
 * the operand-1 carry word [32 samples]; `out` receives 9 words x 32 samples
 * exactly as main() exports them after the 64 metadata samples.
 * op in {1,2,4}; neg in {0,1,2,3} (already combined, cloud.c:787-804);
 * int_bit in {32,64,128,256}.  Returns 0, 126 (MUL at >=256 bit,
 * cloud.c:860-864) or -1 when main() would fall through without output. */
int orc_cloud_values(const orc_cloudkey *ck, int32_t op, int32_t neg, int32_t int_bit,
                     const int32_t *opnd1, const int32_t *opnd2, const int32_t *carry1,
                     int32_t *out);

/* Metadata arithmetic of main() in the clear (cloud.c:775-864).
 * neg1/neg2/bit1/bit2 are the decrypted words.  Outputs the sign code and bit
 * word main() encrypts into answer.data, the combined `neg` routing value and
 * the int_bit used for dispatch.  Returns 0 or 126. */
int orc_cloud_metadata(int32_t op, int32_t neg1, int32_t bit1, int32_t neg2, int32_t bit2,
                       int32_t *code_out, int32_t *bit_out, int32_t *neg_routing,
                       int32_t *int_bit);

#ifdef __cplusplus
}
#endif
#endif
