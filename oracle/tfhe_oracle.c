/*
 * oracle/tfhe_oracle.c -- CPU restatement of libtfhe gate bootstrapping as
 * used by /root/reference/Cloud/cloud.c (call sites cloud.c:21-48,56,62,159).
 *
 * TEST INFRASTRUCTURE ONLY (see tfhe_oracle.h).  "parity unpinned" at
 * ciphertext level: libtfhe itself (github.com/tfhe/tfhe master, un-pinned,
 * README.md:36-48) is not in /root/reference; every function below follows
 * the upstream file named in its comment as restated in SURVEY.md App. A.
 *
 * Plain C11, no dependencies.  gcc -O2 -fPIC -shared.
 */
#define _GNU_SOURCE
#include "tfhe_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Goldilocks field p = 2^64 - 2^32 + 1, used for an EXACT negacyclic  */
/* product: every true coefficient is bounded by                       */
/* (k+1)*l*N*(Bg/2)*2^31 < 2^51 << p/2, so no wrap occurs.             */
/* ------------------------------------------------------------------ */
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p */

static inline uint64_t gl_add(uint64_t a, uint64_t b)
{
    uint64_t r = a + b;
    if (r < a) r += GL_EPS; /* wrapped past 2^64 */
    if (r >= GL_P) r -= GL_P;
    return r;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b)
{
    uint64_t r = a - b;
    if (a < b) r -= GL_EPS; /* borrowed 2^64 */
    return r;
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b)
{
    unsigned __int128 x = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    uint64_t t0 = lo - hh; /* 2^96 = -1 */
    if (lo < hh) t0 -= GL_EPS;
    uint64_t t1 = hl * GL_EPS; /* 2^64 = 2^32-1 */
    uint64_t r = t0 + t1;
    if (r < t1) r += GL_EPS;
    if (r >= GL_P) r -= GL_P;
    return r;
}
static uint64_t gl_pow(uint64_t b, uint64_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline uint64_t gl_from_i64(int64_t v) { return v >= 0 ? (uint64_t)v : GL_P - (uint64_t)(-v); }
static inline int64_t gl_center(uint64_t v) { return v > GL_P / 2 ? -(int64_t)(GL_P - v) : (int64_t)v; }

typedef struct {
    int32_t N, logN;
    uint64_t *psi_rev;     /* psi^{bitrev(i)} */
    uint64_t *psi_inv_rev; /* psi^{-bitrev(i)} */
    uint64_t n_inv;
} gl_plan;

static uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

static void gl_plan_init(gl_plan *pl, int32_t N)
{
    pl->N = N;
    pl->logN = 0;
    while ((1 << pl->logN) < N) pl->logN++;
    pl->psi_rev = (uint64_t *)malloc(sizeof(uint64_t) * N);
    pl->psi_inv_rev = (uint64_t *)malloc(sizeof(uint64_t) * N);
    /* 7 generates the multiplicative group of the Goldilocks field */
    uint64_t psi = gl_pow(7, (GL_P - 1) / (2 * (uint64_t)N));
    uint64_t psi_inv = gl_pow(psi, GL_P - 2);
    uint64_t a = 1, b = 1;
    for (int32_t i = 0; i < N; i++) {
        uint32_t r = bitrev((uint32_t)i, pl->logN);
        pl->psi_rev[r] = a;
        pl->psi_inv_rev[r] = b;
        a = gl_mul(a, psi);
        b = gl_mul(b, psi_inv);
    }
    pl->n_inv = gl_pow((uint64_t)N, GL_P - 2);
}
static void gl_plan_free(gl_plan *pl)
{
    free(pl->psi_rev);
    free(pl->psi_inv_rev);
}
/* negacyclic forward transform, natural order in, bit-reversed out */
static void gl_ntt(const gl_plan *pl, uint64_t *a)
{
    int32_t N = pl->N, t = N;
    for (int32_t m = 1; m < N; m <<= 1) {
        t >>= 1;
        for (int32_t i = 0; i < m; i++) {
            int32_t j1 = 2 * i * t;
            uint64_t S = pl->psi_rev[m + i];
            for (int32_t j = j1; j < j1 + t; j++) {
                uint64_t U = a[j], V = gl_mul(a[j + t], S);
                a[j] = gl_add(U, V);
                a[j + t] = gl_sub(U, V);
            }
        }
    }
}
/* inverse: bit-reversed in, natural out, scaled by 1/N */
static void gl_intt(const gl_plan *pl, uint64_t *a)
{
    int32_t N = pl->N, t = 1;
    for (int32_t m = N; m > 1; m >>= 1) {
        int32_t h = m >> 1, j1 = 0;
        for (int32_t i = 0; i < h; i++) {
            uint64_t S = pl->psi_inv_rev[h + i];
            for (int32_t j = j1; j < j1 + t; j++) {
                uint64_t U = a[j], V = a[j + t];
                a[j] = gl_add(U, V);
                a[j + t] = gl_mul(gl_sub(U, V), S);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
    for (int32_t j = 0; j < N; j++) a[j] = gl_mul(a[j], pl->n_inv);
}

/* ------------------------------------------------------------------ */
/* Approximate FP64 FFT back-end (what libtfhe's                        */
/* tGswFFTExternMulToTLwe does: N/2-point complex transform of the      */
/* folded, twisted polynomial).  Not exact; never a parity target.      */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t N, M, logM;     /* M = N/2 complex points */
    double *tw_re, *tw_im;  /* twist: exp(i*pi*j/N), j<M */
    double *w_re, *w_im;    /* per stage, contiguous: stage with half-size h keeps exp(-2*pi*i*j/(2h)), j<h, at offset h - 1
                               (h = 1, 2, 4, ..., M/2: M - 1 entries) -- unit-stride twiddles let the compiler vectorise every
                               stage with h >= 4; the strided table this replaces (w[j*stride]) ran at about half the speed */
} fft_plan;

static void fft_plan_init(fft_plan *pl, int32_t N)
{
    pl->N = N;
    pl->M = N / 2;
    pl->logM = 0;
    while ((1 << pl->logM) < pl->M) pl->logM++;
    pl->tw_re = (double *)malloc(sizeof(double) * pl->M);
    pl->tw_im = (double *)malloc(sizeof(double) * pl->M);
    pl->w_re = (double *)malloc(sizeof(double) * (pl->M + 1));
    pl->w_im = (double *)malloc(sizeof(double) * (pl->M + 1));
    for (int32_t j = 0; j < pl->M; j++) {
        pl->tw_re[j] = cos(M_PI * j / N);
        pl->tw_im[j] = sin(M_PI * j / N);
    }
    for (int32_t h = 1; h < pl->M; h <<= 1)
        for (int32_t j = 0; j < h; j++) {
            pl->w_re[h - 1 + j] = cos(-M_PI * j / h);
            pl->w_im[h - 1 + j] = sin(-M_PI * j / h);
        }
}
static void fft_plan_free(fft_plan *pl)
{
    free(pl->tw_re);
    free(pl->tw_im);
    free(pl->w_re);
    free(pl->w_im);
}
/* forward DIF: natural in, bit-reversed out (no permutation pass) */
static void fft_fwd(const fft_plan *pl, double *restrict re, double *restrict im)
{
    const int32_t M = pl->M;
    for (int32_t half = M / 2; half >= 4; half >>= 1) {
        const double *restrict wr = pl->w_re + half - 1, *restrict wi = pl->w_im + half - 1;
        for (int32_t base = 0; base < M; base += 2 * half) {
            double *restrict ar = re + base, *restrict ai = im + base, *restrict br = ar + half, *restrict bi = ai + half;
            for (int32_t j = 0; j < half; j++) {
                const double ur = ar[j], ui = ai[j], vr = br[j], vi = bi[j];
                const double dr = ur - vr, di = ui - vi;
                ar[j] = ur + vr;
                ai[j] = ui + vi;
                br[j] = dr * wr[j] - di * wi[j];
                bi[j] = dr * wi[j] + di * wr[j];
            }
        }
    }
    /* the last two stages (half = 2, 1) as one radix-4 butterfly per four consecutive points: twiddles 1, -i */
    for (int32_t b = 0; b + 4 <= M; b += 4) {
        const double x0r = re[b], x0i = im[b], x1r = re[b + 1], x1i = im[b + 1], x2r = re[b + 2], x2i = im[b + 2], x3r = re[b + 3], x3i = im[b + 3];
        const double a0r = x0r + x2r, a0i = x0i + x2i, a1r = x1r + x3r, a1i = x1i + x3i;
        const double d0r = x0r - x2r, d0i = x0i - x2i;
        const double d1r = x1i - x3i, d1i = -(x1r - x3r); /* (x1 - x3) * (-i) */
        re[b] = a0r + a1r;
        im[b] = a0i + a1i;
        re[b + 1] = a0r - a1r;
        im[b + 1] = a0i - a1i;
        re[b + 2] = d0r + d1r;
        im[b + 2] = d0i + d1i;
        re[b + 3] = d0r - d1r;
        im[b + 3] = d0i - d1i;
    }
}
/* inverse DIT: bit-reversed in, natural out, unscaled */
static void fft_inv(const fft_plan *pl, double *restrict re, double *restrict im)
{
    const int32_t M = pl->M;
    for (int32_t b = 0; b + 4 <= M; b += 4) { /* half = 1, 2: twiddles 1, +i */
        const double x0r = re[b], x0i = im[b], x1r = re[b + 1], x1i = im[b + 1], x2r = re[b + 2], x2i = im[b + 2], x3r = re[b + 3], x3i = im[b + 3];
        const double a0r = x0r + x1r, a0i = x0i + x1i, a1r = x0r - x1r, a1i = x0i - x1i;
        const double c0r = x2r + x3r, c0i = x2i + x3i;
        const double c1r = -(x2i - x3i), c1i = x2r - x3r; /* (x2 - x3) * (+i) */
        re[b] = a0r + c0r;
        im[b] = a0i + c0i;
        re[b + 2] = a0r - c0r;
        im[b + 2] = a0i - c0i;
        re[b + 1] = a1r + c1r;
        im[b + 1] = a1i + c1i;
        re[b + 3] = a1r - c1r;
        im[b + 3] = a1i - c1i;
    }
    for (int32_t half = 4; half < M; half <<= 1) {
        const double *restrict wr = pl->w_re + half - 1, *restrict wi = pl->w_im + half - 1;
        for (int32_t base = 0; base < M; base += 2 * half) {
            double *restrict ar = re + base, *restrict ai = im + base, *restrict br = ar + half, *restrict bi = ai + half;
            for (int32_t j = 0; j < half; j++) {
                const double vr = br[j] * wr[j] + bi[j] * wi[j], vi = bi[j] * wr[j] - br[j] * wi[j]; /* * conj(w) */
                const double ur = ar[j], ui = ai[j];
                ar[j] = ur + vr;
                ai[j] = ui + vi;
                br[j] = ur - vr;
                bi[j] = ui - vi;
            }
        }
    }
}
/* int poly (N) -> folded/twisted spectrum (M complex) */
static void fft_from_int(const fft_plan *pl, const int32_t *p, double *re, double *im)
{
    int32_t M = pl->M;
    for (int32_t j = 0; j < M; j++) {
        double x = (double)p[j], y = (double)p[j + M];
        re[j] = x * pl->tw_re[j] - y * pl->tw_im[j];
        im[j] = x * pl->tw_im[j] + y * pl->tw_re[j];
    }
    fft_fwd(pl, re, im);
}
/* spectrum -> Torus32 poly (rounded, wrapped to 32 bits) */
static void fft_to_torus(const fft_plan *pl, double *re, double *im, int32_t *out)
{
    int32_t M = pl->M;
    fft_inv(pl, re, im);
    double s = 1.0 / M;
    for (int32_t j = 0; j < M; j++) {
        double x = (re[j] * pl->tw_re[j] + im[j] * pl->tw_im[j]) * s;
        double y = (im[j] * pl->tw_re[j] - re[j] * pl->tw_im[j]) * s;
        out[j] = (int32_t)(uint32_t)(uint64_t)(int64_t)llrint(x);
        out[j + M] = (int32_t)(uint32_t)(uint64_t)(int64_t)llrint(y);
    }
}

/* ------------------------------------------------------------------ */
struct orc_cloudkey {
    orc_params p;
    int32_t *bk;      /* [n][(k+1)l][k+1][N] */
    int32_t *ksk;     /* [kN][t][base][n+1] */
    int mode;
    gl_plan ntt;
    uint64_t *bk_ntt; /* same shape as bk */
    int have_fft;
    fft_plan fft;
    double *bk_fft;   /* [n][(k+1)l][k+1][2][M] */
    uint64_t nboot;
    struct orc_trace *trace; /* non-NULL while gates are being recorded (orc_defer_begin) */
};

static size_t bk_count(const orc_params *p)
{
    return (size_t)p->n * (p->k + 1) * p->l * (p->k + 1) * p->N;
}
static size_t ksk_count(const orc_params *p)
{
    return (size_t)p->k * p->N * p->ks_t * ((size_t)1 << p->ks_basebit) * (p->n + 1);
}

orc_cloudkey *orc_cloudkey_new(const orc_params *p, const int32_t *bk, const int32_t *ksk)
{
    if (p->k != 1 || (p->N & (p->N - 1)) || p->N < 4) return NULL;
    orc_cloudkey *ck = (orc_cloudkey *)calloc(1, sizeof(*ck));
    ck->p = *p;
    size_t nb = bk_count(p), nk = ksk_count(p);
    ck->bk = (int32_t *)malloc(nb * sizeof(int32_t));
    ck->ksk = (int32_t *)malloc(nk * sizeof(int32_t));
    memcpy(ck->bk, bk, nb * sizeof(int32_t));
    memcpy(ck->ksk, ksk, nk * sizeof(int32_t));
    gl_plan_init(&ck->ntt, p->N);
    ck->bk_ntt = (uint64_t *)malloc(nb * sizeof(uint64_t));
    for (size_t poly = 0; poly < nb / p->N; poly++) {
        uint64_t *d = ck->bk_ntt + poly * p->N;
        const int32_t *s = ck->bk + poly * p->N;
        for (int32_t j = 0; j < p->N; j++) d[j] = gl_from_i64(s[j]);
        gl_ntt(&ck->ntt, d);
    }
    ck->mode = ORC_POLYMUL_NTT;
    return ck;
}
void orc_cloudkey_free(orc_cloudkey *ck)
{
    if (!ck) return;
    free(ck->bk);
    free(ck->ksk);
    free(ck->bk_ntt);
    gl_plan_free(&ck->ntt);
    if (ck->have_fft) {
        fft_plan_free(&ck->fft);
        free(ck->bk_fft);
    }
    free(ck);
}
void orc_cloudkey_set_polymul(orc_cloudkey *ck, int mode)
{
    ck->mode = mode;
    if (mode == ORC_POLYMUL_FFT && !ck->have_fft) {
        const orc_params *p = &ck->p;
        fft_plan_init(&ck->fft, p->N);
        size_t npoly = bk_count(p) / p->N;
        ck->bk_fft = (double *)malloc(npoly * p->N * sizeof(double));
        for (size_t poly = 0; poly < npoly; poly++)
            fft_from_int(&ck->fft, ck->bk + poly * p->N, ck->bk_fft + poly * p->N,
                         ck->bk_fft + poly * p->N + p->N / 2);
        ck->have_fft = 1;
    }
}
const orc_params *orc_cloudkey_params(const orc_cloudkey *ck) { return &ck->p; }
uint64_t orc_cloudkey_bootstrap_count(const orc_cloudkey *ck) { return ck->nboot; }

/* ---- libtfhe numeric-functions.cpp ---- */
int32_t orc_modswitch_to_torus32(int32_t mu, int32_t Msize)
{
    uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    uint64_t phase64 = (uint64_t)(int64_t)mu * interv;
    return (int32_t)(phase64 >> 32);
}
int32_t orc_modswitch_from_torus32(int32_t phase, int32_t Msize)
{
    uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    uint64_t half = interv / 2;
    uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + half; /* wraps at 2^64 */
    return (int32_t)(phase64 / interv);
}

/* ---- libtfhe polynomials_arithmetic: torusPolynomialMulByXai ---- */
static void poly_mul_by_xai(int32_t N, int32_t *out, int32_t a, const int32_t *in)
{
    if (a < N) {
        for (int32_t i = 0; i < a; i++) out[i] = (int32_t)(0u - (uint32_t)in[i - a + N]);
        for (int32_t i = a; i < N; i++) out[i] = in[i - a];
    } else {
        int32_t aa = a - N;
        for (int32_t i = 0; i < aa; i++) out[i] = in[i - aa + N];
        for (int32_t i = aa; i < N; i++) out[i] = (int32_t)(0u - (uint32_t)in[i - aa]);
    }
}
/* torusPolynomialMulByXaiMinusOne */
static void poly_mul_by_xai_minus_one(int32_t N, int32_t *out, int32_t a, const int32_t *in)
{
    if (a < N) {
        for (int32_t i = 0; i < a; i++)
            out[i] = (int32_t)(0u - (uint32_t)in[i - a + N] - (uint32_t)in[i]);
        for (int32_t i = a; i < N; i++) out[i] = (int32_t)((uint32_t)in[i - a] - (uint32_t)in[i]);
    } else {
        int32_t aa = a - N;
        for (int32_t i = 0; i < aa; i++)
            out[i] = (int32_t)((uint32_t)in[i - aa + N] - (uint32_t)in[i]);
        for (int32_t i = aa; i < N; i++)
            out[i] = (int32_t)(0u - (uint32_t)in[i - aa] - (uint32_t)in[i]);
    }
}

/* exact schoolbook negacyclic product with int32 wraparound */
static void negacyclic_schoolbook_acc(int32_t N, uint32_t *acc, const int32_t *small,
                                      const int32_t *big)
{
    for (int32_t i = 0; i < N; i++) {
        uint32_t s = (uint32_t)small[i];
        if (!s) continue;
        for (int32_t j = 0; j < N - i; j++) acc[i + j] += s * (uint32_t)big[j];
        for (int32_t j = N - i; j < N; j++) acc[i + j - N] -= s * (uint32_t)big[j];
    }
}

void orc_negacyclic_mul(int mode, int32_t N, int32_t *out, const int32_t *small, const int32_t *big)
{
    if (mode == ORC_POLYMUL_SCHOOLBOOK) {
        uint32_t *acc = (uint32_t *)calloc(N, sizeof(uint32_t));
        negacyclic_schoolbook_acc(N, acc, small, big);
        for (int32_t j = 0; j < N; j++) out[j] = (int32_t)acc[j];
        free(acc);
    } else if (mode == ORC_POLYMUL_NTT) {
        gl_plan pl;
        gl_plan_init(&pl, N);
        uint64_t *a = (uint64_t *)malloc(sizeof(uint64_t) * N), *b = (uint64_t *)malloc(sizeof(uint64_t) * N);
        for (int32_t j = 0; j < N; j++) {
            a[j] = gl_from_i64(small[j]);
            b[j] = gl_from_i64(big[j]);
        }
        gl_ntt(&pl, a);
        gl_ntt(&pl, b);
        for (int32_t j = 0; j < N; j++) a[j] = gl_mul(a[j], b[j]);
        gl_intt(&pl, a);
        for (int32_t j = 0; j < N; j++) out[j] = (int32_t)(uint32_t)(uint64_t)gl_center(a[j]);
        free(a);
        free(b);
        gl_plan_free(&pl);
    } else {
        fft_plan pl;
        fft_plan_init(&pl, N);
        int32_t M = N / 2;
        double *a = (double *)malloc(sizeof(double) * N), *b = (double *)malloc(sizeof(double) * N);
        double *c = (double *)malloc(sizeof(double) * N);
        fft_from_int(&pl, small, a, a + M);
        fft_from_int(&pl, big, b, b + M);
        for (int32_t j = 0; j < M; j++) {
            c[j] = a[j] * b[j] - a[j + M] * b[j + M];
            c[j + M] = a[j] * b[j + M] + a[j + M] * b[j];
        }
        fft_to_torus(&pl, c, c + M, out);
        free(a);
        free(b);
        free(c);
        fft_plan_free(&pl);
    }
}

/* ---- lwe-bootstrapping-functions-fft.cpp: tfhe_bootstrap_woKS_FFT steps ---- */
int32_t orc_modswitch_sample(const orc_cloudkey *ck, const int32_t *x, int32_t *bara)
{
    const int32_t n = ck->p.n, Nx2 = 2 * ck->p.N;
    for (int32_t i = 0; i < n; i++) bara[i] = orc_modswitch_from_torus32(x[i], Nx2);
    return orc_modswitch_from_torus32(x[n], Nx2);
}

void orc_blind_rotate_init(const orc_cloudkey *ck, int32_t *acc, int32_t barb, int32_t mu)
{
    const int32_t N = ck->p.N, k = ck->p.k;
    int32_t *testvect = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int32_t i = 0; i < N; i++) testvect[i] = mu;
    memset(acc, 0, sizeof(int32_t) * (size_t)k * N);
    if (barb != 0)
        poly_mul_by_xai(N, acc + (size_t)k * N, 2 * N - barb, testvect);
    else
        memcpy(acc + (size_t)k * N, testvect, sizeof(int32_t) * N);
    free(testvect);
}

/* tgsw-functions.cpp: tGswTorus32PolynomialDecompH */
static void decomp_h(const orc_params *p, int32_t *dec /*[l][N]*/, const int32_t *poly)
{
    const int32_t N = p->N, l = p->l, Bgbit = p->Bgbit;
    const uint32_t halfBg = 1u << (Bgbit - 1), mask = (1u << Bgbit) - 1;
    uint32_t offset = 0;
    for (int32_t i = 1; i <= l; i++) offset += halfBg << (32 - i * Bgbit);
    for (int32_t j = 0; j < N; j++) {
        uint32_t v = (uint32_t)poly[j] + offset;
        for (int32_t q = 0; q < l; q++) {
            int32_t decal = 32 - (q + 1) * Bgbit;
            dec[(size_t)q * N + j] = (int32_t)((v >> decal) & mask) - (int32_t)halfBg;
        }
    }
}

/* tfhe_MuxRotate_FFT: acc += BK_i (x) ((X^barai - 1) acc)
 * (tLweMulByXaiMinusOne, tGswFFTExternMulToTLwe, tLweAddTo) */
void orc_blind_rotate_step(const orc_cloudkey *ck, int32_t *acc, int32_t i, int32_t barai)
{
    const orc_params *p = &ck->p;
    const int32_t N = p->N, k = p->k, l = p->l, kpl = (k + 1) * l;
    if (barai == 0) return; /* libtfhe skips; exact arithmetic makes it a no-op anyway */
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(k + 1) * N);
    int32_t *dec = (int32_t *)malloc(sizeof(int32_t) * (size_t)kpl * N);
    for (int32_t c = 0; c <= k; c++)
        poly_mul_by_xai_minus_one(N, tmp + (size_t)c * N, barai, acc + (size_t)c * N);
    for (int32_t c = 0; c <= k; c++) decomp_h(p, dec + (size_t)c * l * N, tmp + (size_t)c * N);
    const size_t bk_i = (size_t)i * kpl * (k + 1) * N;
    if (ck->mode == ORC_POLYMUL_SCHOOLBOOK) {
        for (int32_t c = 0; c <= k; c++) {
            uint32_t *out = (uint32_t *)(acc + (size_t)c * N);
            for (int32_t row = 0; row < kpl; row++)
                negacyclic_schoolbook_acc(N, out, dec + (size_t)row * N,
                                          ck->bk + bk_i + ((size_t)row * (k + 1) + c) * N);
        }
    } else if (ck->mode == ORC_POLYMUL_NTT) {
        uint64_t *dn = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)kpl * N);
        uint64_t *s = (uint64_t *)malloc(sizeof(uint64_t) * N);
        for (int32_t row = 0; row < kpl; row++) {
            uint64_t *d = dn + (size_t)row * N;
            for (int32_t j = 0; j < N; j++) d[j] = gl_from_i64(dec[(size_t)row * N + j]);
            gl_ntt(&ck->ntt, d);
        }
        for (int32_t c = 0; c <= k; c++) {
            memset(s, 0, sizeof(uint64_t) * N);
            for (int32_t row = 0; row < kpl; row++) {
                const uint64_t *b = ck->bk_ntt + bk_i + ((size_t)row * (k + 1) + c) * N;
                const uint64_t *d = dn + (size_t)row * N;
                for (int32_t j = 0; j < N; j++) s[j] = gl_add(s[j], gl_mul(d[j], b[j]));
            }
            gl_intt(&ck->ntt, s);
            uint32_t *out = (uint32_t *)(acc + (size_t)c * N);
            for (int32_t j = 0; j < N; j++) out[j] += (uint32_t)(uint64_t)gl_center(s[j]);
        }
        free(dn);
        free(s);
    } else {
        const int32_t M = N / 2;
        double *df = (double *)malloc(sizeof(double) * (size_t)kpl * N);
        double *s = (double *)malloc(sizeof(double) * N);
        int32_t *r = (int32_t *)malloc(sizeof(int32_t) * N);
        for (int32_t row = 0; row < kpl; row++)
            fft_from_int(&ck->fft, dec + (size_t)row * N, df + (size_t)row * N,
                         df + (size_t)row * N + M);
        for (int32_t c = 0; c <= k; c++) {
            memset(s, 0, sizeof(double) * N);
            for (int32_t row = 0; row < kpl; row++) {
                const double *b = ck->bk_fft + bk_i + ((size_t)row * (k + 1) + c) * N;
                const double *d = df + (size_t)row * N;
                for (int32_t j = 0; j < M; j++) {
                    s[j] += d[j] * b[j] - d[j + M] * b[j + M];
                    s[j + M] += d[j] * b[j + M] + d[j + M] * b[j];
                }
            }
            fft_to_torus(&ck->fft, s, s + M, r);
            uint32_t *out = (uint32_t *)(acc + (size_t)c * N);
            for (int32_t j = 0; j < N; j++) out[j] += (uint32_t)r[j];
        }
        free(df);
        free(s);
        free(r);
    }
    free(tmp);
    free(dec);
}

void orc_blind_rotate(const orc_cloudkey *ck, int32_t *acc, const int32_t *bara)
{
    for (int32_t i = 0; i < ck->p.n; i++) orc_blind_rotate_step(ck, acc, i, bara[i]);
}

/* tlwe-functions.cpp: tLweExtractLweSampleIndex(index=0) */
void orc_sample_extract(const orc_cloudkey *ck, int32_t *u, const int32_t *acc)
{
    const int32_t N = ck->p.N, k = ck->p.k;
    for (int32_t c = 0; c < k; c++) {
        const int32_t *a = acc + (size_t)c * N;
        u[(size_t)c * N] = a[0];
        for (int32_t j = 1; j < N; j++) u[(size_t)c * N + j] = (int32_t)(0u - (uint32_t)a[N - j]);
    }
    u[(size_t)k * N] = acc[(size_t)k * N];
}

/* lwe-keyswitch-functions.cpp: lweKeySwitch / lweKeySwitchTranslate_fromArray */
void orc_keyswitch(const orc_cloudkey *ck, int32_t *out, const int32_t *u)
{
    const orc_params *p = &ck->p;
    const int32_t n = p->n, Nin = p->k * p->N, t = p->ks_t, basebit = p->ks_basebit;
    const int32_t base = 1 << basebit, mask = base - 1;
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    uint32_t *r = (uint32_t *)out;
    int32_t b_in = u[Nin];
    for (int32_t j = 0; j < n; j++) r[j] = 0;
    r[n] = (uint32_t)b_in;
    for (int32_t i = 0; i < Nin; i++) {
        uint32_t aibar = (uint32_t)u[i] + prec_offset;
        for (int32_t j = 0; j < t; j++) {
            uint32_t aij = (aibar >> (32 - (j + 1) * basebit)) & (uint32_t)mask;
            if (aij != 0) {
                const int32_t *row = ck->ksk + (((size_t)i * t + j) * base + aij) * (n + 1);
                for (int32_t q = 0; q <= n; q++) r[q] -= (uint32_t)row[q];
            }
        }
    }
}

/* lwe-bootstrapping-functions-fft.cpp: tfhe_bootstrap_FFT with mu = 1/8 */
void orc_bootstrap(const orc_cloudkey *ck, int32_t *out, const int32_t *x)
{
    const orc_params *p = &ck->p;
    int32_t *bara = (int32_t *)malloc(sizeof(int32_t) * p->n);
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->k + 1) * p->N);
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * ((size_t)p->k * p->N + 1));
    const int32_t mu = orc_modswitch_to_torus32(1, 8);
    int32_t barb = orc_modswitch_sample(ck, x, bara);
    orc_blind_rotate_init(ck, acc, barb, mu);
    orc_blind_rotate(ck, acc, bara);
    orc_sample_extract(ck, u, acc);
    orc_keyswitch(ck, out, u);
    __atomic_fetch_add(&((orc_cloudkey *)ck)->nboot, 1, __ATOMIC_RELAXED);
    free(bara);
    free(acc);
    free(u);
}

/* ---- deferred evaluation -------------------------------------------------
 * The circuits of cloud_oracle.c issue their gates strictly one after the
 * other, as the reference does.  Between orc_defer_begin() and
 * orc_defer_run() the gate functions only RECORD what they were asked to do
 * (type, destination, operands, by address); orc_defer_run() then renames
 * every write (so reuse of a scratch sample creates no false dependency),
 * levelises the recorded gates and evaluates each level's independent gates
 * on several threads.  Every gate is still computed by exactly the code
 * below from exactly the operand values the sequential run would have used,
 * so the results are identical; only the wall time changes (golden vectors
 * at n = 630, all-cores cpu_baseline). */
enum { OP_AND, OP_XOR, OP_OR, OP_NAND, OP_NOT, OP_COPY, OP_CONST0, OP_CONST1, OP_MUX };
typedef struct {
    int32_t type;
    int32_t dep[3]; /* producing op of each operand, or -1: read the caller's memory */
    int32_t *dst;
    const int32_t *src[3];
    int32_t level;
} orc_op;
struct orc_trace {
    orc_op *ops;
    size_t n_ops, cap_ops;
    /* open-addressing map: sample address -> index of the op that last wrote it */
    const int32_t **keys;
    int32_t *vals;
    size_t map_cap, map_used;
    void **deferred_free;
    size_t n_free, cap_free;
};

static size_t trace_slot(const struct orc_trace *t, const int32_t *key)
{
    size_t h = ((uintptr_t)key >> 2) * (size_t)0x9E3779B97F4A7C15ULL;
    h ^= h >> 29;
    size_t i = h & (t->map_cap - 1);
    while (t->keys[i] && t->keys[i] != key) i = (i + 1) & (t->map_cap - 1);
    return i;
}
static void trace_grow(struct orc_trace *t)
{
    size_t old_cap = t->map_cap;
    const int32_t **ok = t->keys;
    int32_t *ov = t->vals;
    t->map_cap = old_cap ? old_cap * 2 : 1 << 16;
    t->keys = (const int32_t **)calloc(t->map_cap, sizeof(*t->keys));
    t->vals = (int32_t *)malloc(t->map_cap * sizeof(*t->vals));
    for (size_t i = 0; i < old_cap; i++)
        if (ok[i]) {
            size_t s = trace_slot(t, ok[i]);
            t->keys[s] = ok[i];
            t->vals[s] = ov[i];
        }
    free(ok);
    free(ov);
}
static int32_t trace_lookup(const struct orc_trace *t, const int32_t *key)
{
    if (!t->map_cap) return -1;
    size_t s = trace_slot(t, key);
    return t->keys[s] ? t->vals[s] : -1;
}
static void trace_record(const orc_cloudkey *ck, int32_t type, int32_t *dst, const int32_t *a,
                         const int32_t *b, const int32_t *c)
{
    struct orc_trace *t = ck->trace;
    if (t->n_ops == t->cap_ops) {
        t->cap_ops = t->cap_ops ? t->cap_ops * 2 : 4096;
        t->ops = (orc_op *)realloc(t->ops, t->cap_ops * sizeof(orc_op));
    }
    orc_op *op = &t->ops[t->n_ops];
    op->type = type;
    op->dst = dst;
    op->src[0] = a;
    op->src[1] = b;
    op->src[2] = c;
    op->level = 0;
    for (int q = 0; q < 3; q++) op->dep[q] = op->src[q] ? trace_lookup(t, op->src[q]) : -1;
    if ((t->map_used + 1) * 2 > t->map_cap) trace_grow(t);
    size_t s = trace_slot(t, dst);
    if (!t->keys[s]) {
        t->keys[s] = dst;
        t->map_used++;
    }
    t->vals[s] = (int32_t)t->n_ops;
    t->n_ops++;
}

void orc_defer_begin(orc_cloudkey *ck)
{
    if (ck->trace) return;
    ck->trace = (struct orc_trace *)calloc(1, sizeof(struct orc_trace));
}

/* cloud_oracle.c releases its scratch arrays through this: while gates are being
 * recorded the memory must stay (recorded operands are addresses) */
void orc_scratch_free(const orc_cloudkey *ck, void *p)
{
    if (!p) return;
    struct orc_trace *t = ck->trace;
    if (!t) {
        free(p);
        return;
    }
    if (t->n_free == t->cap_free) {
        t->cap_free = t->cap_free ? t->cap_free * 2 : 256;
        t->deferred_free = (void **)realloc(t->deferred_free, t->cap_free * sizeof(void *));
    }
    t->deferred_free[t->n_free++] = p;
}

static void gate2(const orc_cloudkey *ck, int32_t *out, int32_t cst, int32_t sa, const int32_t *ca,
                  int32_t sb, const int32_t *cb);
static void mux_now(const orc_cloudkey *ck, int32_t *out, const int32_t *a, const int32_t *b,
                    const int32_t *c);

static void run_op(const orc_cloudkey *ck, int32_t type, int32_t *out, const int32_t *a,
                   const int32_t *b, const int32_t *c)
{
    const int32_t n = ck->p.n, MU = orc_modswitch_to_torus32(1, 8);
    switch (type) {
    case OP_AND: gate2(ck, out, orc_modswitch_to_torus32(-1, 8), 1, a, 1, b); break;
    case OP_XOR: gate2(ck, out, orc_modswitch_to_torus32(1, 4), 2, a, 2, b); break;
    case OP_OR: gate2(ck, out, orc_modswitch_to_torus32(1, 8), 1, a, 1, b); break;
    case OP_NAND: gate2(ck, out, orc_modswitch_to_torus32(1, 8), -1, a, -1, b); break;
    case OP_NOT:
        for (int32_t j = 0; j <= n; j++) out[j] = (int32_t)(0u - (uint32_t)a[j]);
        break;
    case OP_COPY:
        if (out != a) memmove(out, a, sizeof(int32_t) * (n + 1));
        break;
    case OP_CONST0:
    case OP_CONST1:
        memset(out, 0, sizeof(int32_t) * n);
        out[n] = type == OP_CONST1 ? MU : (int32_t)(0u - (uint32_t)MU);
        break;
    default: mux_now(ck, out, a, b, c); break;
    }
}

/* Evaluates everything recorded since orc_defer_begin() on `nthreads` threads (<= 0: all
 * the host's), stores every sample where the sequential run would have left it, and
 * leaves deferred mode.  Returns the number of levels of bootstrapped gates it ran. */
int64_t orc_defer_run(orc_cloudkey *ck, int nthreads)
{
    struct orc_trace *t = ck->trace;
    if (!t) return 0;
    ck->trace = NULL; /* the gates below compute */
    const size_t S = (size_t)ck->p.n + 1;
    int32_t max_level = 0;
    for (size_t i = 0; i < t->n_ops; i++) {
        orc_op *op = &t->ops[i];
        int32_t lv = 0;
        for (int q = 0; q < 3; q++)
            if (op->dep[q] >= 0 && t->ops[op->dep[q]].level > lv) lv = t->ops[op->dep[q]].level;
        const int boots = op->type <= OP_NAND || op->type == OP_MUX;
        op->level = lv + (boots ? 1 : 0); /* free gates ride in their producer's level */
        if (op->level > max_level) max_level = op->level;
    }
    /* one value buffer per recorded write (SSA); ops of a level are independent of each other
     * but a free gate may read a value produced in its own level, so within a level the
     * bootstrapped gates run first (in parallel), then the free ones in recording order */
    int32_t *vals = (int32_t *)malloc(t->n_ops * S * sizeof(int32_t));
    size_t *order = (size_t *)malloc((t->n_ops + 1) * sizeof(size_t));
    size_t *first = (size_t *)calloc((size_t)max_level + 2, sizeof(size_t));
    for (size_t i = 0; i < t->n_ops; i++) first[t->ops[i].level + 1]++;
    for (int32_t l = 0; l <= max_level; l++) first[l + 1] += first[l];
    {
        size_t *fill = (size_t *)malloc(((size_t)max_level + 1) * sizeof(size_t));
        memcpy(fill, first, ((size_t)max_level + 1) * sizeof(size_t));
        for (size_t i = 0; i < t->n_ops; i++) order[fill[t->ops[i].level]++] = i;
        free(fill);
    }
#define OPERAND(op, q) ((op)->src[q] ? ((op)->dep[q] >= 0 ? vals + (size_t)(op)->dep[q] * S : (op)->src[q]) : NULL)
    for (int32_t l = 0; l <= max_level; l++) {
        const size_t lo = first[l], hi = first[l + 1];
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : orc_max_threads())
#endif
        for (size_t q = lo; q < hi; q++) {
            const orc_op *op = &t->ops[order[q]];
            if (!(op->type <= OP_NAND || op->type == OP_MUX)) continue;
            run_op(ck, op->type, vals + order[q] * S, OPERAND(op, 0), OPERAND(op, 1), OPERAND(op, 2));
        }
        for (size_t q = lo; q < hi; q++) {
            const orc_op *op = &t->ops[order[q]];
            if (op->type <= OP_NAND || op->type == OP_MUX) continue;
            run_op(ck, op->type, vals + order[q] * S, OPERAND(op, 0), OPERAND(op, 1), OPERAND(op, 2));
        }
    }
#undef OPERAND
    /* every address ends up holding its last recorded write */
    for (size_t i = 0; i < t->map_cap; i++)
        if (t->keys[i]) memcpy((int32_t *)t->keys[i], vals + (size_t)t->vals[i] * S, S * sizeof(int32_t));
    for (size_t i = 0; i < t->n_free; i++) free(t->deferred_free[i]);
    free(vals);
    free(order);
    free(first);
    free(t->ops);
    free(t->keys);
    free(t->vals);
    free(t->deferred_free);
    free(t);
    return max_level;
}

/* `count` independent gates of one type on `nthreads` threads (all-cores cpu_baseline) */
void orc_gates_batch(const orc_cloudkey *ck, int32_t type, size_t count, int32_t *out, const int32_t *a,
                     const int32_t *b, int nthreads)
{
    const size_t S = (size_t)ck->p.n + 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : orc_max_threads())
#endif
    for (size_t i = 0; i < count; i++) run_op(ck, type, out + i * S, a + i * S, b + i * S, NULL);
    (void)nthreads;
}

/* Threads worth starting: what OpenMP would use, capped by the CPU time this process is actually allowed
 * (cgroup v2 cpu.max, cgroup v1 cfs quota, the affinity mask) and by ORC_THREADS.  A GPU box hands a one-GPU job
 * 16 CPUs' worth of a 128-thread host: 128 spinning threads on a 16-CPU quota crawl at every level barrier. */
int orc_max_threads(void)
{
#ifdef _OPENMP
    long n = omp_get_num_procs();
    const char *env = getenv("ORC_THREADS");
    if (env && atol(env) > 0) return (int)atol(env);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char quota[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            long q = (atol(quota) + period - 1) / period;
            if (q >= 1 && q < n) n = q;
        }
        fclose(f);
    } else {
        long quota = -1, period = 0;
        f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        if (f) {
            if (fscanf(f, "%ld", &quota) != 1) quota = -1;
            fclose(f);
        }
        f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (f) {
            if (fscanf(f, "%ld", &period) != 1) period = 0;
            fclose(f);
        }
        if (quota > 0 && period > 0 && (quota + period - 1) / period < n) n = (quota + period - 1) / period;
    }
    if (n > 32) n = 32; /* the levels of these circuits are a few dozen gates wide: more threads only wait */
    return n < 1 ? 1 : (int)n;
#else
    return 1;
#endif
}

/* ---- boot-gates.cpp ---- */
void orc_gate_constant(const orc_cloudkey *ck, int32_t *out, int32_t value)
{
    if (ck->trace) {
        trace_record(ck, value ? OP_CONST1 : OP_CONST0, out, NULL, NULL, NULL);
        return;
    }
    run_op(ck, value ? OP_CONST1 : OP_CONST0, out, NULL, NULL, NULL);
}
void orc_gate_not(const orc_cloudkey *ck, int32_t *out, const int32_t *ca)
{
    if (ck->trace) {
        trace_record(ck, OP_NOT, out, ca, NULL, NULL);
        return;
    }
    run_op(ck, OP_NOT, out, ca, NULL, NULL);
}
void orc_gate_copy(const orc_cloudkey *ck, int32_t *out, const int32_t *ca)
{
    if (out == ca) return;
    if (ck->trace) {
        trace_record(ck, OP_COPY, out, ca, NULL, NULL);
        return;
    }
    run_op(ck, OP_COPY, out, ca, NULL, NULL);
}
/* t = (0, cst) + sa*ca + sb*cb, then bootstrap */
static void gate2(const orc_cloudkey *ck, int32_t *out, int32_t cst, int32_t sa, const int32_t *ca,
                  int32_t sb, const int32_t *cb)
{
    const int32_t n = ck->p.n;
    int32_t *t = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    for (int32_t j = 0; j <= n; j++)
        t[j] = (int32_t)((uint32_t)sa * (uint32_t)ca[j] + (uint32_t)sb * (uint32_t)cb[j]);
    t[n] = (int32_t)((uint32_t)t[n] + (uint32_t)cst);
    orc_bootstrap(ck, out, t);
    free(t);
}
#define ORC_GATE2(name, OP)                                                                          \
    void orc_gate_##name(const orc_cloudkey *ck, int32_t *out, const int32_t *ca, const int32_t *cb) \
    {                                                                                                \
        if (ck->trace) {                                                                             \
            trace_record(ck, OP, out, ca, cb, NULL);                                                 \
            return;                                                                                  \
        }                                                                                            \
        run_op(ck, OP, out, ca, cb, NULL);                                                           \
    }
ORC_GATE2(and, OP_AND)   /* (0,-1/8) + ca + cb */
ORC_GATE2(xor, OP_XOR)   /* (0, 1/4) + 2(ca + cb) */
ORC_GATE2(or, OP_OR)     /* (0, 1/8) + ca + cb */
ORC_GATE2(nand, OP_NAND) /* (0, 1/8) - ca - cb */

/* tfhe_bootstrap_woKS_FFT: bootstrap without the final key switch; u is [kN+1] */
void orc_bootstrap_woks(const orc_cloudkey *ck, int32_t *u, const int32_t *x)
{
    const orc_params *p = &ck->p;
    int32_t *bara = (int32_t *)malloc(sizeof(int32_t) * p->n);
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->k + 1) * p->N);
    const int32_t mu = orc_modswitch_to_torus32(1, 8);
    int32_t barb = orc_modswitch_sample(ck, x, bara);
    orc_blind_rotate_init(ck, acc, barb, mu);
    orc_blind_rotate(ck, acc, bara);
    orc_sample_extract(ck, u, acc);
    free(bara);
    free(acc);
}

/* boot-gates.cpp bootsMUX(a, b, c) = a ? b : c.  Two bootstraps WITHOUT key switch
 *   u1 = bootstrap_woKS((0,-1/8) + a + b)      "a AND b"
 *   u2 = bootstrap_woKS((0,-1/8) - a + c)      "(NOT a) AND c"
 * then ONE key switch of (0, 1/8) + u1 + u2.  Unused by Cloud/cloud.c;
 * BASELINE.json's north_star names it. */
static void mux_now(const orc_cloudkey *ck, int32_t *out, const int32_t *a, const int32_t *b,
                    const int32_t *c)
{
    const int32_t n = ck->p.n, Nin = ck->p.k * ck->p.N;
    const uint32_t and_const = (uint32_t)orc_modswitch_to_torus32(-1, 8);
    int32_t *t = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    int32_t *u1 = (int32_t *)malloc(sizeof(int32_t) * (Nin + 1));
    int32_t *u2 = (int32_t *)malloc(sizeof(int32_t) * (Nin + 1));
    for (int32_t j = 0; j <= n; j++) t[j] = (int32_t)((uint32_t)a[j] + (uint32_t)b[j]);
    t[n] = (int32_t)((uint32_t)t[n] + and_const);
    orc_bootstrap_woks(ck, u1, t);
    for (int32_t j = 0; j <= n; j++) t[j] = (int32_t)((uint32_t)c[j] - (uint32_t)a[j]);
    t[n] = (int32_t)((uint32_t)t[n] + and_const);
    orc_bootstrap_woks(ck, u2, t);
    for (int32_t j = 0; j <= Nin; j++) u1[j] = (int32_t)((uint32_t)u1[j] + (uint32_t)u2[j]);
    u1[Nin] = (int32_t)((uint32_t)u1[Nin] + (uint32_t)orc_modswitch_to_torus32(1, 8));
    orc_keyswitch(ck, out, u1);
    __atomic_fetch_add(&((orc_cloudkey *)ck)->nboot, 2, __ATOMIC_RELAXED);
    free(t);
    free(u1);
    free(u2);
}
void orc_gate_mux(const orc_cloudkey *ck, int32_t *out, const int32_t *a, const int32_t *b,
                  const int32_t *c)
{
    if (ck->trace) {
        trace_record(ck, OP_MUX, out, a, b, c);
        return;
    }
    mux_now(ck, out, a, b, c);
}

int32_t orc_lwe_phase(const int32_t *sample, const int32_t *key, int32_t n)
{
    uint32_t acc = (uint32_t)sample[n];
    for (int32_t i = 0; i < n; i++) acc -= (uint32_t)sample[i] * (uint32_t)key[i];
    return (int32_t)acc;
}
