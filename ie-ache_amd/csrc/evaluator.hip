// Device evaluator: level-batched gate bootstrapping on MI355X (gfx950).
//
// Per gate instance (one bootsAND/bootsXOR of the reference, cloud.c:30-43,159):
//   K0 gate pre-combination   t = cst + sa*ca + sb*cb          (boot-gates.cpp)
//   K1 mod-switch             bara_i, barb in [0,2N)            (numeric-functions.cpp)
//   K2 test-vector init       acc = (0, X^{2N-barb} * mu)       (lwe-bootstrapping-functions-fft.cpp)
//   K3 blind rotation         n x  acc += BK_i (x) ((X^bara_i - 1) acc)
//   K4 sample extract
//   K5 key switch                                                 (lwe-keyswitch-functions.cpp)
// K0-K4 are one kernel (k_blind_rotate_*), K5 is a second (k_keyswitch_*).
//
// The external product is EXACT.  libtfhe multiplies polynomials with an
// approximate FP64 FFT; here every BK polynomial is split into two balanced
// 16-bit limbs before the transform, so every inverse-transform output is an
// integer of magnitude < 2^37 carried with > 15 spare mantissa bits and
// rounding recovers it exactly.  Results therefore equal the integer
// definition (and the CPU oracle) bit for bit, whatever the FFT schedule.
#include "evaluator.h"

#include <cstring>

#include "blind_rotate_w64.h"
#include "keyswitch_mfma.h"
#include "keyswitch_sliced.h"
#include "device_common.h"
#include "mix_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace ieache {

void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e == hipSuccess) return;
    (void)hipGetLastError();  // the runtime keeps a failure as "last error": clear it, or a later, healthy call's
                              // hipGetLastError() check would report this one again
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    throw std::runtime_error(buf);
}

namespace {

constexpr int kThreads = 256;

using namespace dev;

// In-LDS radix-2 transforms over `npoly` polynomials of M complex points.
// Forward: DIF, natural in -> bit-reversed out.  Inverse: DIT, bit-reversed in
// -> natural out, unscaled.  Neither needs a permutation pass.
__device__ void fft_forward_lds(double2* F, int32_t npoly, int32_t M, int32_t logM, const double2* wtab) {
    const int32_t halfM = M >> 1, total = npoly * halfM;
    for (int32_t sh = 0; sh < logM; sh++) {
        const int32_t half = halfM >> sh;
        for (int32_t t = threadIdx.x; t < total; t += blockDim.x) {
            const int32_t poly = t / halfM, bf = t - poly * halfM;
            const int32_t j = bf & (half - 1), grp = bf >> (logM - 1 - sh);
            const int32_t a = poly * M + (grp * 2 * half) + j, b = a + half;
            const double2 w = wtab[j << sh];
            const double2 u = F[a], v = F[b];
            F[a] = make_double2(u.x + v.x, u.y + v.y);
            F[b] = cmul(make_double2(u.x - v.x, u.y - v.y), w);
        }
        __syncthreads();
    }
}
__device__ void fft_inverse_lds(double2* F, int32_t npoly, int32_t M, int32_t logM, const double2* wtab) {
    const int32_t halfM = M >> 1, total = npoly * halfM;
    for (int32_t st = 0; st < logM; st++) {
        const int32_t half = 1 << st, sh = logM - 1 - st;
        for (int32_t t = threadIdx.x; t < total; t += blockDim.x) {
            const int32_t poly = t / halfM, bf = t - poly * halfM;
            const int32_t j = bf & (half - 1), grp = bf >> st;
            const int32_t a = poly * M + (grp * 2 * half) + j, b = a + half;
            const double2 w = wtab[j << sh];
            const double2 u = F[a], v = cmul_conj(F[b], w);
            F[a] = make_double2(u.x + v.x, u.y + v.y);
            F[b] = make_double2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

// ---- key preparation: BK polynomial -> two-limb spectrum ----
__global__ __launch_bounds__(kThreads) void k_bk_to_spectrum(DevKeys K, const Torus32* bk_raw, double2* bkf) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2* F = reinterpret_cast<double2*>(smem);  // [2][M]
    const int32_t M = K.M;
    const Torus32* src = bk_raw + (size_t)blockIdx.x * K.N;
    for (int32_t j = threadIdx.x; j < M; j += blockDim.x) {
        const int32_t v0 = src[j], v1 = src[j + M];
        const int32_t lo0 = (int16_t)(v0 & 0xFFFF), lo1 = (int16_t)(v1 & 0xFFFF);
        const int32_t hi0 = (int32_t)(((int64_t)v0 - lo0) >> 16), hi1 = (int32_t)(((int64_t)v1 - lo1) >> 16);
        const double2 tw = K.twist[j];
        F[j] = cmul(make_double2((double)lo0, (double)lo1), tw);
        F[M + j] = cmul(make_double2((double)hi0, (double)hi1), tw);
    }
    __syncthreads();
    fft_forward_lds(F, 2, M, K.logM, K.wtab);
    double2* dst = bkf + (size_t)blockIdx.x * 2 * M;
    for (int32_t j = threadIdx.x; j < 2 * M; j += blockDim.x) dst[j] = F[j];
}

// raw KSK rows (n+1) -> padded rows (stride)
__global__ void k_pad_rows(const Torus32* src, Torus32* dst, int64_t rows, int32_t width, int32_t stride) {
    const int64_t total = rows * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / stride;
        const int32_t c = (int32_t)(i - r * stride);
        dst[i] = c < width ? src[r * width + c] : 0;
    }
}

// ---- K0..K4, generic parameters: one workgroup per gate instance ----
// LDS: F [max(kpl,4)][M] double2 | acc [2][N] int32 | bara [n] u16
__global__ __launch_bounds__(kThreads) void k_blind_rotate_generic(DevKeys K, WorkDesc W, Torus32* ext,
                                                                   int32_t steps, Torus32* dbg_acc) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t N = K.N, M = K.M, n = K.n, l = K.l, kpl = K.kpl;
    const int32_t frows = kpl > 4 ? kpl : 4;
    double2* F = reinterpret_cast<double2*>(smem);
    int32_t* acc = reinterpret_cast<int32_t*>(F + (size_t)frows * M);
    uint16_t* bara = reinterpret_cast<uint16_t*>(acc + 2 * N);
    __shared__ int32_t s_barb;

    const int64_t item = (int64_t)blockIdx.x;
    const GateInst g = resolve(W, W.item0 + item, K.stride);
    const int32_t log2N2 = K.logM + 2;

    // K0 + K1
    for (int32_t i = threadIdx.x; i <= n; i += blockDim.x) {
        const int32_t bar = modswitch2N(combined_coef(g, i, n), log2N2);
        if (i < n)
            bara[i] = (uint16_t)bar;
        else
            s_barb = bar;
    }
    __syncthreads();
    // K2: acc = (0, X^{2N-barb} * (mu,...,mu))
    {
        const int32_t a0 = (2 * N - s_barb) & (2 * N - 1);
        for (int32_t j = threadIdx.x; j < N; j += blockDim.x) {
            acc[j] = 0;
            const int32_t idx = (j - a0) & (2 * N - 1);
            acc[N + j] = idx < N ? kMU : -kMU;
        }
    }
    __syncthreads();

    const uint32_t halfBg = 1u << (K.Bgbit - 1), maskBg = (1u << K.Bgbit) - 1;
    const double invM = 1.0 / (double)M;
    const int32_t nsteps = steps < 0 ? n : steps;
    // K3
    for (int32_t i = 0; i < nsteps; i++) {
        const int32_t a = bara[i];
        if (a == 0) continue;  // uniform across the workgroup; exact arithmetic makes the step a no-op
        // (X^a - 1) * acc, gadget decomposition, fold + twist
        for (int32_t j = threadIdx.x; j < M; j += blockDim.x) {
            const double2 tw = K.twist[j];
#pragma unroll 2
            for (int32_t c = 0; c < 2; c++) {
                const int32_t* p = acc + c * N;
                const uint32_t d0 = (uint32_t)rot_coef(p, j, a, N) - (uint32_t)p[j] + K.dec_offset;
                const uint32_t d1 = (uint32_t)rot_coef(p, j + M, a, N) - (uint32_t)p[j + M] + K.dec_offset;
                for (int32_t q = 0; q < l; q++) {
                    const int32_t sh = 32 - (q + 1) * K.Bgbit;
                    const int32_t e0 = (int32_t)((d0 >> sh) & maskBg) - (int32_t)halfBg;
                    const int32_t e1 = (int32_t)((d1 >> sh) & maskBg) - (int32_t)halfBg;
                    F[(size_t)(c * l + q) * M + j] = cmul(make_double2((double)e0, (double)e1), tw);
                }
            }
        }
        __syncthreads();
        fft_forward_lds(F, kpl, M, K.logM, K.wtab);
        // spectrum-domain accumulate: out(c,limb) = sum_row dec[row] * BK_i[row][c][limb]
        const double2* bki = K.bkf + (size_t)i * kpl * 4 * M;
        for (int32_t pt = threadIdx.x; pt < M; pt += blockDim.x) {
            double2 s[4];
#pragma unroll
            for (int32_t q = 0; q < 4; q++) s[q] = make_double2(0.0, 0.0);
            for (int32_t row = 0; row < kpl; row++) {
                const double2 d = F[(size_t)row * M + pt];
                const double2* b = bki + (size_t)row * 4 * M + pt;
#pragma unroll
                for (int32_t q = 0; q < 4; q++) s[q] = cfma(d, b[(size_t)q * M], s[q]);
            }
            // every thread has consumed its own column of F; rows 0..3 become the outputs
#pragma unroll
            for (int32_t q = 0; q < 4; q++) F[(size_t)q * M + pt] = s[q];
        }
        __syncthreads();
        fft_inverse_lds(F, 4, M, K.logM, K.wtab);
        // untwist, round, recombine limbs, accumulate
        for (int32_t j = threadIdx.x; j < M; j += blockDim.x) {
            const double2 tw = K.twist[j];
#pragma unroll 2
            for (int32_t c = 0; c < 2; c++) {
                const double2 lo = cmul_conj(F[(size_t)(2 * c) * M + j], tw);
                const double2 hi = cmul_conj(F[(size_t)(2 * c + 1) * M + j], tw);
                const int64_t r0 = __double2ll_rn(lo.x * invM) + (__double2ll_rn(hi.x * invM) << 16);
                const int64_t r1 = __double2ll_rn(lo.y * invM) + (__double2ll_rn(hi.y * invM) << 16);
                acc[c * N + j] = (int32_t)((uint32_t)acc[c * N + j] + (uint32_t)r0);
                acc[c * N + j + M] = (int32_t)((uint32_t)acc[c * N + j + M] + (uint32_t)r1);
            }
        }
        __syncthreads();
    }
    if (dbg_acc) {
        for (int32_t j = threadIdx.x; j < 2 * N; j += blockDim.x) dbg_acc[(size_t)item * 2 * N + j] = acc[j];
    }
    // K4: u = (a'_0 = acc.a_0, a'_j = -acc.a_{N-j}; b' = acc.b_0)
    if (ext) {
        Torus32* u = ext + (size_t)item * (N + 4);
        for (int32_t j = threadIdx.x; j <= N; j += blockDim.x)
            u[j] = j == 0 ? acc[0] : (j == N ? acc[N] : (int32_t)(0u - (uint32_t)acc[N - j]));
    }
}

// ---- K5, generic: one workgroup per gate instance ----
// LDS: u [N+1] | list [N*t] | count
__global__ __launch_bounds__(kThreads) void k_keyswitch_generic(DevKeys K, WorkDesc W, const Torus32* ext,
                                                                Torus32* flat_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t N = K.N, n = K.n, t = K.ks_t, basebit = K.ks_basebit, stride = K.stride;
    int32_t* u = reinterpret_cast<int32_t*>(smem);
    uint32_t* list = reinterpret_cast<uint32_t*>(u + N + 4);
    __shared__ uint32_t s_count;
    const int64_t item = (int64_t)blockIdx.x;
    const Torus32* src = ext + (size_t)item * (N + 4);
    if (threadIdx.x == 0) s_count = 0;
    for (int32_t j = threadIdx.x; j <= N; j += blockDim.x) u[j] = src[j];
    __syncthreads();
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    const uint32_t mask = (1u << basebit) - 1;
    for (int32_t idx = threadIdx.x; idx < N * t; idx += blockDim.x) {
        const int32_t i = idx / t, j = idx - i * t;
        const uint32_t d = (((uint32_t)u[i] + prec_offset) >> (32 - (j + 1) * basebit)) & mask;
        if (d) list[atomicAdd(&s_count, 1u)] = ((uint32_t)idx << basebit) + d;  // row index [i][j][d]
    }
    __syncthreads();
    const uint32_t cnt = s_count;
    // subtraction mod 2^32 commutes, so the (non-deterministic) list order does not matter
    uint32_t r0 = 0, r1 = 0, r2 = 0;
    const int32_t q0 = threadIdx.x, q1 = q0 + kThreads, q2 = q0 + 2 * kThreads;
    for (uint32_t e = 0; e < cnt; e++) {
        const int32_t* row = K.ksk + (size_t)list[e] * stride;
        if (q0 < stride) r0 -= (uint32_t)row[q0];
        if (q1 < stride) r1 -= (uint32_t)row[q1];
        if (q2 < stride) r2 -= (uint32_t)row[q2];
    }
    Torus32* out = flat_out ? flat_out + (size_t)item * stride : resolve(W, W.item0 + item, stride).out;
    const uint32_t bprime = (uint32_t)u[N];
    if (q0 <= n) out[q0] = (int32_t)(r0 + (q0 == n ? bprime : 0u));
    else if (q0 < stride) out[q0] = 0;
    if (q1 < stride) out[q1] = q1 <= n ? (int32_t)(r1 + (q1 == n ? bprime : 0u)) : 0;
    if (q2 < stride) out[q2] = q2 <= n ? (int32_t)(r2 + (q2 == n ? bprime : 0u)) : 0;
}

// ---- K5, vectorised: one 512-thread workgroup per gate instance ----
// The non-zero digits are compacted into a row list; the 8 waves take list
// entries round-robin, each wave subtracting whole 16-byte-per-lane row pieces
// (NLD dwordx4 loads cover one padded KSK row), four rows in flight per wave;
// the 8 partial sums meet in LDS.  Subtraction mod 2^32 commutes, so neither the
// list order nor the split changes a single bit.
// LDS: u [N+4] | list [N*t] | part [8][stride]
constexpr int kKsThreads = 512;
template <int NLD>
__global__ __launch_bounds__(kKsThreads) void k_keyswitch_vec(DevKeys K, WorkDesc W, const Torus32* ext,
                                                              Torus32* flat_out, int32_t splits) {
    // splits > 1 (launches of a handful of gates, where one workgroup per gate leaves the chip idle and the walk's
    // latency is what counts): blockIdx.y takes coefficients [y*N/splits, (y+1)*N/splits) and ADDS its share to an
    // output row that k_keyswitch_init has set to (0, ..., 0, b); int32 addition commutes, so the bits do not change
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t N = K.N, n = K.n, t = K.ks_t, basebit = K.ks_basebit, stride = K.stride;
    int32_t* u = reinterpret_cast<int32_t*>(smem);
    uint32_t* list = reinterpret_cast<uint32_t*>(u + N + 4);
    int4* part = reinterpret_cast<int4*>(list + (size_t)N * t);
    __shared__ uint32_t s_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t item = (int64_t)blockIdx.x;
    const Torus32* src = ext + (size_t)item * (N + 4);
    if (tid == 0) s_count = 0;
    for (int32_t j = tid; j <= N; j += kKsThreads) u[j] = src[j];
    __syncthreads();
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    const uint32_t mask = (1u << basebit) - 1;
    const int32_t idx0 = splits > 1 ? (int32_t)blockIdx.y * (N / splits) * t : 0;
    const int32_t idx1 = splits > 1 ? idx0 + (N / splits) * t : N * t;
    for (int32_t idx = idx0 + tid; idx < idx1; idx += kKsThreads) {
        const int32_t i = idx / t, j = idx - i * t;
        const uint32_t d = (((uint32_t)u[i] + prec_offset) >> (32 - (j + 1) * basebit)) & mask;
        if (d) list[atomicAdd(&s_count, 1u)] = ((uint32_t)idx << basebit) + d;  // row index [i][j][d]
    }
    __syncthreads();
    const uint32_t cnt = s_count;
    const int32_t nvec = stride >> 2;
    int4 acc[NLD];
#pragma unroll
    for (int v = 0; v < NLD; v++) acc[v] = make_int4(0, 0, 0, 0);
    const int4* kbase = reinterpret_cast<const int4*>(K.ksk);
    uint32_t e = wave;
    for (; e + 24 < cnt; e += 32) {  // four rows (e, e+8, e+16, e+24) in flight
        int4 r[4][NLD];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4* row = kbase + (size_t)list[e + 8 * q] * nvec;
#pragma unroll
            for (int v = 0; v < NLD; v++)
                r[q][v] = (lane + 64 * v < nvec) ? row[lane + 64 * v] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int v = 0; v < NLD; v++) {
                acc[v].x -= r[q][v].x;
                acc[v].y -= r[q][v].y;
                acc[v].z -= r[q][v].z;
                acc[v].w -= r[q][v].w;
            }
    }
    for (; e < cnt; e += 8) {
        const int4* row = kbase + (size_t)list[e] * nvec;
#pragma unroll
        for (int v = 0; v < NLD; v++)
            if (lane + 64 * v < nvec) {
                const int4 rr = row[lane + 64 * v];
                acc[v].x -= rr.x;
                acc[v].y -= rr.y;
                acc[v].z -= rr.z;
                acc[v].w -= rr.w;
            }
    }
#pragma unroll
    for (int v = 0; v < NLD; v++)
        if (lane + 64 * v < nvec) part[(size_t)wave * nvec + lane + 64 * v] = acc[v];
    __syncthreads();
    Torus32* out = flat_out ? flat_out + (size_t)item * stride : resolve(W, W.item0 + item, stride).out;
    const int32_t* parti = reinterpret_cast<const int32_t*>(part);
    for (int32_t q = tid; q < stride; q += kKsThreads) {
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) v += (uint32_t)parti[(size_t)w * stride + q];
        if (splits > 1) {
            if (q <= n) atomicAdd(reinterpret_cast<uint32_t*>(out) + q, v);
            continue;
        }
        if (q == n) v += (uint32_t)u[N];
        out[q] = q <= n ? (int32_t)v : 0;
    }
}

// output rows of a split key switch: (0, ..., 0, b) with b the extracted sample's last word
__global__ __launch_bounds__(256) void k_keyswitch_init(DevKeys K, WorkDesc W, const Torus32* ext, Torus32* flat_out) {
    const int64_t item = (int64_t)blockIdx.x;
    const int32_t n = K.n, stride = K.stride;
    Torus32* out = flat_out ? flat_out + (size_t)item * stride : resolve(W, W.item0 + item, stride).out;
    const Torus32 b = ext[(size_t)item * (K.N + 4) + K.N];
    for (int32_t q = threadIdx.x; q < stride; q += 256) out[q] = q == n ? b : 0;
}

// ---- K5, gate-batched: one workgroup per G gate instances ----
// The key-switch key does not fit the L2s (83 MB), so K5 is bound by how many KSK bytes
// are fetched per gate.  Here a workgroup walks ALL (i, j) positions once, loads the
// three candidate rows [i][j][1..3] and lets each of its G gates subtract the one its
// digit selects: 3 x 2.5 KB x N x t / G bytes per gate instead of ~0.75 x 2.5 KB x N x t.
// One wave per 64 int4 columns of a row (3 waves at n=630), each lane owning one column
// for all G gates, so no partial sums cross waves.  The t digits of a'_i are packed into
// one word per gate, pulled into SGPRs once per i; the digit (wave-uniform) indexes a 4-row
// register table {0, r1, r2, r3} through the SGPR-indexed VGPR mode (s_set_gpr_idx), which is
// 3x cheaper than v_cndmask chains or scalar branches.  Subtraction mod 2^32 commutes, so the
// result is bit-identical to the other kernels.
// LDS: dw [G][N] u16 | bprime [G]
template <int G>
__global__ __launch_bounds__(256) void k_keyswitch_batch(DevKeys K, WorkDesc W, const Torus32* ext, Torus32* flat_out,
                                                         int64_t items) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t N = K.N, n = K.n, t = K.ks_t, basebit = K.ks_basebit, stride = K.stride;
    uint16_t* dw = reinterpret_cast<uint16_t*>(smem);
    int32_t* bprime = reinterpret_cast<int32_t*>(dw + (size_t)G * N);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int64_t item0 = (int64_t)blockIdx.x * G;
    const int32_t gcount = (int32_t)(items - item0 < G ? items - item0 : G);
    const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
    const uint32_t mask = (1u << basebit) - 1;
    // pack the digits of every a'_i of every gate: digit j sits at bits [j*basebit, (j+1)*basebit)
    for (int32_t idx = tid; idx < G * N; idx += nthreads) {
        const int32_t g = idx / N, i = idx - g * N;
        uint32_t packed = 0;
        if (g < gcount) {
            const uint32_t a = (uint32_t)ext[(size_t)(item0 + g) * (N + 4) + i] + prec_offset;
            for (int32_t j = 0; j < t; j++) packed |= ((a >> (32 - (j + 1) * basebit)) & mask) << (j * basebit);
        }
        dw[idx] = (uint16_t)packed;
    }
    if (tid < G) bprime[tid] = tid < gcount ? ext[(size_t)(item0 + tid) * (N + 4) + N] : 0;
    __syncthreads();

    const int32_t nvec = stride >> 2;
    const int32_t col = tid;  // one int4 column per thread
    const bool active = col < nvec;
    const int4* kbase = reinterpret_cast<const int4*>(K.ksk) + (active ? col : 0);  // idle lanes shadow column 0
    const size_t rowpitch = (size_t)nvec;  // int4 per row; rows [pos][d] are consecutive
    int4 acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = make_int4(0, 0, 0, 0);

    // Walk i (the extracted coefficient), then its t digits.  The packed digits of a'_i of all
    // G gates are pulled into SGPRs once per i.  Candidate rows are requested two positions
    // ahead into a ring of three named row sets (the walk is latency-bound otherwise).
    const size_t npos = (size_t)N * t;
#define KS_LOAD(A, B, C, POS)                                        \
    {                                                                \
        size_t pp_ = (POS);                                          \
        if (pp_ >= npos) pp_ = npos - 1;                             \
        const int4* row_ = kbase + pp_ * 4 * rowpitch;               \
        A = row_[1 * rowpitch];                                      \
        B = row_[2 * rowpitch];                                      \
        C = row_[3 * rowpitch];                                      \
    }
#define KS_USE(A, B, C, SH)                                                        \
    {                                                                              \
        const int32_t tab_[16] = {0, 0, 0, 0, A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, C.x, C.y, C.z, C.w}; \
        _Pragma("unroll") for (int g = 0; g < G; g++) {                            \
            const uint32_t d_ = ((dg[g] >> (SH)) & mask) * 4;  /* uniform: SGPR-indexed register read */ \
            acc[g].x -= tab_[d_ + 0];                                              \
            acc[g].y -= tab_[d_ + 1];                                              \
            acc[g].z -= tab_[d_ + 2];                                              \
            acc[g].w -= tab_[d_ + 3];                                              \
        }                                                                          \
    }
    int4 a1, a2, a3, b1, b2, b3, c1, c2, c3;
    KS_LOAD(a1, a2, a3, 0)
    KS_LOAD(b1, b2, b3, 1)
    size_t pos = 0;
    for (int32_t i = 0; i < N; i++) {
        uint32_t dg[G];
#pragma unroll
        for (int g = 0; g < G; g++) dg[g] = __builtin_amdgcn_readfirstlane((uint32_t)dw[g * N + i]);
        int32_t sh = 0;
        for (int32_t j = 0; j < t; j++, pos++, sh += basebit) {
            KS_LOAD(c1, c2, c3, pos + 2)
            KS_USE(a1, a2, a3, sh)
            a1 = b1; a2 = b2; a3 = b3;
            b1 = c1; b2 = c2; b3 = c3;
        }
    }
#undef KS_LOAD
#undef KS_USE
    if (active) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g < gcount) {
                int4 v = acc[g];
                if (col == (n >> 2)) {  // the column holding b'
                    const int32_t bp = bprime[g];
                    switch (n & 3) {
                        case 0: v.x += bp; break;
                        case 1: v.y += bp; break;
                        case 2: v.z += bp; break;
                        default: v.w += bp; break;
                    }
                }
                Torus32* out = flat_out ? flat_out + (size_t)(item0 + g) * stride : resolve(W, W.item0 + item0 + g, stride).out;
                reinterpret_cast<int4*>(out)[col] = v;
            }
        }
    }
}

// bootsMUX: u = (0, 1/8) + u1 + u2 over the extracted samples of the two blind rotations of a gate
// (rows 2g and 2g+1 of `ext`), written to row g of `dst`; the key switch follows on `dst`
__global__ void k_mux_combine(const Torus32* ext, Torus32* dst, int32_t N) {
    const size_t g = blockIdx.x;
    const Torus32* u1 = ext + (2 * g) * (size_t)(N + 4);
    const Torus32* u2 = u1 + (N + 4);
    Torus32* d = dst + g * (size_t)(N + 4);
    for (int32_t j = threadIdx.x; j <= N; j += blockDim.x)
        d[j] = (int32_t)((uint32_t)u1[j] + (uint32_t)u2[j] + (j == N ? (uint32_t)kMU : 0u));
}

// Audit of the one-limb blind rotation: rows of extracted samples it produced against the same gate instances run on the
// two-limb (provably exact) kernel.  One workgroup per row; any differing word counts the row in *mismatches.
// inject: test hook -- row 0 is compared as if its first word differed.
__global__ void k_audit_compare(const Torus32* primary, const Torus32* exact, int32_t N, unsigned* mismatches, int inject) {
    const size_t g = blockIdx.x;
    const Torus32* a = primary + g * (size_t)(N + 4);
    const Torus32* b = exact + g * (size_t)(N + 4);
    int bad = (inject && g == 0 && threadIdx.x == 0) ? 1 : 0;
    for (int32_t j = threadIdx.x; j <= N; j += blockDim.x) bad |= a[j] != b[j];
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicAdd(mismatches, 1u);
}

// outputs of a circuit: out[b][o] = +-store[b][slot] or the constant
__global__ void k_gather_outputs(const OutRef* outs, int32_t n_out, const Torus32* store, int32_t n_slots,
                                 Torus32* out, int64_t batch, int32_t stride, int32_t n) {
    const int64_t row = blockIdx.x;  // b * n_out + o
    const int64_t b = row / n_out;
    const OutRef r = outs[row % n_out];
    Torus32* dst = out + (size_t)row * stride;
    for (int32_t q = threadIdx.x; q < stride; q += blockDim.x) {
        uint32_t v;
        if (r.slot >= 0)
            v = (uint32_t)store[((size_t)b * n_slots + r.slot) * stride + q];
        else
            v = q == n ? 0xE0000000u : 0u;
        if (r.neg) v = 0u - v;
        dst[q] = q <= n ? (int32_t)v : 0;
    }
}

}  // namespace

// ------------------------------------------------------------------------
// One stream's worth of per-launch scratch.  Lane 0 runs on the evaluator's own stream and is all a call uses with
// "overlap" = 0.  Otherwise further lanes -- more streams of the SAME context (one copy of the key), each with its own
// extracted-sample rows, blind-rotation state, key-switch digits and audit scratch -- take either a contiguous share of a
// batch's expressions through every level of a circuit (pipelines: one fork, one join per evaluation) or every other piece
// of a wide level (lane 0 then waits for lane 1 before the next level starts).  See Evaluator::set_option in evaluator.h.
constexpr int kMaxLanes = 4;
struct Lane {
    hipStream_t stream = nullptr;
    Torus32* ext = nullptr;
    size_t ext_items = 0;
    void* br_state = nullptr;  // sliced blind rotation: accumulators + rotation amounts
    size_t br_state_items = 0;
    void* ks_digits = nullptr;
    size_t ks_digits_bytes = 0;
    Torus32* audit_ext = nullptr;
    void* audit_state = nullptr;
};

struct Evaluator::Impl {
    Params p;
    Lane lane[kMaxLanes];
    hipEvent_t ev_fork = nullptr, ev_join[kMaxLanes] = {};  // ev_join[k]: lane k's share of a level / of an evaluation is queued
    // "overlap": 1 = levels of at least overlap_min gate instances are cut in two and issued on two streams (the tail of one
    // piece's launches fills with the other's workgroups, a piece's key switch runs under the next piece's rotation); 0 = one stream
    int32_t overlap = 1;
    int64_t overlap_min = 0;     // set in init(): 16 gates per CU -- each half is then a full round of resident gates
    int64_t overlapped_levels = 0;
    // Circuits over a batch: the batch is cut into two contiguous halves of EXPRESSIONS and each half runs through every
    // level on its own stream -- expressions are independent, so the two pipelines never wait for each other between
    // levels (one fork after the input copy, one join before the outputs are gathered).  Used when the circuit's mean
    // level holds at least pipe_min gate instances over the whole batch.  While both pipelines run, a launch shares the
    // chip with the other stream's launch of the same level: kernels are chosen by the gates in flight on BOTH streams.
    int64_t pipe_min = 0;        // set in init(): 8 gates per CU (measured: 11 per CU +2.9 %, 4 per CU -10 %, profiles/r5_overlap_ab.txt)
    int32_t pipe_lanes = 2;      // pipelines a qualifying evaluation is cut into (2 .. kMaxLanes; "pipe_lanes", IEACHE_PIPE_LANES)
    int32_t concurrency = 1;     // streams issuing launches side by side right now (kernel choice is by cnt x concurrency)
    int64_t pipelined_evals = 0;
    // "pipe_auto" (default 1): with a mean level between pipe_min / 8 and 2 x pipe_min neither stream mode wins everywhere
    // (mul32 x 40: pipelines +5.6 %, muladd64 x 16: -9.7 %, profiles/r5_pipes_vs_mix.txt), so the first four evaluations of a
    // (circuit, batch) there alternate -- without pipelines, with, without, with -- and later ones take whichever mode had the
    // faster evaluation.  Every one of them is a complete evaluation with the same output bits; only the schedule differs.
    int32_t pipe_auto = 1;
    struct Tuned {
        double ms[2] = {-1.0, -1.0};  // best wall time of an evaluation without / with pipelines
        int n[2] = {0, 0};            // trials so far (two each, alternating: a first call also pays for allocations)
    };
    std::map<std::tuple<size_t, int32_t, size_t, size_t, bool>, Tuned> tuned;  // (gates, levels, outputs, batch, exact_fft)
    int64_t tuned_evals = 0;
    // device rows the host-buffer entry points stage their operands and results in: kept between calls, grown on demand
    Torus32* stage[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[4] = {0, 0, 0, 0};
    int64_t stage_allocs = 0;
    DevKeys K{};
    double2* bkf = nullptr;
    double2* bkf_w64 = nullptr;  // spectrum in the wave-per-gate kernel's layout
    double2* tw_w64 = nullptr;   // its twiddle table
    double2* bkf1_w64 = nullptr; // one-limb spectrum of k_blind_rotate_w1
    unsigned* fft_guard = nullptr;  // [0] launches whose rounding deviation exceeded the limit, [1] max deviation (float bits), [2] audit rows that differed
    bool exact_fft = false;      // "exact_fft": never use the one-limb kernel
    bool exact_once = false;     // set while a call is repeated after a guard trip
    int64_t one_limb_min = 0;    // launches of at least this many gate instances use the one-limb kernels
    int64_t four_wave_max = 0;   // ... the four-waves-per-gate one (k_blind_rotate_w4r) up to this many (2 per CU),
    int64_t two_wave_max = 0;    // ... the two-waves-per-gate one up to this many (4 per CU: all resident at once), the one-wave one above
    double guard_max = 0;        // largest rounding deviation seen by the one-limb kernel (of 0.5)
    int64_t guard_reruns = 0;    // calls repeated on the two-limb kernel
    // "fft_audit" = K: every K-th (level, chunk) launch that took a one-limb kernel has a sample of kAuditGates of its gate
    // instances run again on the two-limb kernel and compared word for word (fft_guard[2] counts differing rows); 0 = off
    int32_t fft_audit = 64;
    int64_t exact_one_wave_min = 1025;  // two-limb launches from this size on take k_blind_rotate_x1 (one wave per gate)
    int64_t audit_seq = 0;       // one-limb (level, chunk) launches so far
    int64_t audits = 0, audit_gates = 0, audit_mismatches = 0;
    bool audit_inject = false;   // test hook: the next audit reports a mismatch
    int cus = 0;
    bool use_w64 = false;
    bool force_generic_ks = false;
    int32_t* ksk = nullptr;
    double2* twist = nullptr;
    double2* wtab = nullptr;
    Torus32* ext_mux = nullptr;  // bootsMUX: combined extracted samples, chunk/2 rows
    size_t ext_mux_items = 0;
    size_t chunk = 65536;         // gate instances per launch at most (scratch grows on demand, see grown())
    Torus32* store = nullptr;
    size_t store_bytes = 0;
    DevGate* d_gates = nullptr;
    size_t d_gates_cap = 0;
    OutRef* d_outs = nullptr;
    size_t d_outs_cap = 0;
    size_t br_lds = 0, ks_lds = 0, ksv_lds = 0;
    int ks_nld = 0;  // dwordx4 loads per KSK row per wave; 0 = use the scalar kernel
    bool ks_batch_ok = false;     // gate-batched key switch usable (base == 4, digits fit 16 bits, columns fit 8 waves)
    int64_t ks_batch_min = 4096;  // use it from this many gate instances per launch (one workgroup walk takes ~5 ms)
    bool ks_sliced_ok = false;    // hand-scheduled sliced variant of it usable (t = 8, basebit = 2)
    int64_t ks_sliced_min = 576;  // ... and used from this many gate instances per launch (measured crossover with the per-gate kernel: ~560)
    int32_t ks_slice = 0;         // coefficients per launch of the sliced key switch; 0 = the whole walk
    int32_t ks_gates = 0;         // gate instances per workgroup there (8 / 16 / 32); 0 = by launch size
    // key switch as an int8 product on the MFMA pipe (keyswitch_mfma.hip): byte-limb form of the key, digit scratch, and the
    // launch size from which it takes over from the hand-scheduled walk
    int8_t* ks_limbs = nullptr;
    bool ks_mfma_ok = false;
    int64_t ks_mfma_min = 64;     // measured crossover with the per-gate walk: ~40 gates (0.08 ms either way)
    int32_t ks_mfma_split = 0;    // K split of the product; 0 = by launch size
    int32_t ks_split_max = 16;    // per-gate key switch: workgroups one gate's walk may be cut into when the launch is tiny
    int32_t br_slice = 0;         // CMux steps per blind-rotation launch; 0 = the kernel's default
    int32_t br_variant = w64::default_variant();
    // gate instances per workgroup of the one-wave-per-gate kernels (k_blind_rotate_w1b / _x1): 1 .. 4, 0 = by launch size
    // launches of 4 .. 7 and 8 .. 10.5 gates per CU (mix_plan.h): rotation of roles between the two-waves- and the
    // one-wave-per-gate kernel on three streams (w64::MixPlan); "br_mix" 0/1, "mix_s1" steps of a one-wave turn,
    // "mix_ratio" = 100 x (two-wave steps per one-wave step), "mix_sync" phase barriers, "mix_k" / "mix_tw" a forced geometry
    int32_t br_mix = 1, mix_s1 = 16, mix_ratio = 200, mix_sync = 0, mix_wg = 2, mix_k = 0, mix_tw = 0;
    int64_t mixed_launches = 0;
    bool level_on_two_lanes = false;  // set while a level's halves are being queued on two streams (no rotation of roles then)
    hipEvent_t ev_mix[kMaxLanes] = {};
    int32_t wg_gates = 0;
    int64_t wg3_max = 0;          // set in init(): launches of up to this many gate instances (6 per CU) take three per workgroup
    // launches of at most this many gate instances (one per CU) use the 2L-waves-per-gate kernel in a
    // single launch: what matters there is the latency of one blind rotation, not throughput
    int64_t br_wide_max = 0;
};

Evaluator::Evaluator(const Params& p, int device) : p_(p), device_(device), d_(new Impl) {
    try {
        init();
    } catch (...) {
        destroy();  // the destructor does not run for a half-built object
        throw;
    }
}

void Evaluator::init() {
    const Params& p = p_;
    const int device = device_;
    if (!p.supported()) throw std::invalid_argument("unsupported TFHE parameter set");
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) throw std::runtime_error("no such HIP device");
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    d_->lane[0].stream = stream_;
    {
        int cus = 0;
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        d_->cus = cus;
        d_->br_wide_max = cus;  // one workgroup of the wide kernel fills a CU
        d_->overlap_min = 16 * (int64_t)cus;
        d_->pipe_min = 8 * (int64_t)cus;
        if (const char* e = getenv("IEACHE_PIPE_MIN")) d_->pipe_min = atoll(e);
        if (const char* e = getenv("IEACHE_PIPE_LANES")) d_->pipe_lanes = atoi(e) >= 2 && atoi(e) <= kMaxLanes ? atoi(e) : 2;
        d_->wg3_max = 6 * (int64_t)cus;
        if (const char* e = getenv("IEACHE_BR_MIX")) d_->br_mix = atoi(e) != 0;
        if (const char* e = getenv("IEACHE_WG_GATES")) d_->wg_gates = atoi(e) >= 0 && atoi(e) <= 4 ? atoi(e) : 0;
        if (const char* e = getenv("IEACHE_WG3_MAX")) d_->wg3_max = atoll(e);
        if (const char* e = getenv("IEACHE_OVERLAP")) d_->overlap = atoi(e) != 0;
        if (const char* e = getenv("IEACHE_OVERLAP_MIN")) d_->overlap_min = atoll(e);
        // k_blind_rotate_w1: one wave per gate, 256 VGPRs -> 2 per SIMD = 8 gates per CU
        // ("exact_fft": k_blind_rotate_x1 holds 8 gates per CU too; k_blind_rotate_w2, 2 waves per gate and 35.8 KB of LDS, 4 per CU)
        resident_gates_ = 8 * cus;
        d_->one_limb_min = cus + 1;  // everything the latency kernel does not take
        d_->four_wave_max = 2 * cus;
        d_->two_wave_max = 5 * cus;  // measured crossover with one wave per gate: 1 216 gates 8.6 against 10.0 ms, 1 400 gates 10.8 against 10.1
        if (const char* e = getenv("IEACHE_TWO_WAVE_MAX")) d_->two_wave_max = atoll(e);
        if (const char* e = getenv("IEACHE_BR_WIDE_MAX")) d_->br_wide_max = atoll(e);
        if (const char* e = getenv("IEACHE_ONE_LIMB_MIN")) d_->one_limb_min = atoll(e);
        if (const char* e = getenv("IEACHE_EXACT_FFT")) d_->exact_fft = atoi(e) != 0;
        if (const char* e = getenv("IEACHE_FFT_AUDIT")) d_->fft_audit = atoi(e) > 0 ? atoi(e) : 0;
        if (!w64::one_limb_supported(p)) d_->exact_fft = true;
        resident_two_wave_ = 4 * cus;
        d_->exact_one_wave_min = 4 * cus + 1;  // two-limb launches that do not fit the two-waves-per-gate kernel's 4 gates per CU
        if (const char* e = getenv("IEACHE_EXACT_ONE_WAVE_MIN")) d_->exact_one_wave_min = atoll(e);
        if (d_->exact_fft) resident_two_wave_ = 0;  // k_blind_rotate_x1 holds 8 gates per CU, as k_blind_rotate_w1b does
    }
    d_->p = p;
    DevKeys& K = d_->K;
    K.n = p.n;
    K.N = p.N;
    K.M = p.N / 2;
    K.logM = 0;
    while ((1 << K.logM) < K.M) K.logM++;
    K.l = p.l;
    K.Bgbit = p.Bgbit;
    K.kpl = p.kpl();
    K.ks_t = p.ks_t;
    K.ks_basebit = p.ks_basebit;
    K.ks_base = p.ks_base();
    K.stride = p.lwe_stride();
    K.dec_offset = 0;
    for (int32_t i = 1; i <= p.l; i++) K.dec_offset += (1u << (p.Bgbit - 1)) << (32 - i * p.Bgbit);
    // twiddles, computed once in double precision on the host
    const int32_t M = K.M;
    std::vector<double2> tw(M), w(M / 2 > 0 ? M / 2 : 1);
    for (int32_t j = 0; j < M; j++) tw[j] = make_double2(std::cos(M_PI * j / p.N), std::sin(M_PI * j / p.N));
    for (int32_t j = 0; j < M / 2; j++)
        w[j] = make_double2(std::cos(-2.0 * M_PI * j / M), std::sin(-2.0 * M_PI * j / M));
    HIP_CHECK(hipMalloc(&d_->twist, sizeof(double2) * tw.size()));
    HIP_CHECK(hipMalloc(&d_->wtab, sizeof(double2) * w.size()));
    HIP_CHECK(hipMemcpy(d_->twist, tw.data(), sizeof(double2) * tw.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_->wtab, w.data(), sizeof(double2) * w.size(), hipMemcpyHostToDevice));
    K.twist = d_->twist;
    K.wtab = d_->wtab;
    const int32_t frows = K.kpl > 4 ? K.kpl : 4;
    d_->br_lds = (size_t)frows * M * sizeof(double2) + (size_t)2 * p.N * 4 + (((size_t)p.n * 2 + 15) & ~(size_t)15);
    d_->ks_lds = (size_t)(p.N + 4) * 4 + (size_t)p.N * p.ks_t * 4;
    if (d_->br_lds > 160 * 1024 || d_->ks_lds > 160 * 1024)
        throw std::invalid_argument("parameter set exceeds the 160 KiB LDS of a CU");
    HIP_CHECK(hipFuncSetAttribute((const void*)k_blind_rotate_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)d_->br_lds));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_keyswitch_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)d_->ks_lds));
    {
        const int nvec = K.stride / 4, nld = (nvec + 63) / 64;
        d_->ksv_lds = (size_t)(p.N + 4) * 4 + (size_t)p.N * p.ks_t * 4 + (size_t)8 * K.stride * 4;
        d_->ks_batch_ok = K.ks_base == 4 && p.ks_t * p.ks_basebit <= 16 && p.ks_t % 4 == 0 && nld <= 4;
        if (const char* e = getenv("IEACHE_KS_BATCH_MIN")) d_->ks_batch_min = atoll(e);
        d_->ks_sliced_ok = kss::supported(p) && nld <= 4;
        if (const char* e = getenv("IEACHE_KS_SLICED_MIN")) d_->ks_sliced_min = atoll(e);
        d_->ks_mfma_ok = ksm::supported(p);
        if (const char* e = getenv("IEACHE_KS_MFMA_MIN")) d_->ks_mfma_min = atoll(e);
        if (d_->ks_batch_ok)
            HIP_CHECK(hipFuncSetAttribute((const void*)k_keyswitch_batch<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)((size_t)16 * p.N * 2 + 64)));
        if (nld <= 4 && d_->ksv_lds <= 160 * 1024) {
            d_->ks_nld = nld;
            const void* f = nld == 1 ? (const void*)k_keyswitch_vec<1> : nld == 2 ? (const void*)k_keyswitch_vec<2>
                          : nld == 3 ? (const void*)k_keyswitch_vec<3> : (const void*)k_keyswitch_vec<4>;
            HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)d_->ksv_lds));
        }
    }
}

Evaluator::~Evaluator() { destroy(); }

void Evaluator::destroy() {
    if (!d_) return;
    (void)hipSetDevice(device_);
    for (int k = 1; k < kMaxLanes; k++)
        if (d_->lane[k].stream) (void)hipStreamSynchronize(d_->lane[k].stream);
    if (stream_) (void)hipStreamSynchronize(stream_);
    (void)hipFree(d_->bkf);
    (void)hipFree(d_->bkf_w64);
    (void)hipFree(d_->tw_w64);
    (void)hipFree(d_->bkf1_w64);
    (void)hipFree(d_->fft_guard);
    (void)hipFree(d_->ksk);
    (void)hipFree(d_->ks_limbs);
    (void)hipFree(d_->twist);
    (void)hipFree(d_->wtab);
    for (Lane& ln : d_->lane) {
        (void)hipFree(ln.ext);
        (void)hipFree(ln.br_state);
        (void)hipFree(ln.ks_digits);
        (void)hipFree(ln.audit_ext);
        (void)hipFree(ln.audit_state);
    }
    (void)hipFree(d_->ext_mux);
    for (Torus32* st : d_->stage) (void)hipFree(st);
    if (d_->ev_fork) (void)hipEventDestroy(d_->ev_fork);
    for (int k = 0; k < kMaxLanes; k++)
        if (d_->ev_mix[k]) (void)hipEventDestroy(d_->ev_mix[k]);
    for (int k = 1; k < kMaxLanes; k++) {
        if (d_->ev_join[k]) (void)hipEventDestroy(d_->ev_join[k]);
        if (d_->lane[k].stream) (void)hipStreamDestroy(d_->lane[k].stream);
    }
    (void)hipFree(d_->store);
    (void)hipFree(d_->d_gates);
    (void)hipFree(d_->d_outs);
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
    delete d_;
    d_ = nullptr;
}

void Evaluator::wait_for_stream(hipStream_t producer) {
    HIP_CHECK(hipSetDevice(device_));
    hipEvent_t ev;
    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, producer);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream_, ev, 0);
    (void)hipEventDestroy(ev);
    HIP_CHECK(e);
}

// Staging rows for the host-buffer entry points (slot 0 .. 2: operands, 3: results).  A slot grows to at least `bytes`
// (doubling, so a run of growing batches does not reallocate every call) and is zeroed when it is (re)allocated: callers
// upload n + 1 words per row of lwe_stride() and rely on the padding words of OPERAND rows being zero, which holds because
// nothing but such uploads ever writes slots 0 .. 2.  Every hipMalloc / hipFree is a device-wide synchronisation, which is
// why a warm daemon request must not make one.
Torus32* Evaluator::staging(int slot, size_t bytes) {
    if (slot < 0 || slot >= 4) throw std::invalid_argument("staging slot");
    HIP_CHECK(hipSetDevice(device_));
    bytes = (bytes + 255) & ~(size_t)255;
    if (d_->stage_bytes[slot] < bytes || !d_->stage[slot]) {
        const size_t want = std::max(bytes, std::min<size_t>(2 * d_->stage_bytes[slot], (size_t)1 << 30));
        if (d_->stage[slot]) HIP_CHECK(hipFree(d_->stage[slot]));
        d_->stage[slot] = nullptr;
        d_->stage_bytes[slot] = 0;
        HIP_CHECK(hipMalloc(&d_->stage[slot], want + 16));
        HIP_CHECK(hipMemset(d_->stage[slot], 0, want + 16));
        d_->stage_bytes[slot] = want;
        d_->stage_allocs++;
    }
    return d_->stage[slot];
}

void Evaluator::set_chunk(size_t items) {
    if (items < 1) items = 1;
    d_->chunk = items;
}

bool Evaluator::set_option(const std::string& name, int64_t value) {
    if (name == "chunk" && value >= 1) {
        set_chunk((size_t)value);
    } else if (name == "force_generic") {
        force_generic_ = value != 0;
    } else if (name == "ks_batch_min" && value >= 0) {
        d_->ks_batch_min = value;
    } else if (name == "ks_sliced_min" && value >= 0) {
        d_->ks_sliced_min = value;
    } else if (name == "ks_slice" && value >= 0 && value <= kss::max_slice()) {
        d_->ks_slice = (int32_t)value;
    } else if (name == "ks_gates" && (value == 0 || value == 4 || value == 8 || value == 16 || value == 32)) {
        d_->ks_gates = (int32_t)value;
    } else if (name == "ks_mfma_min" && value >= 0) {
        d_->ks_mfma_min = value;
    } else if (name == "ks_mfma_split" && value >= 0 && value <= 64 && (value & (value - 1)) == 0 &&
               (value == 0 || ksm::split_ok(p_, (int32_t)value))) {  // 0 = by launch size; else a power of two every split of which holds whole loop trips
        d_->ks_mfma_split = (int32_t)value;
    } else if (name == "ks_split_max" && value >= 1 && value <= 64) {
        d_->ks_split_max = (int32_t)value;
    } else if (name == "br_mix" && (value == 0 || value == 1)) {
        d_->br_mix = (int32_t)value;
    } else if (name == "mix_s1" && value >= 1 && value <= 630) {
        d_->mix_s1 = (int32_t)value;
    } else if (name == "mix_ratio" && value >= 100 && value <= 400) {
        d_->mix_ratio = (int32_t)value;
    } else if (name == "mix_k" && value >= 0 && value <= kMaxLanes && value != 1) {
        d_->mix_k = (int32_t)value;
    } else if (name == "mix_tw" && value >= 0 && value < kMaxLanes) {
        d_->mix_tw = (int32_t)value;
    } else if (name == "mix_wg" && value >= 1 && value <= 4) {
        d_->mix_wg = (int32_t)value;
    } else if (name == "mix_sync" && (value == 0 || value == 1)) {
        d_->mix_sync = (int32_t)value;
    } else if (name == "wg_gates" && value >= 0 && value <= 4) {
        d_->wg_gates = (int32_t)value;
    } else if (name == "wg3_max" && value >= 0) {
        d_->wg3_max = value;
    } else if (name == "overlap" && (value == 0 || value == 1)) {
        d_->overlap = (int32_t)value;
    } else if (name == "overlap_min" && value >= 2) {
        d_->overlap_min = value;
    } else if (name == "pipe_min" && value >= 0) {
        d_->pipe_min = value;
    } else if (name == "pipe_auto" && (value == 0 || value == 1)) {
        d_->pipe_auto = (int32_t)value;
    } else if (name == "pipe_lanes" && value >= 2 && value <= kMaxLanes) {
        d_->pipe_lanes = (int32_t)value;
    } else if (name == "br_wide_max" && value >= 0) {
        d_->br_wide_max = value;
    } else if (name == "br_slice" && value >= 0 && value <= 4096) {  // 0 = by kernel and launch size
        d_->br_slice = (int32_t)value;
    } else if (name == "br_variant" && value >= 0 && value <= 1000 && w64::variant_known((int32_t)value)) {
        d_->br_variant = (int32_t)value;
    } else if (name == "exact_fft" && (value == 1 || (value == 0 && w64::one_limb_supported(p_)))) {
        d_->exact_fft = value != 0;
        resident_two_wave_ = d_->exact_fft ? 0 : 4 * d_->cus;
    } else if (name == "one_limb_min" && value >= 0) {
        d_->one_limb_min = value;
    } else if (name == "exact_one_wave_min" && value >= 0) {
        d_->exact_one_wave_min = value;
    } else if (name == "two_wave_max" && value >= 0) {
        d_->two_wave_max = value;
    } else if (name == "four_wave_max" && value >= 0) {
        d_->four_wave_max = value;

    } else if (name == "fft_audit" && value >= 0 && value <= (1 << 30)) {
        d_->fft_audit = (int32_t)value;
    } else if (name == "fft_audit_inject" && value == 1) {
        d_->audit_inject = true;
    } else if (name == "fft_guard_inject" && value == 1 && d_->fft_guard) {
        // test hook: the next call finds the guard tripped and repeats itself on the two-limb kernel
        const unsigned one = 1;
        HIP_CHECK(hipSetDevice(device_));
        HIP_CHECK(hipMemcpy(d_->fft_guard, &one, sizeof one, hipMemcpyHostToDevice));
    } else {
        return false;
    }
    d_->tuned.clear();  // whatever was timed was timed under the old options
    return true;
}

bool Evaluator::get_option(const std::string& name, int64_t* value) const {
    int64_t v;
    if (name == "overlap") v = d_->overlap;
    else if (name == "overlap_min") v = d_->overlap_min;
    else if (name == "overlapped_levels") v = d_->overlapped_levels;  // levels issued on two streams so far (a counter)
    else if (name == "pipe_min") v = d_->pipe_min;
    else if (name == "pipe_lanes") v = d_->pipe_lanes;
    else if (name == "pipe_auto") v = d_->pipe_auto;
    else if (name == "tuned_evals") v = d_->tuned_evals;  // evaluations that were one of the two timed trials of a (circuit, batch)
    else if (name == "pipelined_evals") v = d_->pipelined_evals;      // circuit evaluations run as two expression-half pipelines so far
    else if (name == "br_mix") v = d_->br_mix;
    else if (name == "mix_s1") v = d_->mix_s1;
    else if (name == "mix_ratio") v = d_->mix_ratio;
    else if (name == "mix_sync") v = d_->mix_sync;
    else if (name == "mix_wg") v = d_->mix_wg;
    else if (name == "mix_k") v = d_->mix_k;
    else if (name == "mix_tw") v = d_->mix_tw;
    else if (name == "mixed_launches") v = d_->mixed_launches;  // (level, piece) launches run as a rotation of roles so far
    else if (name == "wg_gates") v = d_->wg_gates;
    else if (name == "wg3_max") v = d_->wg3_max;
    else if (name == "staging_allocations") v = d_->stage_allocs;  // (re)allocations of the host entry points' staging rows so far
    else if (name == "cus") v = d_->cus;
    else if (name == "chunk") v = (int64_t)d_->chunk;
    else if (name == "resident_gates") v = resident_gates_;
    else if (name == "exact_fft") v = d_->exact_fft ? 1 : 0;
    else if (name == "exact_one_wave_min") v = d_->exact_one_wave_min;
    else if (name == "one_limb_min") v = d_->one_limb_min;
    else if (name == "two_wave_max") v = d_->two_wave_max;
    else if (name == "four_wave_max") v = d_->four_wave_max;
    else if (name == "br_wide_max") v = d_->br_wide_max;
    else if (name == "br_variant") v = d_->br_variant;
    else if (name == "br_slice") v = d_->br_slice;
    else if (name == "fft_audit") v = d_->fft_audit;
    else if (name == "ks_mfma_min") v = d_->ks_mfma_min;
    else if (name == "ks_mfma_split") v = d_->ks_mfma_split;
    else return false;
    if (value) *value = v;
    return true;
}

std::string Evaluator::kernel_variant() const {
    if (!w64::supported(p_) || force_generic_) return "generic-radix2";
    // the kernel wide launches take: one wave per gate, on the one-limb spectrum or ("exact_fft") on the two-limb one
    return d_->exact_fft ? "x1x64-radix8-twolimb" : "w1x64-radix8-onelimb";
}

void Evaluator::load_keys_host(const Torus32* bk, const Torus32* ksk) {
    HIP_CHECK(hipSetDevice(device_));
    Torus32 *d_bk = nullptr, *d_ksk = nullptr;
    HIP_CHECK(hipMalloc(&d_bk, p_.bk_count() * 4));
    HIP_CHECK(hipMalloc(&d_ksk, p_.ksk_count() * 4));
    HIP_CHECK(hipMemcpy(d_bk, bk, p_.bk_count() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_ksk, ksk, p_.ksk_count() * 4, hipMemcpyHostToDevice));
    try {
        load_keys_device(d_bk, d_ksk);
    } catch (...) {
        (void)hipFree(d_bk);
        (void)hipFree(d_ksk);
        throw;
    }
    HIP_CHECK(hipFree(d_bk));
    HIP_CHECK(hipFree(d_ksk));
}

void Evaluator::load_keys_device(const Torus32* d_bk, const Torus32* d_ksk) {
    HIP_CHECK(hipSetDevice(device_));
    DevKeys& K = d_->K;
    const size_t npoly = (size_t)p_.n * K.kpl * 2;
    if (!d_->bkf) HIP_CHECK(hipMalloc(&d_->bkf, npoly * 2 * K.M * sizeof(double2)));
    const size_t ks_rows = (size_t)p_.k * p_.N * p_.ks_t * K.ks_base;
    if (!d_->ksk) {
        // 16 rows of slack: the sliced key switch prefetches four positions past the end of its walk
        HIP_CHECK(hipMalloc(&d_->ksk, (ks_rows + 16) * K.stride * 4));
        HIP_CHECK(hipMemsetAsync(d_->ksk + ks_rows * K.stride, 0, (size_t)16 * K.stride * 4, stream_));
    }
    K.bkf = d_->bkf;
    K.ksk = d_->ksk;
    hipLaunchKernelGGL(k_bk_to_spectrum, dim3((unsigned)npoly), dim3(kThreads), 2 * K.M * sizeof(double2), stream_, K,
                       d_bk, d_->bkf);
    HIP_CHECK(hipGetLastError());
    if (w64::supported(p_)) {
        if (!d_->bkf_w64) HIP_CHECK(hipMalloc(&d_->bkf_w64, w64::spectrum_elems(p_) * sizeof(double2)));
        if (!d_->tw_w64) {
            HIP_CHECK(hipMalloc(&d_->tw_w64, w64::twiddle_table_elems() * sizeof(double2)));
            w64::build_twiddle_table(d_->tw_w64, stream_);
            HIP_CHECK(hipGetLastError());
        }
        w64::prepare_spectrum(p_, d_bk, d_->bkf_w64, stream_);
        HIP_CHECK(hipGetLastError());
        if (!d_->bkf1_w64) HIP_CHECK(hipMalloc(&d_->bkf1_w64, w64::spectrum1_elems(p_) * sizeof(double2)));
        if (!d_->fft_guard) {
            HIP_CHECK(hipMalloc(&d_->fft_guard, 4 * sizeof(unsigned)));
            HIP_CHECK(hipMemsetAsync(d_->fft_guard, 0, 4 * sizeof(unsigned), stream_));
        }
        w64::prepare_spectrum1(p_, d_bk, d_->bkf1_w64, stream_);
        HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_pad_rows, dim3(2048), dim3(256), 0, stream_, d_ksk, d_->ksk, (int64_t)ks_rows, p_.n + 1,
                       K.stride);
    HIP_CHECK(hipGetLastError());
    if (d_->ks_mfma_ok) {
        if (!d_->ks_limbs) HIP_CHECK(hipMalloc(&d_->ks_limbs, ksm::limb_matrix_bytes(p_)));
        ksm::prepare(p_, d_->ksk, d_->ks_limbs, stream_);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    keys_loaded_ = true;
}

namespace {
struct Timer {
    std::vector<hipEvent_t> ev;  // pairs
    bool on;
    hipStream_t s;
    Timer(bool enabled, hipStream_t st) : on(enabled), s(st) {}
    ~Timer() {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
    void mark() { mark(s); }
    // a pair of marks brackets launches on ONE stream; pairs may come from different streams (overlapped levels)
    void mark(hipStream_t on_stream) {
        if (!on) return;
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        HIP_CHECK(hipEventRecord(e, on_stream));
        ev.push_back(e);
    }
    // Time during which at least one bracketed interval was open.  On one stream the intervals follow each other and this
    // is their sum; intervals of two streams (overlapped levels) run side by side and are merged on the common device
    // timeline (offsets from the first event), so the figure stays "time the chip spent in these launches".
    double sum_ms() {
        std::vector<std::pair<double, double>> iv;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float t0 = 0, dt = 0;
            if (i) HIP_CHECK(hipEventElapsedTime(&t0, ev[0], ev[i]));
            HIP_CHECK(hipEventElapsedTime(&dt, ev[i], ev[i + 1]));
            iv.emplace_back((double)t0, (double)t0 + (double)dt);
        }
        std::sort(iv.begin(), iv.end());
        double tot = 0, lo = 0, hi = -1;
        for (const auto& x : iv) {
            if (hi < lo || x.first > hi) {
                if (hi >= lo) tot += hi - lo;
                lo = x.first;
                hi = x.second;
            } else if (x.second > hi) {
                hi = x.second;
            }
        }
        if (hi >= lo) tot += hi - lo;
        return tot;
    }
};
}  // namespace

// Which blind-rotation kernel a launch of `cnt` gate instances takes (br_variant 0 = by launch size: the 2L-waves-per-gate
// kernel for a handful of gates, two waves per gate on the one-limb spectrum while every gate is resident at once, one wave
// per gate above; "exact_fft" / a repeat after a guard trip: the two-limb kernels).
static void pick_br_variant(const Params& p, const Evaluator::Impl* d, int64_t cnt, int32_t* variant_out, int32_t* slice_out) {
    int32_t variant = d->br_variant, slice = d->br_slice;
    cnt *= d->concurrency;  // the other stream's launch of the same level shares the chip: choose by the gates in flight
    if (variant == 0) {
        if (cnt <= d->br_wide_max) {
            // the latency kernel, on the one-limb spectrum unless exactness by construction is asked for
            variant = (d->exact_fft || d->exact_once) ? w64::kVariantWide : w64::kVariantWideHandoverOneLimb;
            slice = w64::bara_stride(p);
        } else if (!d->exact_fft && !d->exact_once && cnt >= d->one_limb_min) {
            // one to two gates per CU: four waves per gate (two waves per SIMD); while every gate fits a two-wave slot, two
            // waves per gate finish a step sooner than one
            variant = cnt <= d->four_wave_max ? w64::kVariantOneLimbFourWaves
                      : cnt <= d->two_wave_max ? w64::kVariantOneLimbTwoWaves : w64::kVariantOneLimbDefault;
            // every gate of such a launch is resident at once, so nothing is gained from short slices (they keep the rounds
            // of a WIDE launch on the same BK blocks) and each launch boundary costs a tail and a reload of the accumulators:
            // the whole rotation in one launch for four waves per gate, 64 steps for two (interleaved A/B, profiles/r3_slice_ab.txt)
            // (likewise one wave per gate while the launch is a single round of 8 gates per CU)
            if (slice <= 0)
                slice = variant == w64::kVariantOneLimbFourWaves ? w64::bara_stride(p)
                        : (variant == w64::kVariantOneLimbTwoWaves || cnt <= 8 * (int64_t)d->cus) ? 64 : slice;
        }
        else if (cnt >= d->exact_one_wave_min) {
            variant = w64::kVariantExactOneWave;  // "exact_fft" / a repeat: the two-limb product, one wave per gate
            if (slice <= 0 && cnt <= 8 * (int64_t)d->cus) slice = 64;  // a single round of resident gates: as above
        }
    } else if (d->exact_once && w64::variant_one_limb(variant)) {
        variant = cnt >= d->exact_one_wave_min ? w64::kVariantExactOneWave : 0;
    }
    *variant_out = variant;
    *slice_out = slice;
}

// Gate instances per workgroup of the one-wave-per-gate kernels.  Four share a workgroup (for the twiddle table only) and
// two such workgroups fill a CU; a launch of at most six gates per CU in fours leaves half the CUs with two workgroups and
// half with one, in threes every CU gets the same six waves.
static int pick_wg_gates(const Evaluator::Impl* d, int64_t cnt) {
    if (d->wg_gates) return d->wg_gates;
    return cnt * d->concurrency <= d->wg3_max ? 3 : 4;
}

static bool mix_geometry(const Evaluator::Impl* d, int64_t cnt, int32_t variant, int* k_out, int* tw_out);

std::string Evaluator::kernel_for_launch(int64_t gates) const {
    if (!d_->use_w64) return "k_blind_rotate_generic";
    int32_t variant, slice;
    pick_br_variant(p_, d_, gates < 1 ? 1 : gates, &variant, &slice);
    char tag[96];
    snprintf(tag, sizeof tag, "<%d,%d>", (int)p_.l, (int)p_.Bgbit);
    std::string name;
    int mk = 0, mtw = 0;
    if (mix_geometry(d_, gates < 1 ? 1 : gates, variant, &mk, &mtw)) {
        // a rotation of roles between the two kernels (w64::MixPlan): mtw of mk subsets on two waves at a time
        snprintf(tag, sizeof tag, "<%d,%d> %d of %d subsets on two waves", (int)p_.l, (int)p_.Bgbit, mtw, mk);
        return std::string("k_blind_rotate_w2r+w1b") + tag;
    }
    switch (variant) {
        case w64::kVariantOneLimbDefault: name = "k_blind_rotate_w1b"; break;
        case w64::kVariantOneLimbTwoWaves: name = "k_blind_rotate_w2r"; break;
        case w64::kVariantOneLimbFourWaves: name = "k_blind_rotate_w4r"; break;
        case w64::kVariantWideHandoverOneLimb: name = "k_blind_rotate_wide4"; break;
        case w64::kVariantWide: name = "k_blind_rotate_wide"; break;
        case w64::kVariantExactOneWave: name = "k_blind_rotate_x1"; break;
        case 0: name = "k_blind_rotate_w2"; break;
        default: snprintf(tag, sizeof tag, "<%d,%d> br_variant %d", (int)p_.l, (int)p_.Bgbit, (int)variant); name = "k_blind_rotate"; break;
    }
    return name + tag;
}

// Scratch sized by what calls have needed so far, not by the largest chunk a launch may take (65 536 gate instances are
// ~1 GB of accumulators, extracted samples and key-switch digits): at least `need` (<= cap) items, doubling from 4 096 so
// that a run of growing batches does not reallocate every time.
static size_t grown(size_t have, size_t need, size_t cap) {
    return std::max(need, std::min(std::max(cap, need), std::max<size_t>(2 * have, 4096)));
}

// Whether a launch of `cnt` gate instances runs as a rotation of roles (w64::MixPlan), and with which geometry.  Only where it
// can pay: the kernels chosen by launch size (br_variant 0) on the one-limb spectrum, the launch alone on the chip (no other
// stream of this context at work), a whole rotation, and a size mix_plan.h names: 4 .. 7 gates per CU, or a full round of the
// one-wave kernel plus a small remainder (8 .. 10.5 per CU).
static void ensure_lanes(Evaluator::Impl* d, int lanes);
// -> false, or the geometry (k subsets, tw of them on two waves at a time) a launch of cnt gate instances takes
static bool mix_geometry(const Evaluator::Impl* d, int64_t cnt, int32_t variant, int* k_out, int* tw_out) {
    if (!d->overlap || !d->br_mix || d->br_variant != 0 || d->concurrency != 1 || d->level_on_two_lanes || d->exact_fft || d->exact_once) return false;
    if (!d->use_w64 || !w64::variant_one_limb(variant)) return false;
    MixGeometry g;  // mix_plan.h: "mix_k" / "mix_tw" force a geometry (measurement aid), 0 = by launch size
    if (!mix_geometry_for(d->cus, cnt, d->mix_k, d->mix_tw, &g) || g.k > kMaxLanes) return false;
    *k_out = g.k;
    *tw_out = g.tw;
    return true;
}
static bool plan_mix(const Params& p, Evaluator::Impl* d, Lane& ln, int64_t cnt, int32_t variant, int32_t steps, w64::MixPlan* mix) {
    MixGeometry g;
    if (steps >= 0 || &ln != &d->lane[0] || !mix_geometry(d, cnt, variant, &g.k, &g.tw)) return false;
    MixSteps m;
    if (!mix_steps_for(p.n, g, d->mix_s1, d->mix_ratio, &m)) return false;
    ensure_lanes(d, g.k);
    for (int j = 0; j < g.k; j++)
        if (!d->ev_mix[j]) HIP_CHECK(hipEventCreateWithFlags(&d->ev_mix[j], hipEventDisableTiming));
    mix->k = g.k;
    mix->tw = g.tw;
    mix->s1 = m.s1;
    mix->s2 = m.s2;
    mix->cycles = m.cycles;
    mix->tail_s1 = m.tail_s1;
    mix->tail_s2 = m.tail_s2;
    mix->sync = d->mix_sync != 0;
    mix->wg = d->mix_wg;
    for (int j = 0; j < g.k; j++) {
        mix->streams[j] = d->lane[j].stream;
        mix->ev[j] = d->ev_mix[j];
    }
    return true;
}

// Runs `items` gate instances described by W (item0 is advanced per chunk).
static int launch_blind_rotate(const Params& p, Evaluator::Impl* d, Lane& ln, const WorkDesc& w, int64_t cnt,
                                Torus32* ext, int32_t steps, Torus32* dbg_acc) {
    hipStream_t stream = ln.stream;
    if (d->use_w64) {
        if (ln.br_state_items < (size_t)cnt) {
            if (ln.br_state) HIP_CHECK(hipFree(ln.br_state));
            ln.br_state = nullptr;
            const size_t items = grown(ln.br_state_items, (size_t)cnt, d->chunk);
            ln.br_state_items = 0;
            HIP_CHECK(hipMalloc(&ln.br_state, items * w64::state_bytes_per_item(p)));
            ln.br_state_items = items;
        }
        int32_t variant, slice;
        pick_br_variant(p, d, cnt, &variant, &slice);
        w64::MixPlan mix;
        const bool mixed = plan_mix(p, d, ln, cnt, variant, steps, &mix);
        if (mixed) d->mixed_launches++;
        return w64::launch(p, d->K, d->bkf_w64, d->bkf1_w64, d->fft_guard, w, cnt, ln.br_state, ext, steps, dbg_acc, slice, variant,
                           d->tw_w64, stream, pick_wg_gates(d, cnt), mixed ? &mix : nullptr);
    }
    else
        hipLaunchKernelGGL(k_blind_rotate_generic, dim3((unsigned)cnt), dim3(kThreads), d->br_lds, stream, d->K, w, ext,
                           steps, dbg_acc);
    return 1;
}

// The sampled audit behind the rounding guard: after a (level, chunk) launch that took a one-limb kernel, every
// fft_audit-th time, kAuditGates consecutive gate instances of it (at an offset that moves from audit to audit) are run
// again on the two-limb kernel -- exact by construction -- and their extracted samples compared word for word with what the
// one-limb kernel wrote to `ext`.  A differing row is counted on the device; the call then repeats itself on the
// two-limb kernels like a call whose guard tripped (Evaluator::fft_guard_tripped).  The guard watches the error LEVEL of
// every launch; this compares BITS, of a sample.
constexpr int64_t kAuditGates = 64;
static void maybe_audit(const Params& p, Evaluator::Impl* d, Lane& ln, const WorkDesc& w, int64_t cnt, const Torus32* ext) {
    hipStream_t stream = ln.stream;
    if (!d->use_w64 || d->fft_audit <= 0 || !d->fft_guard) return;
    int32_t variant, slice;
    pick_br_variant(p, d, cnt, &variant, &slice);
    if (!w64::variant_one_limb(variant)) return;  // the launch was exact by construction
    if (++d->audit_seq % d->fft_audit != 0) return;
    const int64_t m = std::min<int64_t>(kAuditGates, cnt);
    const int64_t off = cnt > m ? (int64_t)(((uint64_t)d->audit_seq * 0x9E3779B97F4A7C15ull >> 33) % (uint64_t)(cnt - m + 1)) : 0;
    if (!ln.audit_ext) HIP_CHECK(hipMalloc(&ln.audit_ext, (size_t)kAuditGates * (size_t)(d->K.N + 4) * 4));
    if (!ln.audit_state) HIP_CHECK(hipMalloc(&ln.audit_state, (size_t)kAuditGates * w64::state_bytes_per_item(p)));
    WorkDesc wa = w;
    wa.item0 = w.item0 + off;
    w64::launch(p, d->K, d->bkf_w64, d->bkf1_w64, d->fft_guard, wa, m, ln.audit_state, ln.audit_ext, -1, nullptr, w64::bara_stride(p),
                w64::kVariantWide, d->tw_w64, stream);
    hipLaunchKernelGGL(k_audit_compare, dim3((unsigned)m), dim3(256), 0, stream, ext + (size_t)off * (size_t)(d->K.N + 4), ln.audit_ext,
                       d->K.N, d->fft_guard + 2, d->audit_inject ? 1 : 0);
    HIP_CHECK(hipGetLastError());
    d->audit_inject = false;
    d->audits++;
    d->audit_gates += m;
}

// digit scratch of the MFMA key switch for launches of up to `cnt` gates (doubling from 4 096 gates' worth, capped at a chunk's)
static void reserve_ks_digits(Evaluator::Impl* d, Lane& ln, int64_t cnt) {
    const size_t need = ksm::digit_scratch_bytes(d->p, cnt);
    if (ln.ks_digits_bytes >= need) return;
    const size_t have = ln.ks_digits_bytes;
    if (ln.ks_digits) HIP_CHECK(hipFree(ln.ks_digits));
    ln.ks_digits = nullptr;
    ln.ks_digits_bytes = 0;
    const size_t want = std::max(need, std::min(ksm::digit_scratch_bytes(d->p, (int64_t)d->chunk),
                                                std::max<size_t>(2 * have, ksm::digit_scratch_bytes(d->p, 4096))));
    HIP_CHECK(hipMalloc(&ln.ks_digits, want));
    ln.ks_digits_bytes = want;
}

static void launch_keyswitch(Evaluator::Impl* d, Lane& ln, const WorkDesc& w, int64_t cnt, const Torus32* ext,
                             Torus32* flat_out, bool force_generic) {
    hipStream_t stream = ln.stream;
    const DevKeys& K = d->K;
    const dim3 grid((unsigned)cnt), blk(kKsThreads);
    const int nld = force_generic ? 0 : d->ks_nld;
    if (!force_generic && d->ks_mfma_ok && d->ks_limbs && cnt >= d->ks_mfma_min) {
        reserve_ks_digits(d, ln, cnt);
        ksm::launch(d->p, K, w, cnt, ext, flat_out, d->ks_limbs, ln.ks_digits, d->ks_mfma_split, d->cus, stream);
        return;
    }
    if (nld > 0 && d->ks_sliced_ok && cnt >= d->ks_sliced_min) {
        kss::launch(d->p, K, w, cnt, ext, flat_out, d->ks_slice, d->ks_gates, stream);
        return;
    }
    if (nld > 0 && d->ks_batch_ok && cnt >= d->ks_batch_min) {
        constexpr int G = 16;
        const size_t lds = (size_t)G * K.N * 2 + (size_t)G * 4;
        hipLaunchKernelGGL(k_keyswitch_batch<G>, dim3((unsigned)((cnt + G - 1) / G)), dim3(64 * nld), lds, stream, K, w, ext,
                           flat_out, cnt);
        return;
    }
    // a handful of gates: cut each gate's walk into `splits` workgroups
    int32_t splits = 1;
    if (nld > 0 && d->ks_split_max > 1) {
        // measured: pays while gates x splits stays within ~1.5 workgroups per CU (1-8 gates: 0.18 -> 0.03 ms, 44: 0.09, 256: no gain)
        while (splits < d->ks_split_max && cnt * splits * 2 <= (3 * (int64_t)d->cus) / 2 && K.N % (splits * 2) == 0) splits *= 2;
    }
    dim3 vgrid((unsigned)cnt, (unsigned)splits);
    if (splits > 1) hipLaunchKernelGGL(k_keyswitch_init, grid, dim3(256), 0, stream, K, w, ext, flat_out);
    switch (nld) {
        case 1: hipLaunchKernelGGL(k_keyswitch_vec<1>, vgrid, blk, d->ksv_lds, stream, K, w, ext, flat_out, splits); break;
        case 2: hipLaunchKernelGGL(k_keyswitch_vec<2>, vgrid, blk, d->ksv_lds, stream, K, w, ext, flat_out, splits); break;
        case 3: hipLaunchKernelGGL(k_keyswitch_vec<3>, vgrid, blk, d->ksv_lds, stream, K, w, ext, flat_out, splits); break;
        case 4: hipLaunchKernelGGL(k_keyswitch_vec<4>, vgrid, blk, d->ksv_lds, stream, K, w, ext, flat_out, splits); break;
        default:
            hipLaunchKernelGGL(k_keyswitch_generic, grid, dim3(kThreads), d->ks_lds, stream, K, w, ext, flat_out);
    }
}

// How a level of `items` gate instances is issued: on lane 0 in pieces of at most a chunk, or (overlap) in pieces of at most
// half the level that alternate between the two lanes.
struct LevelPlan {
    bool two_lanes;
    int64_t piece;  // gate instances per (blind rotation, key switch) pair of launches at most
};
static LevelPlan plan_level(const Evaluator::Impl* d, int64_t items) {
    const int64_t chunk = (int64_t)d->chunk;
    LevelPlan pl{false, chunk};
    if (d->overlap && d->use_w64 && d->concurrency == 1 && items >= d->overlap_min && items >= 2) {
        pl.two_lanes = true;
        const int64_t half = (((items + 1) / 2) + 3) & ~(int64_t)3;  // whole workgroups of the one-wave-per-gate kernels
        pl.piece = std::min(chunk, half);
    }
    return pl;
}

static void reserve_lane(const Params& p, Evaluator::Impl* d, Lane& ln, size_t need) {
    if (ln.ext_items < need) {
        const size_t n = grown(ln.ext_items, need, d->chunk);
        if (ln.ext) HIP_CHECK(hipFree(ln.ext));
        ln.ext = nullptr;
        ln.ext_items = 0;
        HIP_CHECK(hipMalloc(&ln.ext, n * (size_t)(d->K.N + 4) * 4));
        ln.ext_items = n;
    }
    if (d->use_w64 && ln.br_state_items < need) {
        const size_t n = grown(ln.br_state_items, need, d->chunk);
        if (ln.br_state) HIP_CHECK(hipFree(ln.br_state));
        ln.br_state = nullptr;
        ln.br_state_items = 0;
        HIP_CHECK(hipMalloc(&ln.br_state, n * w64::state_bytes_per_item(p)));
        ln.br_state_items = n;
    }
    // the MFMA key switch's digit scratch for the widest launch that will take it
    if (!d->force_generic_ks && d->ks_mfma_ok && d->ks_limbs && (int64_t)need >= d->ks_mfma_min) reserve_ks_digits(d, ln, (int64_t)need);
}

static void ensure_lanes(Evaluator::Impl* d, int lanes) {
    if (!d->ev_fork) HIP_CHECK(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
    for (int k = 1; k < lanes; k++) {
        if (!d->lane[k].stream) HIP_CHECK(hipStreamCreateWithFlags(&d->lane[k].stream, hipStreamNonBlocking));
        if (!d->ev_join[k]) HIP_CHECK(hipEventCreateWithFlags(&d->ev_join[k], hipEventDisableTiming));
    }
}
static void ensure_second_lane(Evaluator::Impl* d) { ensure_lanes(d, 2); }
// fork: lanes 1 .. lanes-1 start when everything queued so far on lane 0 is done; join: lane 0 waits for all of them
static void fork_lanes(Evaluator::Impl* d, int lanes) {
    HIP_CHECK(hipEventRecord(d->ev_fork, d->lane[0].stream));
    for (int k = 1; k < lanes; k++) HIP_CHECK(hipStreamWaitEvent(d->lane[k].stream, d->ev_fork, 0));
}
static void join_lanes(Evaluator::Impl* d, int lanes) {
    for (int k = 1; k < lanes; k++) {
        HIP_CHECK(hipEventRecord(d->ev_join[k], d->lane[k].stream));
        HIP_CHECK(hipStreamWaitEvent(d->lane[0].stream, d->ev_join[k], 0));
    }
}

// Before an evaluation starts: scratch for its widest launch in one go (the per-launch checks below then find it in place),
// so that a circuit whose levels widen does not reallocate -- and synchronise -- between them.  level_items: gate instances
// of each level the evaluation will issue.
static void reserve_scratch(const Params& p, Evaluator::Impl* d, const int64_t* level_items, size_t n_levels) {
    size_t need0 = 1, need1 = 0;
    for (size_t i = 0; i < n_levels; i++) {
        const int64_t items = std::max<int64_t>(level_items[i], 1);
        const LevelPlan pl = plan_level(d, items);
        const size_t piece = (size_t)std::min<int64_t>(pl.piece, items);
        need0 = std::max(need0, piece);
        if (pl.two_lanes) need1 = std::max(need1, std::min<size_t>(piece, (size_t)(items - (int64_t)piece)));
    }
    reserve_lane(p, d, d->lane[0], need0);
    if (need1) {
        ensure_second_lane(d);
        reserve_lane(p, d, d->lane[1], need1);
    }
}

// One level: `items` independent gate instances described by W (item0 is advanced per piece).  Everything queued so far
// on lane 0 (the previous level) is complete before any piece starts; lane 0 has every piece behind it when this returns.
// fixed_lane >= 0: the whole level on that lane, in pieces of at most a chunk, no fork / join (a pipeline of its own, see
// eval_circuit_device_once); its scratch has been reserved by the caller.
static void run_items(const Params& p, Evaluator::Impl* d, WorkDesc W, int64_t items, Timer& tbr, Timer& tks, EvalStats* stats,
                      int fixed_lane = -1) {
    LevelPlan pl = plan_level(d, items);
    if (fixed_lane >= 0) {
        pl.two_lanes = false;
        pl.piece = (int64_t)d->chunk;
        reserve_lane(p, d, d->lane[fixed_lane], (size_t)std::min<int64_t>(pl.piece, std::max<int64_t>(items, 1)));
    } else {
        reserve_scratch(p, d, &items, 1);
    }
    if (pl.two_lanes) {
        fork_lanes(d, 2);
        d->overlapped_levels++;
    }
    d->level_on_two_lanes = pl.two_lanes;
    int k = 0;
    for (int64_t done = 0; done < items; done += pl.piece, k++) {
        const int64_t cnt = std::min<int64_t>(pl.piece, items - done);
        Lane& ln = d->lane[fixed_lane >= 0 ? fixed_lane : (pl.two_lanes ? (k & 1) : 0)];
        WorkDesc w = W;
        w.item0 = W.item0 + done;
        tbr.mark(ln.stream);
        const int nbr = launch_blind_rotate(p, d, ln, w, cnt, ln.ext, -1, nullptr);
        tbr.mark(ln.stream);
        HIP_CHECK(hipGetLastError());
        maybe_audit(p, d, ln, w, cnt, ln.ext);
        tks.mark(ln.stream);
        launch_keyswitch(d, ln, w, cnt, ln.ext, nullptr, d->force_generic_ks);
        tks.mark(ln.stream);
        HIP_CHECK(hipGetLastError());
        if (stats) {
            stats->blind_rotate_launches += nbr;
            stats->keyswitch_launches++;
            stats->chunks++;
        }
    }
    d->level_on_two_lanes = false;
    if (pl.two_lanes) join_lanes(d, 2);
    if (stats) stats->bootstraps += items;
}

// After a synchronous call: fold the one-limb kernel's guard record into the context and tell whether the call has to be
// repeated on the two-limb kernel (some launch saw a coefficient further than kGuardLimit from an integer).
bool Evaluator::fft_guard_tripped() {
    if (!d_->fft_guard) return false;
    unsigned h[3] = {0, 0, 0};
    HIP_CHECK(hipMemcpy(h, d_->fft_guard, sizeof h, hipMemcpyDeviceToHost));
    float m;
    memcpy(&m, &h[1], sizeof m);
    if ((double)m > d_->guard_max) d_->guard_max = (double)m;
    if (h[0] == 0 && h[2] == 0) return false;
    d_->audit_mismatches += h[2];
    // the two counts only; the maximum stays
    HIP_CHECK(hipMemset(d_->fft_guard, 0, sizeof(unsigned)));
    HIP_CHECK(hipMemset(d_->fft_guard + 2, 0, sizeof(unsigned)));
    return true;
}

void Evaluator::fft_audit_counts(int64_t* audits, int64_t* gates, int64_t* mismatches) const {
    if (audits) *audits = d_->audits;
    if (gates) *gates = d_->audit_gates;
    if (mismatches) *mismatches = d_->audit_mismatches;
}

double Evaluator::fft_guard_max() const { return d_->guard_max; }
int64_t Evaluator::fft_guard_reruns() const { return d_->guard_reruns; }

namespace {
bool overlaps(const Torus32* a, size_t na, const Torus32* b, size_t nb) {
    return a && b && a < b + nb && b < a + na;
}
// Repeats `once` on the two-limb kernels when the guard tripped or an audited row differed; the first attempt's time stays
// in the stats, its counts do not.  A call whose output overlaps its inputs cannot be repeated (the first attempt has
// overwritten them), so it runs on the two-limb kernels -- exact by construction -- from the start.
template <class F>
void run_guarded(Evaluator& ev, bool* exact_once, int64_t* reruns, bool inputs_intact, EvalStats* stats, F&& once) {
    if (!inputs_intact && !*exact_once) {
        *exact_once = true;
        try {
            once();
        } catch (...) {
            *exact_once = false;
            throw;
        }
        *exact_once = false;
        (void)ev.fft_guard_tripped();  // folds the record of earlier calls; nothing this call did can trip it
        return;
    }
    const EvalStats before = stats ? *stats : EvalStats{};
    once();
    if (!ev.fft_guard_tripped()) return;
    (*reruns)++;
    EvalStats first{};
    if (stats) {
        first = *stats;
        *stats = before;
    }
    *exact_once = true;
    try {
        once();
    } catch (...) {
        *exact_once = false;
        throw;
    }
    *exact_once = false;
    if (stats) stats->total_ms += first.total_ms - before.total_ms;
}
}  // namespace

void Evaluator::gates_device(int32_t type, size_t count, const Torus32* d_a, const Torus32* d_b, Torus32* d_out,
                             EvalStats* stats) {
    const size_t len = count * (size_t)d_->K.stride;
    run_guarded(*this, &d_->exact_once, &d_->guard_reruns, !overlaps(d_out, len, d_a, len) && !overlaps(d_out, len, d_b, len), stats,
                [&] { gates_device_once(type, count, d_a, d_b, d_out, stats); });
}

void Evaluator::mux_device(size_t count, const Torus32* d_a, const Torus32* d_b, const Torus32* d_c, Torus32* d_out,
                           EvalStats* stats) {
    const size_t len = count * (size_t)d_->K.stride;
    run_guarded(*this, &d_->exact_once, &d_->guard_reruns,
                !overlaps(d_out, len, d_a, len) && !overlaps(d_out, len, d_b, len) && !overlaps(d_out, len, d_c, len), stats,
                [&] { mux_device_once(count, d_a, d_b, d_c, d_out, stats); });
}

void Evaluator::eval_circuit_device(const Circuit& c, size_t batch, const Torus32* d_in, Torus32* d_out, EvalStats* stats) {
    const size_t stride = (size_t)d_->K.stride;
    run_guarded(*this, &d_->exact_once, &d_->guard_reruns,
                !overlaps(d_out, batch * c.outputs.size() * stride, d_in, batch * (size_t)c.n_inputs * stride), stats,
                [&] { eval_circuit_device_once(c, batch, d_in, d_out, stats); });
}

void Evaluator::debug_blind_rotate(size_t count, const Torus32* d_x, Torus32* d_acc, int32_t steps) {
    run_guarded(*this, &d_->exact_once, &d_->guard_reruns, true, nullptr, [&] { debug_blind_rotate_once(count, d_x, d_acc, steps); });
}

void Evaluator::gates_device_once(int32_t type, size_t count, const Torus32* d_a, const Torus32* d_b, Torus32* d_out,
                                  EvalStats* stats) {
    if (!keys_loaded_) throw std::runtime_error("cloud key not loaded");
    HIP_CHECK(hipSetDevice(device_));
    if (count == 0) return;
    d_->use_w64 = w64::supported(p_) && !force_generic_;
    d_->force_generic_ks = force_generic_;
    Timer tall(stats != nullptr, stream_), tbr(stats != nullptr, stream_), tks(stats != nullptr, stream_);
    WorkDesc W{};
    W.gates = nullptr;
    W.flat_a = d_a;
    W.flat_b = d_b;
    W.flat_out = d_out;
    W.flat_type = type;
    W.item0 = 0;
    tall.mark();
    run_items(p_, d_, W, (int64_t)count, tbr, tks, stats);
    tall.mark();
    HIP_CHECK(hipStreamSynchronize(stream_));
    if (stats) {
        stats->total_ms += tall.sum_ms();
        stats->blind_rotate_ms += tbr.sum_ms();
        stats->keyswitch_ms += tks.sum_ms();
        stats->levels += 1;
    }
}

// bootsMUX (boot-gates.cpp): two blind rotations per gate, their extracted samples added, one key switch
void Evaluator::mux_device_once(size_t count, const Torus32* d_a, const Torus32* d_b, const Torus32* d_c, Torus32* d_out,
                                EvalStats* stats) {
    if (!keys_loaded_) throw std::runtime_error("cloud key not loaded");
    HIP_CHECK(hipSetDevice(device_));
    if (count == 0) return;
    d_->use_w64 = w64::supported(p_) && !force_generic_;
    d_->force_generic_ks = force_generic_;
    const DevKeys& K = d_->K;
    const size_t chunk = std::max<size_t>(d_->chunk & ~(size_t)1, 2), gates_per_chunk = chunk / 2;
    const size_t mux_need = std::min(gates_per_chunk, count);  // two extracted samples per MUX gate
    Lane& ln = d_->lane[0];
    if (ln.ext_items < 2 * mux_need) {
        const size_t n = grown(ln.ext_items, 2 * mux_need, chunk);
        if (ln.ext) HIP_CHECK(hipFree(ln.ext));
        ln.ext = nullptr;
        ln.ext_items = 0;
        HIP_CHECK(hipMalloc(&ln.ext, n * (size_t)(K.N + 4) * 4));
        ln.ext_items = n;
    }
    if (d_->ext_mux_items < mux_need) {
        const size_t n = grown(d_->ext_mux_items, mux_need, gates_per_chunk);
        if (d_->ext_mux) HIP_CHECK(hipFree(d_->ext_mux));
        d_->ext_mux = nullptr;
        d_->ext_mux_items = 0;
        HIP_CHECK(hipMalloc(&d_->ext_mux, n * (size_t)(K.N + 4) * 4));
        d_->ext_mux_items = n;
    }
    Timer tall(stats != nullptr, stream_), tbr(stats != nullptr, stream_), tks(stats != nullptr, stream_);
    tall.mark();
    for (size_t done = 0; done < count; done += gates_per_chunk) {
        const int64_t cnt = (int64_t)std::min(gates_per_chunk, count - done);
        WorkDesc W{};
        W.flat_a = d_a + done * K.stride;
        W.flat_b = d_b + done * K.stride;
        W.flat_c = d_c + done * K.stride;
        W.flat_type = kFlatMux;
        W.item0 = 0;
        tbr.mark();
        const int nbr = launch_blind_rotate(p_, d_, ln, W, 2 * cnt, ln.ext, -1, nullptr);
        tbr.mark();
        HIP_CHECK(hipGetLastError());
        maybe_audit(p_, d_, ln, W, 2 * cnt, ln.ext);
        tks.mark();
        hipLaunchKernelGGL(k_mux_combine, dim3((unsigned)cnt), dim3(256), 0, stream_, ln.ext, d_->ext_mux, K.N);
        WorkDesc Wk{};
        launch_keyswitch(d_, ln, Wk, cnt, d_->ext_mux, d_out + done * K.stride, d_->force_generic_ks);
        tks.mark();
        HIP_CHECK(hipGetLastError());
        if (stats) {
            stats->blind_rotate_launches += nbr;
            stats->keyswitch_launches++;
            stats->chunks++;
        }
    }
    tall.mark();
    HIP_CHECK(hipStreamSynchronize(stream_));
    if (stats) {
        stats->bootstraps += 2 * (int64_t)count;  // blind rotations; libtfhe counts a MUX as two bootstraps and one key switch
        stats->total_ms += tall.sum_ms();
        stats->blind_rotate_ms += tbr.sum_ms();
        stats->keyswitch_ms += tks.sum_ms();
        stats->levels += 1;
    }
}

// Everything an evaluation of `c` over `batch` expressions allocates -- the wire store, the gate / output tables, the scratch
// of its widest level (extracted samples, blind-rotation state, key-switch digits) -- so that the evaluation itself makes no
// allocation (each one is a device-wide synchronisation).  eval_circuit_device calls it; a caller that times its first
// evaluation calls it beforehand (ieache_prepare_batch).
// Whether an evaluation of `c` over `batch` expressions runs as two expression-half pipelines (Impl::pipe_min).
// -> 0 (no) or the number of pipelines (each gets at least one expression)
// trial (may be null): set to 0 / 1 when this evaluation is one of the two timed trials of its (circuit, batch) -- without /
// with pipelines -- and to -1 otherwise; the caller then reports the evaluation's wall time to tune_report().
using TuneKey = std::tuple<size_t, int32_t, size_t, size_t, bool>;
static TuneKey tune_key(const Evaluator::Impl* d, const Circuit& c, size_t batch) {
    return std::make_tuple(c.gates.size(), (int32_t)c.n_levels(), c.outputs.size(), batch, d->exact_fft || d->exact_once);
}
static int pipelined(Evaluator::Impl* d, const Circuit& c, size_t batch, int* trial = nullptr, bool either = false) {
    if (trial) *trial = -1;
    if (!d->overlap || !d->use_w64 || batch < 2 || c.n_levels() < 1) return 0;
    const int lanes = (int)std::min<size_t>((size_t)d->pipe_lanes, batch);
    const int64_t gates = (int64_t)c.level_offset[c.n_levels()] - (int64_t)c.level_offset[0];
    const int64_t work = gates * (int64_t)batch, bar = d->pipe_min * (int64_t)c.n_levels();  // mean level against pipe_min
    // From 2 x pipe_min on pipelines won every measurement.  Below, it depends on where the levels fall among the kernels'
    // regimes (a level of 2 200 gate instances as two pipelines' launches of 1 100 costs a second, nearly empty round; on one
    // stream it runs as a rotation of roles): tried both ways from pipe_min / 8 up.
    if (work >= 2 * bar || (!d->pipe_auto && work >= bar)) return lanes;
    if (d->pipe_auto && d->pipe_min > 0 && work * 8 >= bar) {
        if (either) return lanes;  // scratch for both modes
        const auto it = d->tuned.find(tune_key(d, c, batch));
        const Evaluator::Impl::Tuned t = it == d->tuned.end() ? Evaluator::Impl::Tuned{} : it->second;
        const bool trying = t.n[0] < 2 || t.n[1] < 2;
        const int mode = trying ? (t.n[0] <= t.n[1] ? 0 : 1) : (t.ms[1] < t.ms[0] ? 1 : 0);
        if (trial && trying) *trial = mode;
        return mode ? lanes : 0;
    }
    return 0;
}
static void tune_report(Evaluator::Impl* d, const Circuit& c, size_t batch, int trial, double ms) {
    if (trial < 0 || trial > 1) return;
    if (d->tuned.size() > 256) d->tuned.clear();  // a daemon sees many batch sizes; the table is a cache, not a record
    Evaluator::Impl::Tuned& t = d->tuned[tune_key(d, c, batch)];
    t.ms[trial] = t.n[trial] == 0 ? ms : std::min(t.ms[trial], ms);
    t.n[trial]++;
    d->tuned_evals++;
}
// expressions [first, first + count) of pipeline k of `lanes`: contiguous, sizes differing by at most one, the first ones longer
static void pipe_slice(size_t batch, int lanes, int k, size_t* first, size_t* count) {
    const size_t base = batch / (size_t)lanes, extra = batch % (size_t)lanes;
    *first = (size_t)k * base + std::min<size_t>((size_t)k, extra);
    *count = base + ((size_t)k < extra ? 1 : 0);
}

void Evaluator::prepare_circuit(const Circuit& c, size_t batch) {
    if (!keys_loaded_) throw std::runtime_error("cloud key not loaded");
    HIP_CHECK(hipSetDevice(device_));
    if (batch == 0) return;
    d_->use_w64 = w64::supported(p_) && !force_generic_;
    d_->force_generic_ks = force_generic_;
    const size_t row_bytes = (size_t)d_->K.stride * 4;
    const size_t need = batch * (size_t)c.n_slots * row_bytes;
    if (d_->store_bytes < need) {
        if (d_->store) HIP_CHECK(hipFree(d_->store));
        d_->store = nullptr;
        d_->store_bytes = 0;
        HIP_CHECK(hipMalloc(&d_->store, need));
        d_->store_bytes = need;
    }
    if (d_->d_gates_cap < c.gates.size()) {
        if (d_->d_gates) HIP_CHECK(hipFree(d_->d_gates));
        d_->d_gates = nullptr;
        d_->d_gates_cap = 0;
        HIP_CHECK(hipMalloc(&d_->d_gates, c.gates.size() * sizeof(DevGate)));
        d_->d_gates_cap = c.gates.size();
    }
    if (d_->d_outs_cap < c.outputs.size()) {
        if (d_->d_outs) HIP_CHECK(hipFree(d_->d_outs));
        d_->d_outs = nullptr;
        d_->d_outs_cap = 0;
        HIP_CHECK(hipMalloc(&d_->d_outs, c.outputs.size() * sizeof(OutRef)));
        d_->d_outs_cap = c.outputs.size();
    }
    std::vector<int64_t> level_items;
    for (int32_t L = 1; L <= c.n_levels(); L++)
        level_items.push_back((int64_t)(c.level_offset[L] - c.level_offset[L - 1]) * (int64_t)batch);
    if (const int lanes = pipelined(d_, c, batch, nullptr, /*either=*/true)) {
        // pipelines of batch / lanes expressions each (the first ones take the odd ones); where the mode is still being
        // tried out (pipe_auto) the one-stream scratch is reserved as well
        if (!pipelined(d_, c, batch)) reserve_scratch(p_, d_, level_items.data(), level_items.size());
        int64_t widest = 1;
        for (int32_t L = 1; L <= c.n_levels(); L++) widest = std::max<int64_t>(widest, c.level_offset[L] - c.level_offset[L - 1]);
        ensure_lanes(d_, lanes);
        for (int k = 0; k < lanes; k++) {
            size_t first = 0, count = 0;
            pipe_slice(batch, lanes, k, &first, &count);
            reserve_lane(p_, d_, d_->lane[k], std::min<size_t>(d_->chunk, (size_t)widest * count));
        }
    } else {
        reserve_scratch(p_, d_, level_items.data(), level_items.size());
    }
}

void Evaluator::eval_circuit_device_once(const Circuit& c, size_t batch, const Torus32* d_in, Torus32* d_out,
                                         EvalStats* stats) {
    if (!keys_loaded_) throw std::runtime_error("cloud key not loaded");
    HIP_CHECK(hipSetDevice(device_));
    if (batch == 0) return;
    d_->use_w64 = w64::supported(p_) && !force_generic_;
    d_->force_generic_ks = force_generic_;
    const int32_t stride = d_->K.stride;
    const size_t row_bytes = (size_t)stride * 4;
    prepare_circuit(c, batch);
    if (!c.gates.empty())
        HIP_CHECK(hipMemcpyAsync(d_->d_gates, c.gates.data(), c.gates.size() * sizeof(DevGate), hipMemcpyHostToDevice, stream_));
    HIP_CHECK(hipMemcpyAsync(d_->d_outs, c.outputs.data(), c.outputs.size() * sizeof(OutRef), hipMemcpyHostToDevice, stream_));
    Timer tall(stats != nullptr, stream_), tbr(stats != nullptr, stream_), tks(stats != nullptr, stream_);
    tall.mark();
    // inputs -> slots 0..n_inputs-1 of every expression
    HIP_CHECK(hipMemcpy2DAsync(d_->store, (size_t)c.n_slots * row_bytes, d_in, (size_t)c.n_inputs * row_bytes,
                               (size_t)c.n_inputs * row_bytes, batch, hipMemcpyDeviceToDevice, stream_));
    int trial = -1;
    const int pipes = pipelined(d_, c, batch, &trial);
    const auto wall0 = std::chrono::steady_clock::now();
    if (pipes) {
        // fork: the other pipelines start when the inputs are in the wire store
        fork_lanes(d_, pipes);
        d_->concurrency = pipes;
        d_->pipelined_evals++;
    }
    try {
        for (int32_t L = 1; L <= c.n_levels(); L++) {
            WorkDesc W{};
            W.gates = d_->d_gates;
            W.g0 = c.level_offset[L - 1];
            W.ng = c.level_offset[L] - c.level_offset[L - 1];
            W.store = d_->store;
            W.n_slots = c.n_slots;
            W.item0 = 0;
            if (!pipes) {
                run_items(p_, d_, W, (int64_t)W.ng * (int64_t)batch, tbr, tks, stats);
            } else {
                // items are expression-major (item = expression x ng + gate): a contiguous range of expressions is a contiguous range of items
                for (int k = 0; k < pipes; k++) {
                    size_t first = 0, count = 0;
                    pipe_slice(batch, pipes, k, &first, &count);
                    W.item0 = (int64_t)W.ng * (int64_t)first;
                    run_items(p_, d_, W, (int64_t)W.ng * (int64_t)count, tbr, tks, stats, k);
                }
            }
            if (stats) stats->levels++;
        }
    } catch (...) {
        d_->concurrency = 1;
        throw;
    }
    if (pipes) {
        d_->concurrency = 1;
        join_lanes(d_, pipes);
    }
    const int32_t n_out = (int32_t)c.outputs.size();
    hipLaunchKernelGGL(k_gather_outputs, dim3((unsigned)(batch * n_out)), dim3(128), 0, stream_, d_->d_outs, n_out,
                       d_->store, c.n_slots, d_out, (int64_t)batch, stride, p_.n);
    HIP_CHECK(hipGetLastError());
    tall.mark();
    HIP_CHECK(hipStreamSynchronize(stream_));
    tune_report(d_, c, batch, trial, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    if (stats) {
        stats->total_ms += tall.sum_ms();
        stats->blind_rotate_ms += tbr.sum_ms();
        stats->keyswitch_ms += tks.sum_ms();
    }
}

void Evaluator::debug_blind_rotate_once(size_t count, const Torus32* d_x, Torus32* d_acc, int32_t steps) {
    if (!keys_loaded_) throw std::runtime_error("cloud key not loaded");
    HIP_CHECK(hipSetDevice(device_));
    WorkDesc W{};
    W.flat_a = d_x;
    W.flat_b = nullptr;
    W.flat_out = nullptr;
    W.flat_type = -1;
    d_->use_w64 = w64::supported(p_) && !force_generic_;
    d_->force_generic_ks = force_generic_;
    launch_blind_rotate(p_, d_, d_->lane[0], W, (int64_t)count, nullptr, steps, d_acc);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void Evaluator::debug_keyswitch(size_t count, const Torus32* d_u, Torus32* d_out) {
    if (!keys_loaded_) throw std::runtime_error("cloud key not loaded");
    HIP_CHECK(hipSetDevice(device_));
    // the kernel reads rows of N+4 ints; repack the caller's N+1 rows
    Torus32* tmp = nullptr;
    HIP_CHECK(hipMalloc(&tmp, count * (size_t)(p_.N + 4) * 4));
    HIP_CHECK(hipMemcpy2DAsync(tmp, (size_t)(p_.N + 4) * 4, d_u, (size_t)(p_.N + 1) * 4, (size_t)(p_.N + 1) * 4, count,
                               hipMemcpyDeviceToDevice, stream_));
    WorkDesc W{};
    launch_keyswitch(d_, d_->lane[0], W, (int64_t)count, tmp, d_out, force_generic_);
    hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(stream_);
    (void)hipFree(tmp);
    HIP_CHECK(e);
}

}  // namespace ieache
