// Device primitives of the 64-lane blind-rotation kernels (blind_rotate_w64.hip): the 512-point complex transform as 8 x 8 x 8
// radix-8 passes in registers (8 points per lane), its register <-> lane transposes (cross-lane or through a padded,
// conflict-free LDS tile), the twiddle table, the twist constants and the BK block load through the buffer path.
// No reference counterpart: libtfhe multiplies polynomials with a scalar FFT (tGswFFTExternMulToTLwe).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "dft8_twist.h"

namespace ieache {
namespace w64 {
namespace {

constexpr int kN = 1024, kM = 512;

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// forward: multiply by -i ; inverse: by +i
template <bool INV>
__device__ __forceinline__ double2 rot90(double2 z) {
    return INV ? make_double2(-z.y, z.x) : make_double2(z.y, -z.x);
}
// a * b  or  a * conj(b)
template <bool CONJ>
__device__ __forceinline__ double2 cmulx(double2 a, double2 b) {
    return CONJ ? make_double2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -a.x * b.y))
                : make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}

// 8-point DFT in registers, natural order in and out.  52 FP64 operations.
template <bool INV>
__device__ __forceinline__ void dft8(double2 (&x)[8]) {
    const double2 a0 = cadd(x[0], x[4]), a1 = cadd(x[1], x[5]), a2 = cadd(x[2], x[6]), a3 = cadd(x[3], x[7]);
    const double2 b0 = csub(x[0], x[4]), t1 = csub(x[1], x[5]), t2 = csub(x[2], x[6]), t3 = csub(x[3], x[7]);
    // even outputs: DFT4(a)
    const double2 c0 = cadd(a0, a2), c1 = cadd(a1, a3), c2 = csub(a0, a2), c3 = rot90<INV>(csub(a1, a3));
    x[0] = cadd(c0, c1);
    x[4] = csub(c0, c1);
    x[2] = cadd(c2, c3);
    x[6] = csub(c2, c3);
    // odd outputs: DFT4(b), b_j = t_j * W8^j with the 1/sqrt2 factors deferred into the last FMAs
    const double2 b2 = rot90<INV>(t2);
    // forward: t1*(1-i), t3*(-1-i) ; inverse: t1*(1+i), t3*(-1+i)
    const double2 b1 = INV ? make_double2(t1.x - t1.y, t1.x + t1.y) : make_double2(t1.x + t1.y, t1.y - t1.x);
    const double2 b3 = INV ? make_double2(-t3.x - t3.y, t3.x - t3.y) : make_double2(t3.y - t3.x, -t3.x - t3.y);
    const double2 e0 = cadd(b0, b2), e2 = csub(b0, b2);
    const double2 s = cadd(b1, b3), d = rot90<INV>(csub(b1, b3));
    x[1] = make_double2(fma(kR, s.x, e0.x), fma(kR, s.y, e0.y));
    x[5] = make_double2(fma(-kR, s.x, e0.x), fma(-kR, s.y, e0.y));
    x[3] = make_double2(fma(kR, d.x, e2.x), fma(kR, d.y, e2.y));
    x[7] = make_double2(fma(-kR, d.x, e2.x), fma(-kR, d.y, e2.y));
}

// Orders this wave's LDS traffic without a workgroup barrier: the DS instructions of
// one wave execute in issue order, so only the compiler has to be held back.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// diagnostic cycle stamp (s_memtime), fenced so segments are not reordered across it
__device__ __forceinline__ unsigned long long stamp() {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <bool WSYNC>
__device__ __forceinline__ void tile_sync() {
    if (WSYNC)
        wave_sync();
    else
        __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
}

// Twiddle table, one per workgroup in LDS (9 KiB), used by every transform in both
// directions (the inverse multiplies by the conjugates):
//   tw[k*64 + lane]       = exp(i*pi*lane/1024) * exp(-2*pi*i*lane*k/512)   twist (lane part) x first inter-pass twiddle
//   tw[512 + k*8 + p0]    = exp(-2*pi*i*p0*k/64)                            second inter-pass twiddle, p0 = lane & 7
// Keeping them in registers costs 60 VGPRs per wave, which the BK prefetch needs more.
constexpr int kTwElems = 8 * 64 + 8 * 8;
struct LaneRoots {
    const double2* t1;  // &tw[lane], stride 64
    const double2* t2;  // &tw[512 + (lane & 7)], stride 8
    __device__ __forceinline__ double2 a(int k) const { return t1[k * 64]; }
    __device__ __forceinline__ double2 b(int k) const { return t2[k * 8]; }
};
__device__ __forceinline__ void build_twiddles(double2* tw, int tid, int nthreads) {
    double s, c;
    for (int idx = tid; idx < 512; idx += nthreads) {
        const int k = idx >> 6, lane = idx & 63;
        sincospi((double)(lane * (1 - 4 * k)) / 1024.0, &s, &c);  // lane/1024 - 2*lane*k/512
        tw[idx] = make_double2(c, s);
    }
    for (int idx = tid; idx < 64; idx += nthreads) {
        const int k = idx >> 3, p0 = idx & 7;
        sincospi(-(double)(p0 * k) / 32.0, &s, &c);
        tw[512 + idx] = make_double2(c, s);
    }
}

// The table is built once per context (k_build_twiddle_table) and copied into LDS at kernel start:
// computing it per workgroup (4.5 sincospi per thread) cost ~4 % of a 16-step slice's vector work.
__global__ __launch_bounds__(128) void k_build_twiddle_table(double2* tw) { build_twiddles(tw, threadIdx.x, 128); }

__device__ __forceinline__ void load_twiddles(double2* sTw, const double2* __restrict__ gtw, int tid, int nthreads) {
    for (int idx = tid; idx < kTwElems; idx += nthreads) sTw[idx] = gtw[idx];
}

__device__ __forceinline__ LaneRoots make_roots(const double2* tw, int lane) {
    LaneRoots r;
    r.t1 = tw + lane;
    r.t2 = tw + 512 + (lane & 7);
    return r;
}

// ---- register <-> lane transposes without LDS ----
// Both transposes of the 8x8x8 transform swap the 3 bits of the register index with 3 bits
// of the lane index.  Swapping ONE register bit with ONE lane bit B is an exchange between
// lanes l and l ^ (1 << B): v_permlane32_swap / v_permlane16_swap do exactly that for B = 5, 4
// (one instruction per dword pair), DPP row/quad moves for B = 3..0.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

template <int B>
__device__ __forceinline__ void swap_dwords(unsigned& lo, unsigned& hi) {
    // lo: a dword of x[r] (register bit clear), hi: the same dword of x[r | bit].
    // After the call, lanes with bit B clear hold in `hi` what the partner lane had in `lo`, and
    // lanes with bit B set hold in `lo` what the partner had in `hi`.
    static_assert(B == 5 || B == 4, "v_permlane32_swap / v_permlane16_swap");
    const v2u_t r = B == 5 ? __builtin_amdgcn_permlane32_swap(lo, hi, false, false) : __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
    lo = r[0];
    hi = r[1];
}

// Lane bit 3 has no swap instruction.  v_cndmask_b32 takes a DPP source itself -- new_hi = set ? hi : lo[lane ^ 8],
// new_lo = set ? hi[lane ^ 8] : lo with set = lane bit 3, VCC flipped in between by the scalar unit -- two instructions per
// dword pair, into fresh registers (two DPP moves with bank masks need a register copy on top: three).  Four dwords (one
// double2) per block; the s_nop covers the VALU-write -> DPP-read wait states the assembler does not insert inside an asm block.
__device__ __forceinline__ void swap4_row_ror8(unsigned (&a)[4], unsigned (&b)[4]) {
    unsigned na0, na1, na2, na3, nb0, nb1, nb2, nb3;
    asm volatile("s_mov_b32 vcc_lo, 0xff00ff00\n\t"
                 "s_mov_b32 vcc_hi, 0xff00ff00\n\t"
                 "s_nop 1\n\t"
                 "v_cndmask_b32_dpp %4, %8, %12, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %5, %9, %13, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %6, %10, %14, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %7, %11, %15, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %0, %12, %8, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %1, %13, %9, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %2, %14, %10, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_dpp %3, %15, %11, vcc row_ror:8 row_mask:0xf bank_mask:0xf"
                 : "=&v"(na0), "=&v"(na1), "=&v"(na2), "=&v"(na3), "=&v"(nb0), "=&v"(nb1), "=&v"(nb2), "=&v"(nb3)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
                 : "vcc");
    a[0] = na0, a[1] = na1, a[2] = na2, a[3] = na3;
    b[0] = nb0, b[1] = nb1, b[2] = nb2, b[3] = nb3;
}

// register bit (B - 3) <-> lane bit B, B = 3, 4, 5
template <int B>
__device__ __forceinline__ void bitswap(double2 (&x)[8]) {
    constexpr int m = 1 << (B - 3);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r & m) continue;
        unsigned a[4] = {(unsigned)__double2loint(x[r].x), (unsigned)__double2hiint(x[r].x), (unsigned)__double2loint(x[r].y),
                         (unsigned)__double2hiint(x[r].y)};
        unsigned b[4] = {(unsigned)__double2loint(x[r | m].x), (unsigned)__double2hiint(x[r | m].x), (unsigned)__double2loint(x[r | m].y),
                         (unsigned)__double2hiint(x[r | m].y)};
        if constexpr (B == 3) {
            swap4_row_ror8(a, b);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) swap_dwords<B>(a[q], b[q]);
        }
        x[r] = make_double2(__hiloint2double((int)a[1], (int)a[0]), __hiloint2double((int)a[3], (int)a[2]));
        x[r | m] = make_double2(__hiloint2double((int)b[1], (int)b[0]), __hiloint2double((int)b[3], (int)b[2]));
    }
}
// register index <-> lane bits 3..5 (what the first LDS transpose of the forward transform does)
__device__ __forceinline__ void xlane_hi(double2 (&x)[8]) {
    bitswap<3>(x);
    bitswap<4>(x);
    bitswap<5>(x);
}

// Transpose tiles hold element (h, m, l) -- three 3-bit digits -- at h*72 + m*9 + l.
// The 9/72 padding makes every ds_write_b128 / ds_read_b128 of both transposes
// bank-conflict free AND lets each access be "per-lane base + immediate offset".
constexpr int kTile = 8 * 72;  // double2 elements per tile (9216 B)

// Forward 512-point transform of the twisted polynomial.
//   in : x[r] = y_{64r+lane}, untwisted (the first pass applies exp(i*pi*r/16), the first twiddle set the lane part tL)
//   out: x[k2] = X[k0 + 8*k1 + 64*k2] with lane = 8*k0 + k1
// XLANE = 1: the first (lane-high) transpose cross-lane (v_permlane*_swap / v_cndmask_b32_dpp) instead of through the tile.
// MID: called once the first inter-pass twiddles are consumed (their 32 VGPRs are free from there on) or, MID_LATE, once
// the second set is consumed too: the place to request data the caller needs right after the transform.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <bool WSYNC, int XLANE = 0, class MID = NoHook, bool MID_LATE = false>
__device__ __forceinline__ void fft512_forward(double2 (&x)[8], double2* sT, int lane, const LaneRoots& R, MID mid = MID()) {
    static_assert(XLANE == 0 || XLANE == 1, "lane-low transposes go through the tile");
    const int hi = lane >> 3, lo = lane & 7;
    const int own = hi * 9 + lo;   // (m, l) = (lane>>3, lane&7) inside a row-block h
    const int blk = hi * 72 + lo;  // (h, l) = (lane>>3, lane&7)
    // twiddles are fetched from the LDS table ahead of the butterflies that hide their latency
    double2 tA[8], tB[8];
#pragma unroll
    for (int k = 0; k < 8; k++) tA[k] = R.a(k);
    dft8_twist_fwd(x);                       // over r -> k0, the register part e^{i pi r/16} of the twist included
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cmulx<false>(x[k], tA[k]);  // * tL * w512^(lane*k0)
    if (!std::is_same<MID, NoHook>::value && !MID_LATE) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 1; k < 8; k++) tB[k] = R.b(k);
    if (XLANE) {
        xlane_hi(x);                                                // reg k0 <-> lane bits 3..5: lane = (k0, p0), reg = p1
    } else {
#pragma unroll
        for (int k0 = 0; k0 < 8; k0++) sT[own + 72 * k0] = x[k0];   // element (k0, p1, p0), lane = (p1, p0)
        tile_sync<WSYNC>();
#pragma unroll
        for (int p1 = 0; p1 < 8; p1++) x[p1] = sT[blk + 9 * p1];    // lane = (k0, p0)
        tile_sync<WSYNC>();
    }
    dft8<false>(x);                          // over p1 -> k1 ; lane = 8*k0 + p0
#pragma unroll
    for (int k = 1; k < 8; k++) x[k] = cmulx<false>(x[k], tB[k]);  // * w64^(p0*k1)
    if (!std::is_same<MID, NoHook>::value && MID_LATE) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) sT[blk + 9 * k1] = x[k1];        // element (k0, k1, p0), lane = (k0, p0)
    tile_sync<WSYNC>();
    const int rd = hi * 72 + lo * 9;                                // lane = (k0, k1)
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = sT[rd + q];
    tile_sync<WSYNC>();
    dft8<false>(x);                          // over p0 -> k2 ; lane = 8*k0 + k1
}

// Inverse of fft512_forward (unnormalised: 512 x), also removing the lane part of the twist:
//   in : spectrum in the layout fft512_forward produces
//   out: x[r] * untwist_gain(r) = y_{64r+lane}: untwisted and normalised up to one real factor per register, which the
//        caller folds into the FMA that rounds (round_coef)
template <bool WSYNC>
__device__ __forceinline__ void fft512_inverse(double2 (&x)[8], double2* sT, int lane, const LaneRoots& R) {
    const int hi = lane >> 3, lo = lane & 7;
    const int own = hi * 9 + lo, blk = hi * 72 + lo, rd = hi * 72 + lo * 9;
    double2 tA[8], tB[8];
#pragma unroll
    for (int k = 1; k < 8; k++) tB[k] = R.b(k);
    dft8<true>(x);  // k2 -> p0 ; lane = (k0, k1)
#pragma unroll
    for (int k = 0; k < 8; k++) tA[k] = R.a(k);
#pragma unroll
    for (int q = 0; q < 8; q++) sT[rd + q] = x[q];              // element (k0, k1, p0)
    tile_sync<WSYNC>();
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = sT[blk + 9 * k];         // lane = (k0, p0)
    tile_sync<WSYNC>();
#pragma unroll
    for (int k = 1; k < 8; k++) x[k] = cmulx<true>(x[k], tB[k]);
    dft8<true>(x);  // k1 -> p1 ; lane = 8*k0 + p0
#pragma unroll
    for (int p1 = 0; p1 < 8; p1++) sT[blk + 9 * p1] = x[p1];    // element (k0, p1, p0)
    tile_sync<WSYNC>();
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) x[k0] = sT[own + 72 * k0];   // lane = (p1, p0)
    tile_sync<WSYNC>();
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cmulx<true>(x[k], tA[k]);  // conj(tL * w1^k0)
    dft8_untwist_inv(x);  // k0 -> r, e^{-i pi r/16} included up to untwist_gain(r)
}

// Two inverse transforms (the lo and hi limb sums of one output polynomial) interleaved in one
// instruction stream through ONE tile: the DS instructions of a wave execute in order, so as
// long as each [write, read] pair of one transform is issued whole, the other transform's
// butterflies run while that round trip is in flight.  (Alone, a wave spends ~2/3 of a
// transform waiting on its four LDS round trips.)
template <bool WSYNC>
__device__ __forceinline__ void fft512_inverse_pair(double2 (&x)[8], double2 (&y)[8], double2* sT, int lane,
                                                    const LaneRoots& R) {
    const int hi = lane >> 3, lo = lane & 7;
    const int own = hi * 9 + lo, blk = hi * 72 + lo, rd = hi * 72 + lo * 9;
    double2 tA[8], tB[8];
#pragma unroll
    for (int k = 1; k < 8; k++) tB[k] = R.b(k);
    dft8<true>(x);
#pragma unroll
    for (int q = 0; q < 8; q++) sT[rd + q] = x[q];
    tile_sync<WSYNC>();
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = sT[blk + 9 * k];          // x round trip 1 in flight ...
    dft8<true>(y);                                                // ... under y's first pass
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; k++) tA[k] = R.a(k);
#pragma unroll
    for (int q = 0; q < 8; q++) sT[rd + q] = y[q];               // issued after x's reads: in-order LDS keeps them apart
    tile_sync<WSYNC>();
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = sT[blk + 9 * k];          // y round trip 1 ...
#pragma unroll
    for (int k = 1; k < 8; k++) x[k] = cmulx<true>(x[k], tB[k]);
    dft8<true>(x);                                                // ... under x's second pass
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p1 = 0; p1 < 8; p1++) sT[blk + 9 * p1] = x[p1];
    tile_sync<WSYNC>();
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) x[k0] = sT[own + 72 * k0];    // x round trip 2 ...
#pragma unroll
    for (int k = 1; k < 8; k++) y[k] = cmulx<true>(y[k], tB[k]);
    dft8<true>(y);                                                // ... under y's second pass
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p1 = 0; p1 < 8; p1++) sT[blk + 9 * p1] = y[p1];
    tile_sync<WSYNC>();
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) y[k0] = sT[own + 72 * k0];    // y round trip 2 ...
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cmulx<true>(x[k], tA[k]);
    dft8_untwist_inv(x);                                          // ... under x's last pass
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = cmulx<true>(y[k], tA[k]);
    dft8_untwist_inv(y);
    tile_sync<WSYNC>();  // the tile may be reused by the caller
}

// Rounds one inverse-transformed coefficient: x * gain + 1.5 * 2^52 carries round(x * gain) in its low mantissa bits.  watch:
// the product and the sum are separate operations and the distance to the nearest integer is folded into dev_max (the
// one-limb kernels' rounding guard); otherwise one FMA.
constexpr double kMagic52 = 6755399441055744.0;  // 1.5 * 2^52
__device__ __forceinline__ uint32_t round_coef(double v, double gain, bool watch, double& dev_max) {
    double t;
    if (watch) {  // a compile-time constant at every call site once the register loop is unrolled
        const double z = v * gain;
        t = z + kMagic52;
        dev_max = fmax(dev_max, fabs(z - (t - kMagic52)));
    } else {
        t = fma(v, gain, kMagic52);
    }
    return (uint32_t)__double2loint(t);
}

// One 8-register block [8][64] double2 of the BK spectrum through the buffer path: resource and byte offset in SGPRs,
// the lane's 16 bytes as the only vector operand, the register index as the instruction's immediate (0-3 KiB) -- no
// per-load 64-bit vector address arithmetic (global_load needs ~12 v_add_co / v_addc per row of two blocks: measured +2.7 %
// for the one-wave-per-gate kernel, profiles/r2_j_variants.txt).
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_bk_block(double2 (&dst)[8], __amdgpu_buffer_rsrc_t rsrc, int lane16, int soff) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const v4i_t d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16 + (k & 3) * 1024, soff + (k >> 2) * 4096, 0);
        dst[k] = make_double2(__hiloint2double(d.y, d.x), __hiloint2double(d.w, d.z));
    }
}

}  // namespace
}  // namespace w64
}  // namespace ieache
