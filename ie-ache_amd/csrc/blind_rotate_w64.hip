// Blind rotation for the N=1024 ring (tfhe_blindRotate_FFT of libtfhe), on 64-lane wavefronts.
//
// Same mathematics as k_blind_rotate_generic, re-laid out for CDNA4:
//   * the 512-point complex transform is 8 x 8 x 8: three radix-8 passes done entirely in registers (8 points per lane),
//     separated by two register<->lane transposes -- through a padded, bank-conflict-free LDS tile per wave, or cross-lane
//     with v_permlane*_swap / v_cndmask_b32_dpp (the forward lane-high one, by default);
//   * the spectrum stays in registers between the forward transform, the point-wise multiply-accumulate with BK_i and the
//     inverse transform; BK_i is stored in exactly the (register, lane) order the forward transform leaves its output in, so
//     every BK load is one coalesced 1 KiB load per wave, issued ahead of its MACs;
//   * the CMux step index is the OUTER loop of the evaluator: one launch advances every gate of a chunk by a slice of steps,
//     so all resident workgroups read the same BK blocks while they are hot in the XCD's L2;
//   * the accumulator (2 x 1024 int32) lives in LDS only because the X^a rotation needs arbitrary shifts; twiddles come from a
//     9 KiB LDS table.
// One kernel per launch-size regime (variant table in blind_rotate_w64.h):
//   k_blind_rotate_w1b    one wave per gate            launches of more than 5 gates per CU (the throughput kernel)
//   k_blind_rotate_w2r    two waves per gate           2 .. 5 gates per CU
//   k_blind_rotate_w4r    four waves per gate          1 .. 2 gates per CU
//   k_blind_rotate_wide4  2L = 6 waves per gate        at most one gate per CU (single expressions: the reference's own mode)
// all on the ONE-limb spectrum (libtfhe's own product, rounded to the exact integer under a rounding guard and a sampled
// audit), and k_blind_rotate_w2 / k_blind_rotate_wide<..., 2> on the TWO-limb spectrum (exact by construction: "exact_fft",
// repeats, audits).  Earlier kernels and measured dead ends stay selectable as A/B partners.
#include "blind_rotate_w64.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <type_traits>

namespace ieache {
namespace w64 {

using namespace dev;

namespace {

constexpr int kN = 1024, kM = 512;
constexpr double kR = 0.70710678118654752440;  // 1/sqrt(2)

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
// The same table read from GLOBAL memory through the buffer path (SGPR resource, one per-lane byte offset, the index as the
// instruction's immediate): twiddle reads leave the LDS pipe for the vector-memory one.  FROM: 1 = only the first set
// (a, 8 per transform), 2 = both; the rest comes from the LDS copy.
typedef int v4i_tw __attribute__((ext_vector_type(4)));
template <int FROM>
struct BufRoots {
    LaneRoots lds;
    __amdgpu_buffer_rsrc_t rsrc;
    int off1, off2;  // byte offsets of tw[lane] and tw[512 + (lane & 7)]
    __device__ __forceinline__ double2 a(int k) const {
        const v4i_tw d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off1, k * 1024, 0);
        return make_double2(__hiloint2double(d.y, d.x), __hiloint2double(d.w, d.z));
    }
    __device__ __forceinline__ double2 b(int k) const {
        if (FROM < 2) return lds.b(k);
        const v4i_tw d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off2, k * 128, 0);
        return make_double2(__hiloint2double(d.y, d.x), __hiloint2double(d.w, d.z));
    }
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
__device__ __forceinline__ void swap_dwords(unsigned& lo, unsigned& hi, int lane) {
    // lo: a dword of x[r] (register bit clear), hi: the same dword of x[r | bit].
    // After the call, lanes with bit B clear hold in `hi` what the partner lane had in `lo`, and
    // lanes with bit B set hold in `lo` what the partner had in `hi`.
    if constexpr (B == 5) {
        const v2u_t r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0];
        hi = r[1];
    } else if constexpr (B == 4) {
        const v2u_t r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
        lo = r[0];
        hi = r[1];
    } else if constexpr (B == 3) {
        const unsigned nh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)lo, 0x128, 0xF, 0x3, false);  // row_ror:8
        const unsigned nl = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)hi, 0x128, 0xF, 0xC, false);
        lo = nl;
        hi = nh;
    } else if constexpr (B == 2) {
        const unsigned nh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)lo, 0x104, 0xF, 0x5, false);  // row_shl:4
        const unsigned nl = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)hi, 0x114, 0xF, 0xA, false);  // row_shr:4
        lo = nl;
        hi = nh;
    } else {
        constexpr int ctrl = B == 1 ? 0x4E : 0xB1;  // quad_perm [2,3,0,1] / [1,0,3,2]
        const unsigned tl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, 0xF, 0xF, true);
        const unsigned th = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, 0xF, 0xF, true);
        const bool set = (lane >> B) & 1;
        hi = set ? hi : tl;
        lo = set ? th : lo;
    }
}

// Lane bit 3 has no swap instruction.  As two DPP moves with bank masks the exchange costs a register copy on top (the
// second move needs the first one's overwritten source): 3 instructions per dword pair.  v_cndmask_b32 takes a DPP source
// itself -- new_hi = set ? hi : lo[lane ^ 8], new_lo = set ? hi[lane ^ 8] : lo with set = lane bit 3, VCC flipped in between
// by the scalar unit -- 2 per pair, into fresh registers.  Four dwords (one double2) per block; the s_nop covers the
// VALU-write -> DPP-read wait states the assembler does not insert inside an asm block.
#define IEACHE_SWAP4_DPP(MASK, CTRL_HI, CTRL_LO, BC)                                                              \
    asm volatile("s_mov_b32 vcc_lo, " MASK "\n\t"                                                                \
                 "s_mov_b32 vcc_hi, " MASK "\n\t"                                                                \
                 "s_nop 1\n\t"                                                                                   \
                 "v_cndmask_b32_dpp %4, %8, %12, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf" BC "\n\t"                  \
                 "v_cndmask_b32_dpp %5, %9, %13, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf" BC "\n\t"                  \
                 "v_cndmask_b32_dpp %6, %10, %14, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf" BC "\n\t"                 \
                 "v_cndmask_b32_dpp %7, %11, %15, vcc " CTRL_HI " row_mask:0xf bank_mask:0xf" BC "\n\t"                 \
                 "s_not_b64 vcc, vcc\n\t"                                                                        \
                 "v_cndmask_b32_dpp %0, %12, %8, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf" BC "\n\t"                  \
                 "v_cndmask_b32_dpp %1, %13, %9, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf" BC "\n\t"                  \
                 "v_cndmask_b32_dpp %2, %14, %10, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf" BC "\n\t"                 \
                 "v_cndmask_b32_dpp %3, %15, %11, vcc " CTRL_LO " row_mask:0xf bank_mask:0xf" BC                    \
                 : "=&v"(na0), "=&v"(na1), "=&v"(na2), "=&v"(na3), "=&v"(nb0), "=&v"(nb1), "=&v"(nb2), "=&v"(nb3)   \
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])           \
                 : "vcc")
// new_b = set ? b : a[partner], new_a = set ? b[partner] : a, set = lane bit B, partner = lane ^ (1 << B); lanes whose
// DPP source falls outside the row (bit 2: row_shl / row_shr by 4) take the other operand anyway, but WITHOUT bound_ctrl
// such a lane is not written at all
template <int B>
__device__ __forceinline__ void swap4_dpp(unsigned (&a)[4], unsigned (&b)[4]) {
    unsigned na0, na1, na2, na3, nb0, nb1, nb2, nb3;
    if constexpr (B == 3) {
        IEACHE_SWAP4_DPP("0xff00ff00", "row_ror:8", "row_ror:8", "");
    } else if constexpr (B == 2) {
        IEACHE_SWAP4_DPP("0xf0f0f0f0", "row_shl:4", "row_shr:4", " bound_ctrl:0");  // out-of-row sources read 0 and are not selected
    } else if constexpr (B == 1) {
        IEACHE_SWAP4_DPP("0xcccccccc", "quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]", "");
    } else {
        IEACHE_SWAP4_DPP("0xaaaaaaaa", "quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]", "");
    }
    a[0] = na0, a[1] = na1, a[2] = na2, a[3] = na3;
    b[0] = nb0, b[1] = nb1, b[2] = nb2, b[3] = nb3;
}

// SW3 = 0: round 2's bit-3 exchange (two DPP moves and a copy), kept for k_blind_rotate_w1 as the A/B partner
template <int B, int SW3 = 1>
__device__ __forceinline__ void bitswap(double2 (&x)[8], int lane) {
    constexpr int m = 1 << (B % 3);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r & m) continue;
        unsigned a[4] = {(unsigned)__double2loint(x[r].x), (unsigned)__double2hiint(x[r].x), (unsigned)__double2loint(x[r].y),
                         (unsigned)__double2hiint(x[r].y)};
        unsigned b[4] = {(unsigned)__double2loint(x[r | m].x), (unsigned)__double2hiint(x[r | m].x), (unsigned)__double2loint(x[r | m].y),
                         (unsigned)__double2hiint(x[r | m].y)};
        if constexpr (B <= 3 && SW3 == 1) {
            swap4_dpp<B>(a, b);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) swap_dwords<B>(a[q], b[q], lane);
        }
        x[r] = make_double2(__hiloint2double((int)a[1], (int)a[0]), __hiloint2double((int)a[3], (int)a[2]));
        x[r | m] = make_double2(__hiloint2double((int)b[1], (int)b[0]), __hiloint2double((int)b[3], (int)b[2]));
    }
}
// register index <-> lane bits 3..5 (what the first LDS transpose does)
template <int SW3 = 1>
__device__ __forceinline__ void xlane_hi(double2 (&x)[8], int lane) {
    bitswap<3, SW3>(x, lane);
    bitswap<4>(x, lane);
    bitswap<5>(x, lane);
}
// register index <-> lane bits 0..2 (what the second LDS transpose does)
template <int SW3 = 1>
__device__ __forceinline__ void xlane_lo(double2 (&x)[8], int lane) {
    bitswap<0, SW3>(x, lane);
    bitswap<1, SW3>(x, lane);
    bitswap<2, SW3>(x, lane);
}

// Transpose tiles hold element (h, m, l) -- three 3-bit digits -- at h*72 + m*9 + l.
// The 9/72 padding makes every ds_write_b128 / ds_read_b128 of both transposes
// bank-conflict free AND lets each access be "per-lane base + immediate offset".
constexpr int kTile = 8 * 72;  // double2 elements per tile (9216 B)

// Forward 512-point transform of the twisted polynomial.
//   in : x[r] = y_{64r+lane} * exp(i*pi*r/16)  (the lane part tL of the twist is applied here)
//   out: x[k2] = X[k0 + 8*k1 + 64*k2] with lane = 8*k0 + k1
// ILV: issue each ds_write right behind the multiply that produces its data instead of as a burst of 8 after
// all of them (left alone the compiler clusters the stores): the wave's own LDS-store issue (~13 cycles each on
// CDNA4) then runs under its next twiddle multiply instead of stalling its instruction stream.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// MID: called once the first inter-pass twiddles are consumed (their 32 VGPRs are free from there on): the place to
// request data the caller needs right after the transform
// POST: called once the reads of the last (lane-low) transpose are issued and before their data is used: work that does not
// depend on them runs under that LDS round trip
// TBF: keep the second twiddle set's loads behind the first set's multiplies (a scheduling fence): the compiler otherwise
// hoists them, and a caller that holds 128 registers of spectrum sums (k_blind_rotate_x1) cannot afford both sets live at once
template <bool WSYNC, int XLANE = 0, int ILV = 0, class MID = NoHook, bool MID_LATE = false, int SW3 = 1, class POST = NoHook, class ROOTS = LaneRoots, bool TBF = false>
__device__ __forceinline__ void fft512_forward(double2 (&x)[8], double2* sT, int lane, const ROOTS& R, MID mid = MID(), POST post = POST()) {
    const int hi = lane >> 3, lo = lane & 7;
    const int own = hi * 9 + lo;   // (m, l) = (lane>>3, lane&7) inside a row-block h
    const int blk = hi * 72 + lo;  // (h, l) = (lane>>3, lane&7)
    // twiddles are fetched from the LDS table ahead of the butterflies that hide their latency
    double2 tA[8], tB[8];
#pragma unroll
    for (int k = 0; k < 8; k++) tA[k] = R.a(k);
    dft8<false>(x);                          // over r -> k0
    if (ILV && !(XLANE & 1)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x[k] = cmulx<false>(x[k], tA[k]);
            sT[own + 72 * k] = x[k];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 1; k < 8; k++) tB[k] = R.b(k);
        tile_sync<WSYNC>();
#pragma unroll
        for (int p1 = 0; p1 < 8; p1++) x[p1] = sT[blk + 9 * p1];
        tile_sync<WSYNC>();
        dft8<false>(x);
        sT[blk] = x[0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 1; k < 8; k++) {
            x[k] = cmulx<false>(x[k], tB[k]);
            sT[blk + 9 * k] = x[k];
            __builtin_amdgcn_sched_barrier(0);
        }
        tile_sync<WSYNC>();
        const int rd2 = hi * 72 + lo * 9;
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = sT[rd2 + q];
        tile_sync<WSYNC>();
        dft8<false>(x);
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cmulx<false>(x[k], tA[k]);  // * tL * w512^(lane*k0)
    if (!std::is_same<MID, NoHook>::value && !MID_LATE) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (TBF) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 1; k < 8; k++) tB[k] = R.b(k);
    if (XLANE & 1) {
        xlane_hi<SW3>(x, lane);                                     // reg k0 <-> lane bits 3..5: lane = (k0, p0), reg = p1
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
    if (!std::is_same<MID, NoHook>::value && MID_LATE) {  // ... or once the second twiddles are consumed too
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (XLANE & 2) {
        xlane_lo<SW3>(x, lane);                                     // reg k1 <-> lane bits 0..2: lane = (k0, k1), reg = p0
    } else {
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) sT[blk + 9 * k1] = x[k1];    // element (k0, k1, p0), lane = (k0, p0)
        tile_sync<WSYNC>();
        const int rd = hi * 72 + lo * 9;                            // lane = (k0, k1)
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = sT[rd + q];
        if (!std::is_same<POST, NoHook>::value) {
            __builtin_amdgcn_sched_barrier(0);
            post();
            __builtin_amdgcn_sched_barrier(0);
        }
        tile_sync<WSYNC>();
    }
    dft8<false>(x);                          // over p0 -> k2 ; lane = 8*k0 + k1
}

// Inverse of fft512_forward (unnormalised: 512 x), also removing the lane part of the twist:
//   in : spectrum in the layout fft512_forward produces
//   out: x[r] = y_{64r+lane} * exp(i*pi*r/16)   (caller multiplies by exp(-i*pi*r/16))
template <bool WSYNC, int XLANE = 0, class ROOTS = LaneRoots>
__device__ __forceinline__ void fft512_inverse(double2 (&x)[8], double2* sT, int lane, const ROOTS& R) {
    const int hi = lane >> 3, lo = lane & 7;
    const int own = hi * 9 + lo, blk = hi * 72 + lo, rd = hi * 72 + lo * 9;
    double2 tA[8], tB[8];
#pragma unroll
    for (int k = 1; k < 8; k++) tB[k] = R.b(k);
    dft8<true>(x);  // k2 -> p0 ; lane = (k0, k1)
#pragma unroll
    for (int k = 0; k < 8; k++) tA[k] = R.a(k);
    if (XLANE & 2) {
        xlane_lo(x, lane);                                          // back to lane = (k0, p0), reg = k1
    } else {
#pragma unroll
        for (int q = 0; q < 8; q++) sT[rd + q] = x[q];              // element (k0, k1, p0)
        tile_sync<WSYNC>();
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = sT[blk + 9 * k];         // lane = (k0, p0)
        tile_sync<WSYNC>();
    }
#pragma unroll
    for (int k = 1; k < 8; k++) x[k] = cmulx<true>(x[k], tB[k]);
    dft8<true>(x);  // k1 -> p1 ; lane = 8*k0 + p0
    if (XLANE & 1) {
        xlane_hi(x, lane);                                          // back to lane = (p1, p0), reg = k0
    } else {
#pragma unroll
        for (int p1 = 0; p1 < 8; p1++) sT[blk + 9 * p1] = x[p1];    // element (k0, p1, p0)
        tile_sync<WSYNC>();
#pragma unroll
        for (int k0 = 0; k0 < 8; k0++) x[k0] = sT[own + 72 * k0];   // lane = (p1, p0)
        tile_sync<WSYNC>();
    }
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = cmulx<true>(x[k], tA[k]);  // conj(tL * w1^k0); 1/512 is in untwist_reg()
    dft8<true>(x);  // k0 -> r
}

// Two inverse transforms (the lo and hi limb sums of one output polynomial) interleaved in one
// instruction stream through ONE tile: the DS instructions of a wave execute in order, so as
// long as each [write, read] pair of one transform is issued whole, the other transform's
// butterflies run while that round trip is in flight.  (Alone, a wave spends ~2/3 of a
// transform waiting on its four LDS round trips.)
template <bool WSYNC, class ROOTS = LaneRoots>
__device__ __forceinline__ void fft512_inverse_pair(double2 (&x)[8], double2 (&y)[8], double2* sT, int lane,
                                                    const ROOTS& R) {
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
    dft8<true>(x);                                                // ... under x's last pass
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = cmulx<true>(y[k], tA[k]);
    dft8<true>(y);
    tile_sync<WSYNC>();  // the tile may be reused by the caller
}

// exp(i*pi*r/16), r = 0..7: the register part of the twist
__device__ __forceinline__ double2 twist_reg(int r) {
    constexpr double C[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                             0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                             0.19509032201612826785};
    constexpr double S[8] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                             0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                             0.98078528040323044913};
    return make_double2(C[r], S[r]);
}

// exp(i*pi*r/16) / 512: multiplying by its conjugate removes the register part of the
// twist and normalises the inverse transform (exact: a power-of-two scale)
__device__ __forceinline__ double2 untwist_reg(int r) {
    const double2 t = twist_reg(r);
    return make_double2(t.x * (1.0 / 512.0), t.y * (1.0 / 512.0));
}

// ---- key preparation: BK polynomial -> two-limb spectrum in the w64 register/lane order ----
// bkf layout: [n][2L rows][q = 2*c + limb][k2 = 8][lane = 64]
__global__ __launch_bounds__(64) void k_bk_to_spectrum_w64(const Torus32* bk_raw, double2* bkf) {
    __shared__ __align__(16) double2 sT[kTile];
    __shared__ __align__(16) double2 sTw[kTwElems];
    const int lane = threadIdx.x;
    build_twiddles(sTw, lane, 64);
    __syncthreads();
    const LaneRoots R = make_roots(sTw, lane);
    const size_t poly = blockIdx.x;  // (i * 2L + row) * 2 + c
    const Torus32* src = bk_raw + poly * kN;
    const size_t irow = poly >> 1, c = poly & 1;
#pragma unroll 1
    for (int limb = 0; limb < 2; limb++) {
        double2 x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int32_t v0 = src[64 * r + lane], v1 = src[64 * r + lane + kM];
            const int32_t lo0 = (int16_t)(v0 & 0xFFFF), lo1 = (int16_t)(v1 & 0xFFFF);
            const int32_t e0 = limb ? (int32_t)(((int64_t)v0 - lo0) >> 16) : lo0;
            const int32_t e1 = limb ? (int32_t)(((int64_t)v1 - lo1) >> 16) : lo1;
            x[r] = r == 0 ? make_double2((double)e0, (double)e1)  // exp(0) = 1: nothing to multiply
                            : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
        }
        fft512_forward<true>(x, sT, lane, R);
        double2* dst = bkf + ((irow * 4 + c * 2 + limb) * 8) * 64 + lane;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) dst[k2 * 64] = x[k2];
    }
}

// ---- key preparation, one-limb form: BK polynomial -> spectrum of its 32-bit coefficients ----
// bkf1 layout: [n][2L rows][c = 2][k2 = 8][lane = 64]   (half the two-limb form: 61.9 MB at n = 630)
__global__ __launch_bounds__(64) void k_bk_to_spectrum_w64_1(const Torus32* bk_raw, double2* bkf1) {
    __shared__ __align__(16) double2 sT[kTile];
    __shared__ __align__(16) double2 sTw[kTwElems];
    const int lane = threadIdx.x;
    build_twiddles(sTw, lane, 64);
    __syncthreads();
    const LaneRoots R = make_roots(sTw, lane);
    const size_t poly = blockIdx.x;  // (i * 2L + row) * 2 + c
    const Torus32* src = bk_raw + poly * kN;
    double2 x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double2 v = make_double2((double)src[64 * r + lane], (double)src[64 * r + lane + kM]);
        x[r] = r == 0 ? v : cmulx<false>(v, twist_reg(r));
    }
    fft512_forward<true>(x, sT, lane, R);
    double2* dst = bkf1 + (poly * 8) * 64 + lane;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) dst[k2 * 64] = x[k2];
}

// ---- K0..K2: gate pre-combination, mod-switch, test-vector init ----
// One 128-thread workgroup per gate instance.  Writes the rotation amounts
// bara[n] (u16, row stride nb) and the initial accumulator [2][1024] to the
// blind-rotation state in HBM, from where the sliced kernel below picks up.
__global__ __launch_bounds__(128) void k_br_prologue(DevKeys K, WorkDesc W, uint16_t* st_bara, int32_t nb, int32_t* st_acc) {
    __shared__ int32_t s_barb;
    const int tid = threadIdx.x;
    const int32_t n = K.n;
    const int64_t item = (int64_t)blockIdx.x;
    const GateInst g = resolve(W, W.item0 + item, K.stride);
    uint16_t* bara = st_bara + (size_t)item * nb;
    for (int32_t i = tid; i <= n; i += 128) {
        const int32_t bar = modswitch2N(combined_coef(g, i, n), 11);
        if (i < n)
            bara[i] = (uint16_t)bar;
        else
            s_barb = bar;
    }
    __syncthreads();
    // acc = (0, X^{2N-barb} * (mu,...,mu))
    const int32_t a0 = (2 * kN - s_barb) & (2 * kN - 1);
    int32_t* acc = st_acc + (size_t)item * 2 * kN;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int32_t j = 128 * r + tid;
        acc[j] = 0;
        acc[kN + j] = ((j - a0) & (2 * kN - 1)) < kN ? kMU : -kMU;
    }
}

// One 8-register block [8][64] double2 of the BK spectrum through the buffer path: resource and byte offset in SGPRs,
// the lane's 16 bytes as the only vector operand, the register index as the instruction's immediate (0-3 KiB) -- no
// per-load 64-bit vector address arithmetic (global_load needs ~12 v_add_co / v_addc per row of two blocks).
// Measured: +2.7 % for k_blind_rotate_w1 (vector-issue bound; br_variant 30 = the same kernel with global_load), neutral
// for k_blind_rotate_w2s, -9 % for the two-limb k_blind_rotate_w2 and -13 % for the latency kernel (2.78 -> 3.20 ms), which
// therefore keep global_load.
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <bool BUF = true>
__device__ __forceinline__ void load_bk_block(double2 (&dst)[8], __amdgpu_buffer_rsrc_t rsrc, int lane16, int soff,
                                              const double2* __restrict__ base = nullptr) {
    if (!BUF) {  // the same block with per-lane 64-bit addresses (global_load_dwordx4): the A/B partner of the measurement
        const double2* p = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + soff + lane16);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = p[k * 64];
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const v4i_t d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16 + (k & 3) * 1024, soff + (k >> 2) * 4096, 0);
        dst[k] = make_double2(__hiloint2double(d.y, d.x), __hiloint2double(d.w, d.z));
    }
}

// ---- K3 (+K4): CMux steps [i0, i1) for every gate instance of the launch ----
// One 128-thread workgroup (two waves) per gate instance.  The step index is the
// OUTER loop of the evaluator: a chunk of gates is advanced S steps per launch, so
// all resident workgroups read the same S blocks BK_i0..BK_i1 while those are hot
// in the XCD's L2 (with the whole blind rotation in one launch, workgroups drift
// apart in i and each streams its own BK_i from Infinity Cache/HBM: measured 51 MB
// of fetch per gate).  The accumulator lives in HBM between launches (8 KB per gate).
// Wave w decomposes accumulator polynomial w (its 3 digit rows -> 3 forward
// transforms), owns the spectrum-domain sums of OUTPUT polynomial w (both
// limbs) and inverse-transforms them.  Each forward spectrum is handed to the
// partner wave through the producing wave's own (then idle) transpose tile.
// dynamic LDS: sT [2][kTile] double2 | tw [kTwElems] double2 | acc [2][1024] int32
template <int L, int BGBIT, bool DIAG, bool WSYNC, int XLANE = 0, int ILV = 0, bool ENDBAR = false, bool PAIRINV = false>
__global__ __launch_bounds__(128, 2) void k_blind_rotate_w2(DevKeys K, const double2* __restrict__ bkf,
                                                           const uint16_t* __restrict__ st_bara, int32_t nb,
                                                           int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                           unsigned long long* diag, const double2* __restrict__ gtw) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2* sT_all = reinterpret_cast<double2*>(smem);
    double2* sTw = sT_all + 2 * kTile;
    int32_t* acc = reinterpret_cast<int32_t*>(sTw + kTwElems);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2* sT = sT_all + wave * kTile;
    const double2* sTp = sT_all + (wave ^ 1) * kTile;
    const int64_t item = (int64_t)blockIdx.x;
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, 128);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[128 * r + tid] = src[128 * r + tid];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52: (x + magic) carries round(x) in its low mantissa bits
    int32_t* accw = acc + wave * kN;  // the polynomial this wave decomposes and updates
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if (DIAG) tlast = stamp();
#define IEACHE_STAMP(idx)                   \
    if (DIAG) {                             \
        const unsigned long long t_ = stamp(); \
        tsum[idx] += t_ - tlast;            \
        tlast = t_;                         \
    }

    // this slice's rotation amounts: one per lane, fetched once, then read with readlane
    // (a dependent global load at the head of every step costs ~2-3k cycles)
    const int32_t my_a = (i0 + lane < i1) ? (int32_t)bara[i0 + lane] : 0;
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readlane(my_a, i - i0);
        if (a == 0) continue;  // workgroup-uniform; exact arithmetic makes the step a no-op
        // BK_i rows [2L][4][8][64]; this wave reads outputs o = 2*wave, 2*wave+1 of every row
        const double2* __restrict__ bki = bkf + (size_t)i * (2 * L * 4 * kM) + (size_t)(2 * wave) * kM + lane;
        double2 s[2][8];  // written (not accumulated into) by digit 0's own-row products below
        // (X^a - 1) * acc_w at this lane's 16 coefficients, plus the decomposition offset
        uint32_t v0[8], v1[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int32_t j = 64 * r + lane;
            // (x + C) ^ C with C = sum_q halfBg << shift_q: digit q's field then holds digit ^ halfBg, whose
            // sign-extended BGBIT-bit value IS digit - halfBg (one v_bfe_i32 per digit below)
            v0[r] = (((uint32_t)rot_coef(accw, j, a, kN) - (uint32_t)accw[j]) + dec_offset) ^ dec_offset;
            v1[r] = (((uint32_t)rot_coef(accw, j + kM, a, kN) - (uint32_t)accw[j + kM]) + dec_offset) ^ dec_offset;
        }
        IEACHE_STAMP(0)
        auto digit_row = [&](const int q, auto first) {
            constexpr bool FIRST = decltype(first)::value;  // digit 0: its own-row products initialise s
            const int sh = 32 - (q + 1) * BGBIT;
            const double2* __restrict__ bown = bki + (size_t)(wave * L + q) * (4 * kM);
            const double2* __restrict__ bpar = bki + (size_t)((wave ^ 1) * L + q) * (4 * kM);
            double2 x[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);  // v_bfe_i32
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)  // exp(0) = 1: nothing to multiply
                            : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            // BK loads are issued well ahead of their use (each is an L2 round trip of ~700 cycles):
            //   bA = own row, limb 0      before the transform
            //   bB = own row, limb 1      } right after it, consumed behind bA's MACs
            //   bC = partner row, limb 0  }
            //   bD = partner row, limb 1  after bA is consumed (reuses its registers)
            double2 bA[8], bB[8], bC[8];
#pragma unroll
            for (int k = 0; k < 8; k++) bA[k] = bown[k * 64];
            __builtin_amdgcn_sched_barrier(0);
            fft512_forward<WSYNC, XLANE, ILV>(x, sT, lane, R);
            IEACHE_STAMP(1)
            // hand the spectrum to the partner wave through our own (now idle) tile
#pragma unroll
            for (int k = 0; k < 8; k++) sT[k * 64 + lane] = x[k];
#pragma unroll
            for (int k = 0; k < 8; k++) bB[k] = bown[(8 + k) * 64];
#pragma unroll
            for (int k = 0; k < 8; k++) bC[k] = bpar[k * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[0][k] = FIRST ? cmulx<false>(x[k], bA[k])
                                : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[0][k].x)),
                                               fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[0][k].y)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++) bA[k] = bpar[(8 + k) * 64];  // bD
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[1][k] = FIRST ? cmulx<false>(x[k], bB[k])
                                : make_double2(fma(x[k].x, bB[k].x, fma(-x[k].y, bB[k].y, s[1][k].x)),
                                               fma(x[k].x, bB[k].y, fma(x[k].y, bB[k].x, s[1][k].y)));
            IEACHE_STAMP(2)
            __syncthreads();
            IEACHE_STAMP(3)
            // partner's row
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = sTp[k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[0][k] = make_double2(fma(x[k].x, bC[k].x, fma(-x[k].y, bC[k].y, s[0][k].x)),
                                       fma(x[k].x, bC[k].y, fma(x[k].y, bC[k].x, s[0][k].y)));
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[1][k] = make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[1][k].x)),
                                       fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[1][k].y)));
            IEACHE_STAMP(4)
            __syncthreads();  // partner has read our tile before the next transform reuses it
            IEACHE_STAMP(5)
        };
        digit_row(0, std::true_type{});
#pragma unroll 1
        for (int q = 1; q < L; q++) digit_row(q, std::false_type{});
        // back to coefficients, round, recombine the two limbs, accumulate into polynomial `wave`
        if (XLANE == 0 || PAIRINV) {  // PAIRINV: cross-lane transposes in the forward transforms only
            fft512_inverse_pair<WSYNC>(s[0], s[1], sT, lane, R);
        } else {
            fft512_inverse<WSYNC, XLANE>(s[0], sT, lane, R);
            fft512_inverse<WSYNC, XLANE>(s[1], sT, lane, R);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double2 zl = r == 0 ? make_double2(s[0][0].x * (1.0 / 512.0), s[0][0].y * (1.0 / 512.0)) : cmulx<true>(s[0][r], untwist_reg(r));
            const double2 zh = r == 0 ? make_double2(s[1][0].x * (1.0 / 512.0), s[1][0].y * (1.0 / 512.0)) : cmulx<true>(s[1][r], untwist_reg(r));
            const uint32_t l0 = (uint32_t)__double2loint(zl.x + kMagic), l1 = (uint32_t)__double2loint(zl.y + kMagic);
            const uint32_t h0 = (uint32_t)__double2loint(zh.x + kMagic), h1 = (uint32_t)__double2loint(zh.y + kMagic);
            const int32_t j = 64 * r + lane;
            accw[j] = (int32_t)((uint32_t)accw[j] + l0 + (h0 << 16));
            accw[j + kM] = (int32_t)((uint32_t)accw[j + kM] + l1 + (h1 << 16));
        }
        IEACHE_STAMP(6)
        // wave w reads and updates only polynomial w, and the partner is done with this wave's tile since the
        // last digit's second barrier: nothing crosses waves here, so no barrier (ENDBAR = true, variant 11, keeps
        // round 1's; measured 0.3-0.4 % slower)
        if (ENDBAR) __syncthreads();
        IEACHE_STAMP(7)
    }
    if (!ENDBAR) __syncthreads();  // the epilogue below reads both polynomials with all threads
#undef IEACHE_STAMP
    if (DIAG && diag && lane == 0) {
#pragma unroll
        for (int t = 0; t < 8; t++) atomicAdd(&diag[wave * 8 + t], tsum[t]);
    }
    if (ext) {
        // K4: sample extract after the last slice
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += 128)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[128 * r + tid] = src[128 * r + tid];
    }
}


// ---- K3 (+K4), throughput form: ONE wave per gate instance, one-limb spectrum ----
// The two-limb transform above is exact by construction (every rounded sum stays below 2^37 of the 2^53 an FP64
// mantissa holds) and pays for it with a second inverse transform and a second set of row products per output
// polynomial.  libtfhe itself multiplies with ONE double-precision transform of the 32-bit coefficients; the sums
// then reach 2^49.6 in the worst case and ~2^43 on real data, where the transform's rounding error is ~2^-9 of an
// integer step (largest seen in a whole bench run: 0.014) -- far from the 0.5 that would change a rounded
// coefficient, but not provably so.  This kernel takes that form and WATCHES the error: every inverse-transformed
// coefficient's distance to the nearest integer is folded into a running maximum, published per launch (guard[1],
// float bits) and counted (guard[0]) when it exceeds kGuardLimit; the evaluator then repeats the call on the
// two-limb kernel.  With 6 forward + 2 inverse transforms and 12 row products per step (10 + 24 before) the whole
// step fits ONE wave: no spectra cross waves, so the step has no workgroup barrier at all, and a SIMD's two
// resident waves belong to unrelated gates that never wait for each other.
// Four gates share a workgroup only for the twiddle table.
// dynamic LDS: sT [4][kTile] double2 | tw [kTwElems] double2 | acc [4][2][1024] int32     (78 848 B -> 2 per CU)
constexpr int kW1Gates = 4;
constexpr float kGuardLimit = 0.0625f;
template <int L, int BGBIT, bool GUARD, int XLANE = 1, int EARLYB = 0, bool BUF = true>
__global__ __launch_bounds__(64 * kW1Gates, 2) void k_blind_rotate_w1(DevKeys K, const double2* __restrict__ bkf1,
                                                                      const uint16_t* __restrict__ st_bara, int32_t nb,
                                                                      int32_t* st_acc, int64_t items, int32_t i0, int32_t i1,
                                                                      Torus32* ext, unsigned* guard,
                                                                      const double2* __restrict__ gtw) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2* sT_all = reinterpret_cast<double2*>(smem);
    double2* sTw = sT_all + kW1Gates * kTile;
    int32_t* acc_all = reinterpret_cast<int32_t*>(sTw + kTwElems);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    int32_t* acc = acc_all + wave * 2 * kN;
    const int64_t item = (int64_t)blockIdx.x * kW1Gates + wave;
    load_twiddles(sTw, gtw, tid, 64 * kW1Gates);
    __syncthreads();  // the only workgroup barrier: from here on a wave touches nothing another wave writes
    if (item >= items) return;
    const LaneRoots R = make_roots(sTw, lane);
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[64 * r + lane] = src[64 * r + lane];
    }
    wave_sync();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    double dev_max = 0.0;
    constexpr int kRowBytes = 2 * kM * (int)sizeof(double2), kStepBytes = 2 * L * kRowBytes;
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(bkf1), (short)0, K.n * kStepBytes, 0x00020000);  // raw dwords, bounds = the whole spectrum
    const int lane16 = lane * (int)sizeof(double2);

    const int32_t my_a = (i0 + lane < i1) ? (int32_t)bara[i0 + lane] : 0;
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readlane(my_a, i - i0);
        if (a == 0) continue;  // wave-uniform; exact arithmetic makes the step a no-op
        // BK_i rows [2L][2][8][64]
        const int bki = i * kStepBytes;  // byte offset of BK_i, rows [2L][2][8][64] double2
        double2 s[2][8];
        uint32_t v0[8], v1[8];
        auto decompose = [&](const int32_t* accp) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t j = 64 * r + lane;
                v0[r] = (((uint32_t)rot_coef(accp, j, a, kN) - (uint32_t)accp[j]) + dec_offset) ^ dec_offset;
                v1[r] = (((uint32_t)rot_coef(accp, j + kM, a, kN) - (uint32_t)accp[j + kM]) + dec_offset) ^ dec_offset;
            }
        };
        auto digit_row = [&](const int sh, const int brow, auto first) {
            constexpr bool FIRST = decltype(first)::value;  // the first row's products initialise s
            double2 x[8], bA[8], bB[8];
            load_bk_block<BUF>(bA, bk_rsrc, lane16, brow, bkf1);                  // -> output polynomial 0
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);  // v_bfe_i32
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                              : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            if (EARLYB == 1) {
                load_bk_block<BUF>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (EARLYB >= 2) {
                // the second block is requested inside the transform, into the registers its twiddles leave
                // (2: after the first inter-pass twiddles, 3: after the second)
                auto req = [&]() {
                    load_bk_block<BUF>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);
                };
                fft512_forward<true, XLANE, 0, decltype(req), EARLYB == 3, 0>(x, sT, lane, R, req);
            } else {
                fft512_forward<true, XLANE, 0, NoHook, false, 0>(x, sT, lane, R);
            }
            if (EARLYB == 0) {
                load_bk_block<BUF>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);  // -> output polynomial 1
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[0][k] = FIRST ? cmulx<false>(x[k], bA[k])
                                : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[0][k].x)),
                                               fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[0][k].y)));
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[1][k] = FIRST ? cmulx<false>(x[k], bB[k])
                                : make_double2(fma(x[k].x, bB[k].x, fma(-x[k].y, bB[k].y, s[1][k].x)),
                                               fma(x[k].x, bB[k].y, fma(x[k].y, bB[k].x, s[1][k].y)));
        };
        decompose(acc);
        digit_row(32 - BGBIT, bki, std::true_type{});
#pragma unroll 1
        for (int row = 1; row < 2 * L; row++) {
            if (row == L) decompose(acc + kN);
            const int q = row >= L ? row - L : row;
            digit_row(32 - (q + 1) * BGBIT, bki + row * kRowBytes, std::false_type{});
        }
        fft512_inverse_pair<true>(s[0], s[1], sT, lane, R);
        // back to coefficients: s[c] holds output polynomial c; round and accumulate
#pragma unroll
        for (int c = 0; c < 2; c++) {
            uint32_t* accc = reinterpret_cast<uint32_t*>(acc) + c * kN;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double2 z = r == 0 ? make_double2(s[c][0].x * (1.0 / 512.0), s[c][0].y * (1.0 / 512.0))
                                         : cmulx<true>(s[c][r], untwist_reg(r));
                const double t0 = z.x + kMagic, t1 = z.y + kMagic;
                if (GUARD) {
                    dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic)));
                    dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic)));
                }
                const int32_t j = 64 * r + lane;
                // ds_add_u32 (no return): one LDS instruction instead of read, add, write
                __hip_atomic_fetch_add(&accc[j], (uint32_t)__double2loint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&accc[j + kM], (uint32_t)__double2loint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        wave_sync();
    }
    if (GUARD) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);  // non-negative floats order like their bit patterns
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        // K4: sample extract after the last slice
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = lane; j <= kN; j += 64)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[64 * r + lane] = src[64 * r + lane];
    }
}

// ---- K3 (+K4), throughput form, round 3: k_blind_rotate_w1 with fewer non-FP64 instructions ----
// Same mapping, same transforms, same BK schedule as k_blind_rotate_w1; what changed is everything around the FP64 work:
//   * the accumulators stand FIRST in the workgroup's LDS, each polynomial on a 4 KiB boundary, so the byte address of
//     coefficient (j - a) mod N is one v_and_or_b32 of a per-step lane value plus 256 r, and the address of coefficient
//     j + 512 is that address ^ 2048 (before: and / shift / add per coefficient);
//   * the negacyclic sign is a v_bfe_i32 of the same per-step value (before: and, compare, select);
//   * (X^a - 1) acc + offset is formed as (rot ^ m) + ((offset - acc_j) - m)  (v_xad_u32);
//   * GUARD = 2 folds the distance to the nearest integer of ONE rounded coefficient in four into the running maximum
//     (registers r = 0 and r = 4 of both output polynomials: 8 of 32 per lane and step) -- the guard is a monitor of the
//     error LEVEL of a launch, and a quarter of ~10^8 coefficients per launch is the same monitor;
//     the evaluator's audit (every K-th launch re-run on the two-limb kernel, evaluator.hip) is the per-bit check;
//   * XMIX: which forward transforms take their lane-high transpose through LDS instead of cross-lane
//     (0 none, 1 the rows of polynomial 1, 2 every second row): balances vector issue against LDS stores.
// dynamic LDS: acc [4][2][1024] int32 | sT [4][kTile] double2 | tw [kTwElems] double2          (78 848 B -> 2 per CU)
// TWG: where the transforms take their twiddles from: 0 the LDS copy, 1 the first inter-pass set (8 per transform) from the
// global table (L1 / L2), 2 both sets -- 105 of a step's 341 LDS instructions moved to the vector-memory path
// PF: L2 prefetch of the NEXT step's BK blocks.  All resident waves of an XCD walk the same BK blocks nearly in step, so the
// first wave to touch a block takes the L2 miss (Infinity Cache, ~2 k cycles) and the others queue behind the same fill:
// phase stamps (br_variant 49) put ~1.8 k cycles of waiting into every digit row.  With PF every wave, once per step, touches
// one 8 KiB slice of BK_{i+1} -- one buffer_load_dword, each lane a different 128-byte line, the slice picked by the wave's
// number so that the waves of an XCD cover all twelve many times over -- a whole step before anybody needs it.
template <int L, int BGBIT, int GUARD, int XMIX = 0, int TWG = 0, bool DIAG = false, bool PF = false>
__global__ __launch_bounds__(64 * kW1Gates, 2) void k_blind_rotate_w1b(DevKeys K, const double2* __restrict__ bkf1,
                                                                       const uint16_t* __restrict__ st_bara, int32_t nb,
                                                                       int32_t* st_acc, int64_t items, int32_t i0, int32_t i1,
                                                                       Torus32* ext, unsigned* guard,
                                                                       const double2* __restrict__ gtw, unsigned long long* diag = nullptr) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc_all = reinterpret_cast<int32_t*>(smem);
    double2* sT_all = reinterpret_cast<double2*>(smem + (size_t)kW1Gates * 2 * kN * 4);
    double2* sTw = sT_all + kW1Gates * kTile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    int32_t* acc = acc_all + wave * 2 * kN;
    const int64_t item = (int64_t)blockIdx.x * kW1Gates + wave;
    load_twiddles(sTw, gtw, tid, 64 * kW1Gates);
    __syncthreads();  // the only workgroup barrier
    if (item >= items) return;
    const LaneRoots Rl = make_roots(sTw, lane);
    const BufRoots<(TWG == 2 ? 2 : 1)> Rb{Rl, __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(gtw), (short)0, kTwElems * (int)sizeof(double2), 0x00020000),
                                       lane * (int)sizeof(double2), (512 + (lane & 7)) * (int)sizeof(double2)};
    auto roots = [&]() -> decltype(auto) {
        if constexpr (TWG == 0) return (Rl);
        else return (Rb);
    };
    const auto& R = roots();
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[64 * r + lane] = src[64 * r + lane];
    }
    wave_sync();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    double dev_max = 0.0;
    constexpr int kRowBytes = 2 * kM * (int)sizeof(double2), kStepBytes = 2 * L * kRowBytes;
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(bkf1), (short)0, K.n * kStepBytes, 0x00020000);
    const int lane16 = lane * (int)sizeof(double2);
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc_all);  // LDS offset 0 of the workgroup
    const uint32_t pb0 = (uint32_t)wave * (2 * kN * 4);                            // this gate's polynomial 0; polynomial 1 at + 4096

    constexpr int kPfSlices = kStepBytes / (64 * 128);  // 128-byte lines of a step's BK blocks, 64 per wave-instruction
    const int pf_off = (int)((((blockIdx.x >> 3) * kW1Gates + wave) % kPfSlices) * 64 + lane) * 128;
    int pf_sink = 0;
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if (DIAG) tlast = stamp();
#define IEACHE_STAMP(idx)                      \
    if (DIAG) {                                \
        const unsigned long long t_ = stamp(); \
        tsum[idx] += t_ - tlast;               \
        tlast = t_;                            \
    }
    const int32_t my_a = (i0 + lane < i1) ? (int32_t)bara[i0 + lane] : 0;
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readlane(my_a, i - i0);
        if (a == 0) continue;  // wave-uniform; exact arithmetic makes the step a no-op
        const int bki = i * kStepBytes;
        double2 s[2][8];
        uint32_t v0[8], v1[8];
        // byte offset of coefficient (lane - a) in the 2N-ring [acc, -acc]: bits 2..11 address, bit 12 = negate
        const uint32_t jb4 = ((uint32_t)(lane - a) & (2 * kN - 1)) << 2;
        auto decompose = [&](const uint32_t pb) {
            const int32_t* accp = reinterpret_cast<const int32_t*>(accb + pb);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t t = jb4 + 256u * r;
                const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
                const int32_t m0 = __builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = __builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
                const uint32_t rv0 = *reinterpret_cast<const uint32_t*>(accb + o0), rv1 = *reinterpret_cast<const uint32_t*>(accb + o1);
                const uint32_t pv0 = (uint32_t)accp[64 * r + lane], pv1 = (uint32_t)accp[64 * r + lane + kM];
                // +/- rot - acc_j + offset, then ^ offset: digit q's field holds digit ^ halfBg, whose sign-extended value IS the digit
                v0[r] = ((rv0 ^ (uint32_t)m0) + ((dec_offset - pv0) - (uint32_t)m0)) ^ dec_offset;
                v1[r] = ((rv1 ^ (uint32_t)m1) + ((dec_offset - pv1) - (uint32_t)m1)) ^ dec_offset;
            }
        };
        auto digit_row = [&](const int sh, const int brow, auto first, auto via_lds) {
            constexpr bool FIRST = decltype(first)::value;
            constexpr int XL = decltype(via_lds)::value == 1 ? 0 : (decltype(via_lds)::value == 2 ? 3 : 1);  // 0: lane-high cross-lane, 1: both through LDS, 2: both cross-lane
            double2 x[8], bA[8], bB[8];
            load_bk_block<true>(bA, bk_rsrc, lane16, brow, bkf1);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);  // v_bfe_i32
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                              : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            __builtin_amdgcn_sched_barrier(0);
            IEACHE_STAMP(1)
            fft512_forward<true, XL, 0, NoHook, false, 1, NoHook>(x, sT, lane, R);
            IEACHE_STAMP(2)
            load_bk_block<true>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[0][k] = FIRST ? cmulx<false>(x[k], bA[k])
                                : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[0][k].x)),
                                               fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[0][k].y)));
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[1][k] = FIRST ? cmulx<false>(x[k], bB[k])
                                : make_double2(fma(x[k].x, bB[k].x, fma(-x[k].y, bB[k].y, s[1][k].x)),
                                               fma(x[k].x, bB[k].y, fma(x[k].y, bB[k].x, s[1][k].y)));
            IEACHE_STAMP(3)
        };
        if (XMIX == 3) {
            // software pipeline over the 2L rows: the digits / conversion / twist of row + 1 (and, before the first row of
            // polynomial 1, its decomposition) run under row's lane-low transpose, into a second set of registers
            auto prep = [&](double2 (&x)[8], const int sh) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);
                    const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                    x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                                  : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
                }
            };
            double2 xa[8], xb[8];
            decompose(pb0);
            prep(xa, 32 - BGBIT);
            auto row_body = [&](auto row_c, double2 (&x)[8], double2 (&xn)[8]) {
                constexpr int row = decltype(row_c)::value;
                constexpr bool FIRST = row == 0;
                const int brow = bki + row * kRowBytes;
                double2 bA[8], bB[8];
                load_bk_block<true>(bA, bk_rsrc, lane16, brow, bkf1);
                __builtin_amdgcn_sched_barrier(0);
                auto next = [&]() {
                    if (row + 1 < 2 * L) {
                        if (row + 1 == L) decompose(pb0 + 4096u);
                        constexpr int q = (row + 1) >= L ? row + 1 - L : row + 1;
                        prep(xn, 32 - (q + 1) * BGBIT);
                    }
                };
                fft512_forward<true, 1, 0, NoHook, false, 1, decltype(next)>(x, sT, lane, R, NoHook(), next);
                load_bk_block<true>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    s[0][k] = FIRST ? cmulx<false>(x[k], bA[k])
                                    : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[0][k].x)),
                                                   fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[0][k].y)));
#pragma unroll
                for (int k = 0; k < 8; k++)
                    s[1][k] = FIRST ? cmulx<false>(x[k], bB[k])
                                    : make_double2(fma(x[k].x, bB[k].x, fma(-x[k].y, bB[k].y, s[1][k].x)),
                                                   fma(x[k].x, bB[k].y, fma(x[k].y, bB[k].x, s[1][k].y)));
            };
            row_body(std::integral_constant<int, 0>{}, xa, xb);
            row_body(std::integral_constant<int, 1>{}, xb, xa);
            if (L > 2) row_body(std::integral_constant<int, (L > 2 ? 2 : 0)>{}, xa, xb);
            if (L > 2) {
                row_body(std::integral_constant<int, (L > 2 ? 3 : 0)>{}, xb, xa);
                row_body(std::integral_constant<int, (L > 2 ? 4 : 0)>{}, xa, xb);
                row_body(std::integral_constant<int, (L > 2 ? 5 : 0)>{}, xb, xa);
            } else {
                row_body(std::integral_constant<int, 2>{}, xa, xb);
                row_body(std::integral_constant<int, (L > 2 ? 0 : 3)>{}, xb, xa);
            }
        } else {
        decompose(pb0);
        IEACHE_STAMP(0)
        digit_row(32 - BGBIT, bki, std::true_type{}, std::integral_constant<int, (XMIX >= 4 ? 2 : 0)>{});
        if (XMIX == 2) {
#pragma unroll 1
            for (int row = 1; row < 2 * L; row += 2) {   // odd rows through LDS, even rows cross-lane
                if (row == L) decompose(pb0 + 4096u);
                const int q = row >= L ? row - L : row;
                digit_row(32 - (q + 1) * BGBIT, bki + row * kRowBytes, std::false_type{}, std::integral_constant<int, 1>{});
                if (row + 1 < 2 * L) {
                    if (row + 1 == L) decompose(pb0 + 4096u);
                    const int q2 = row + 1 >= L ? row + 1 - L : row + 1;
                    digit_row(32 - (q2 + 1) * BGBIT, bki + (row + 1) * kRowBytes, std::false_type{}, std::integral_constant<int, 0>{});
                }
            }
        } else {
#pragma unroll 1
            for (int row = 1; row < L; row++)
                digit_row(32 - (row + 1) * BGBIT, bki + row * kRowBytes, std::false_type{}, std::integral_constant<int, (XMIX >= 4 ? 2 : 0)>{});
            decompose(pb0 + 4096u);
            IEACHE_STAMP(0)
#pragma unroll 1
            for (int row = L; row < 2 * L; row++)
                digit_row(32 - (row - L + 1) * BGBIT, bki + row * kRowBytes, std::false_type{}, std::integral_constant<int, (XMIX == 1 ? 1 : XMIX == 5 ? 2 : 0)>{});
        }
        }
        if (PF) {  // nothing reads the result; its slot in the in-order return queue is long gone when the next step's loads wait
            pf_sink = __builtin_amdgcn_raw_buffer_load_b32(bk_rsrc, pf_off, bki + kStepBytes, 0);
            asm volatile("" ::"v"(pf_sink));
        }
        fft512_inverse_pair<true>(s[0], s[1], sT, lane, R);
        IEACHE_STAMP(4)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            uint32_t* accc = reinterpret_cast<uint32_t*>(acc) + c * kN;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool watched = GUARD == 1 || (GUARD == 2 && (r & 3) == 0);
                double t0, t1;
                if (r == 0) {
                    if (watched) {
                        const double zx = s[c][0].x * (1.0 / 512.0), zy = s[c][0].y * (1.0 / 512.0);
                        t0 = zx + kMagic, t1 = zy + kMagic;
                        dev_max = fmax(dev_max, fabs(zx - (t0 - kMagic)));
                        dev_max = fmax(dev_max, fabs(zy - (t1 - kMagic)));
                    } else {
                        t0 = fma(s[c][0].x, 1.0 / 512.0, kMagic), t1 = fma(s[c][0].y, 1.0 / 512.0, kMagic);
                    }
                } else {
                    const double2 z = cmulx<true>(s[c][r], untwist_reg(r));
                    t0 = z.x + kMagic, t1 = z.y + kMagic;
                    if (watched) {
                        dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic)));
                        dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic)));
                    }
                }
                const int32_t j = 64 * r + lane;
                __hip_atomic_fetch_add(&accc[j], (uint32_t)__double2loint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&accc[j + kM], (uint32_t)__double2loint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        wave_sync();
        IEACHE_STAMP(5)
    }
#undef IEACHE_STAMP
    if (DIAG && diag && lane == 0) {
#pragma unroll
        for (int t = 0; t < 8; t++) atomicAdd(&diag[(wave & 1) * 8 + t], tsum[t]);
    }
    if (GUARD) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = lane; j <= kN; j += 64)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[64 * r + lane] = src[64 * r + lane];
    }
}

// ---- K3 (+K4), throughput form of the PROVABLY EXACT product, round 4: one wave per gate on the TWO-limb spectrum ----
// k_blind_rotate_w2 (above) gives the two-limb product -- BK split into balanced 16-bit limbs, every rounded sum below 2^35
// of the 2^53 an FP64 mantissa holds, so rounding recovers the integer whatever the transform's schedule -- to two waves
// per gate that exchange every forward spectrum through LDS: six workgroup barriers per CMux step.  This is that product on
// k_blind_rotate_w1b's mapping: ONE wave owns a gate, four gates share a workgroup for the twiddle table only, no barrier
// and nothing crossing waves inside a step, BK through buffer loads with the block index in the scalar offset, accumulators
// first in LDS on 4 KiB boundaries (rotation address = one v_and_or of a per-step lane value), ds_add_u32 update.
// Per step and gate: 2L forward transforms (the digits, as in the one-limb kernels), 4 x 2L row products into FOUR
// spectrum sums s[2 c + limb] (128 VGPRs), four inverse transforms (two interleaved pairs through the gate's one tile),
// the two limbs of an output recombined as lo + (hi << 16) mod 2^32.  With the sums taking half the register file, only two
// BK blocks are in flight at a time: block 0 of a row is requested inside its forward transform (once the second twiddle
// set is consumed), block q + 2 when block q has been multiplied.  No guard: nothing here can round wrongly.
// dynamic LDS as k_blind_rotate_w1b: acc [4][2][1024] int32 | sT [4][kTile] double2 | tw [kTwElems] double2   (78 848 B -> 2 per CU)
template <int L, int BGBIT, int BKMODE = 0>
__global__ __launch_bounds__(64 * kW1Gates, 2) void k_blind_rotate_x1(DevKeys K, const double2* __restrict__ bkf,
                                                                      const uint16_t* __restrict__ st_bara, int32_t nb,
                                                                      int32_t* st_acc, int64_t items, int32_t i0, int32_t i1,
                                                                      Torus32* ext, const double2* __restrict__ gtw) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc_all = reinterpret_cast<int32_t*>(smem);
    double2* sT_all = reinterpret_cast<double2*>(smem + (size_t)kW1Gates * 2 * kN * 4);
    double2* sTw = sT_all + kW1Gates * kTile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    int32_t* acc = acc_all + wave * 2 * kN;
    const int64_t item = (int64_t)blockIdx.x * kW1Gates + wave;
    load_twiddles(sTw, gtw, tid, 64 * kW1Gates);
    __syncthreads();  // the only workgroup barrier
    if (item >= items) return;
    const LaneRoots R = make_roots(sTw, lane);
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[64 * r + lane] = src[64 * r + lane];
    }
    wave_sync();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    constexpr int kBlockBytes = kM * (int)sizeof(double2), kRowBytes = 4 * kBlockBytes, kStepBytes = 2 * L * kRowBytes;
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(bkf), (short)0, K.n * kStepBytes, 0x00020000);
    const int lane16 = lane * (int)sizeof(double2);
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc_all);  // LDS offset 0 of the workgroup
    const uint32_t pb0 = (uint32_t)wave * (2 * kN * 4);                            // this gate's polynomial 0; polynomial 1 at + 4096

    const int32_t my_a = (i0 + lane < i1) ? (int32_t)bara[i0 + lane] : 0;
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readlane(my_a, i - i0);
        if (a == 0) continue;  // wave-uniform; exact arithmetic makes the step a no-op
        const int bki = i * kStepBytes;  // BK_i rows [2L][q = 2 c + limb][8][64] double2
        double2 s[4][8];
        uint32_t v0[8], v1[8];
        // byte offset of coefficient (lane - a) in the 2N-ring [acc, -acc]: bits 2..11 address, bit 12 = negate
        const uint32_t jb4 = ((uint32_t)(lane - a) & (2 * kN - 1)) << 2;
        auto decompose = [&](const uint32_t pb, const uint32_t jb) {
            const int32_t* accp = reinterpret_cast<const int32_t*>(accb + pb);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t t = jb + 256u * r;
                const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
                const int32_t m0 = __builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = __builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
                const uint32_t rv0 = *reinterpret_cast<const uint32_t*>(accb + o0), rv1 = *reinterpret_cast<const uint32_t*>(accb + o1);
                const uint32_t pv0 = (uint32_t)accp[64 * r + lane], pv1 = (uint32_t)accp[64 * r + lane + kM];
                v0[r] = ((rv0 ^ (uint32_t)m0) + ((dec_offset - pv0) - (uint32_t)m0)) ^ dec_offset;
                v1[r] = ((rv1 ^ (uint32_t)m1) + ((dec_offset - pv1) - (uint32_t)m1)) ^ dec_offset;
            }
        };
        auto mac = [&](double2 (&acc_s)[8], const double2 (&x)[8], const double2 (&b)[8], auto first) {
            constexpr bool FIRST = decltype(first)::value;
#pragma unroll
            for (int k = 0; k < 8; k++)
                acc_s[k] = FIRST ? cmulx<false>(x[k], b[k])
                                 : make_double2(fma(x[k].x, b[k].x, fma(-x[k].y, b[k].y, acc_s[k].x)),
                                                fma(x[k].x, b[k].y, fma(x[k].y, b[k].x, acc_s[k].y)));
        };
        auto digit_row = [&](const int sh, const int brow, auto first) {
            double2 x[8], bA[8], bB[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);  // v_bfe_i32
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                              : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            __builtin_amdgcn_sched_barrier(0);
            auto req = [&]() { load_bk_block<true>(bA, bk_rsrc, lane16, brow); };  // block 0: output 0, low limb
            if (BKMODE == 1) {
                req();
                __builtin_amdgcn_sched_barrier(0);
                fft512_forward<true, 1, 0, NoHook, false, 1, NoHook, LaneRoots, true>(x, sT, lane, R);
            } else {
                fft512_forward<true, 1, 0, decltype(req), true, 1, NoHook, LaneRoots, true>(x, sT, lane, R, req);
            }
            load_bk_block<true>(bB, bk_rsrc, lane16, brow + kBlockBytes);          // block 1: output 0, high limb
            __builtin_amdgcn_sched_barrier(0);
            if (BKMODE == 2) {  // block granularity: block q + 2 requested when block q has been multiplied (A/B partner)
                mac(s[0], x, bA, first);
                __builtin_amdgcn_sched_barrier(0);
                load_bk_block<true>(bA, bk_rsrc, lane16, brow + 2 * kBlockBytes);
                __builtin_amdgcn_sched_barrier(0);
                mac(s[1], x, bB, first);
                __builtin_amdgcn_sched_barrier(0);
                load_bk_block<true>(bB, bk_rsrc, lane16, brow + 3 * kBlockBytes);
                __builtin_amdgcn_sched_barrier(0);
                mac(s[2], x, bA, first);
                mac(s[3], x, bB, first);
            } else {
                // register granularity: each 16-byte register of block q is re-requested for block q + 2 right behind the
                // products that consumed it, so a request always has two blocks' worth of products (64 FMAs) to arrive under
                constexpr bool FIRST = decltype(first)::value;
#define IEACHE_X1_MAC2(S, B, NEXT, NEXT_OFF)                                                                              \
    _Pragma("unroll") for (int k = 0; k < 8; k += 2) {                                                                  \
        _Pragma("unroll") for (int kk = k; kk < k + 2; kk++)                                                            \
            S[kk] = FIRST ? cmulx<false>(x[kk], B[kk])                                                                  \
                          : make_double2(fma(x[kk].x, B[kk].x, fma(-x[kk].y, B[kk].y, S[kk].x)),                        \
                                         fma(x[kk].x, B[kk].y, fma(x[kk].y, B[kk].x, S[kk].y)));                        \
        if (NEXT) {                                                                                                     \
            _Pragma("unroll") for (int kk = k; kk < k + 2; kk++) {                                                      \
                const v4i_t d = __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane16 + (kk & 3) * 1024, (NEXT_OFF) + (kk >> 2) * 4096, 0); \
                B[kk] = make_double2(__hiloint2double(d.y, d.x), __hiloint2double(d.w, d.z));                           \
            }                                                                                                           \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
                IEACHE_X1_MAC2(s[0], bA, true, brow + 2 * kBlockBytes)
                IEACHE_X1_MAC2(s[1], bB, true, brow + 3 * kBlockBytes)
                IEACHE_X1_MAC2(s[2], bA, false, 0)
                IEACHE_X1_MAC2(s[3], bB, false, 0)
#undef IEACHE_X1_MAC2
            }
        };
        decompose(pb0, jb4);
        digit_row(32 - BGBIT, bki, std::true_type{});
#pragma unroll 1
        for (int row = 1; row < L; row++) digit_row(32 - (row + 1) * BGBIT, bki + row * kRowBytes, std::false_type{});
        // opaque copy: the addresses and sign masks of the second decomposition are recomputed (~50 integer instructions) rather
        // than kept from the first one across three digit rows -- 46 registers this kernel does not have (they were spilled)
        uint32_t jb4b = jb4;
        asm volatile("" : "+v"(jb4b));
        decompose(pb0 + 4096u, jb4b);
#pragma unroll 1
        for (int row = L; row < 2 * L; row++) digit_row(32 - (row - L + 1) * BGBIT, bki + row * kRowBytes, std::false_type{});
        // back to coefficients: s[2 c] / s[2 c + 1] hold the low / high limb sums of output polynomial c
#pragma unroll
        for (int c = 0; c < 2; c++) {
            fft512_inverse_pair<true>(s[2 * c], s[2 * c + 1], sT, lane, R);
            uint32_t* accc = reinterpret_cast<uint32_t*>(acc) + c * kN;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double l0, l1, h0, h1;
                if (r == 0) {
                    l0 = fma(s[2 * c][0].x, 1.0 / 512.0, kMagic), l1 = fma(s[2 * c][0].y, 1.0 / 512.0, kMagic);
                    h0 = fma(s[2 * c + 1][0].x, 1.0 / 512.0, kMagic), h1 = fma(s[2 * c + 1][0].y, 1.0 / 512.0, kMagic);
                } else {
                    const double2 zl = cmulx<true>(s[2 * c][r], untwist_reg(r)), zh = cmulx<true>(s[2 * c + 1][r], untwist_reg(r));
                    l0 = zl.x + kMagic, l1 = zl.y + kMagic, h0 = zh.x + kMagic, h1 = zh.y + kMagic;
                }
                const int32_t j = 64 * r + lane;
                const uint32_t d0 = (uint32_t)__double2loint(l0) + ((uint32_t)__double2loint(h0) << 16);
                const uint32_t d1 = (uint32_t)__double2loint(l1) + ((uint32_t)__double2loint(h1) << 16);
                __hip_atomic_fetch_add(&accc[j], d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&accc[j + kM], d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        wave_sync();
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = lane; j <= kN; j += 64)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[64 * r + lane] = src[64 * r + lane];
    }
}

// ---- K3 (+K4), mid-size launches, round 3: two waves per gate, the ROWS split between them ----
// k_blind_rotate_w2s (below) splits a gate by OUTPUT polynomial: wave w owns output w, so each of its three forward
// spectra has to reach the partner through LDS -- two workgroup barriers per digit row, six per step, and a lone wave per
// SIMD spends them waiting.  Here the split is by ROW of BK_i: wave w decomposes accumulator polynomial w, transforms its
// three digits and multiplies each with BOTH output blocks of its own rows (k_blind_rotate_w1's digit row, three times
// instead of six), keeping two partial spectrum sums.  Only then do the waves meet: each hands the partial sum of the
// OTHER output to its partner through its own (idle) tile, adds what it receives, inverse-transforms output w and updates
// accumulator polynomial w -- two barriers per step.  The next step's decomposition reads polynomial w only, so nothing
// else crosses waves.  Arithmetic differs from k_blind_rotate_w1 only in the order of two exact-after-rounding FP64 sums.
// dynamic LDS: acc [2][1024] int32 | sT [2][kTile] double2 | tw [kTwElems] double2       (35 840 B -> 4 per CU)
template <int L, int BGBIT, int GUARD>
__global__ __launch_bounds__(128, 2) void k_blind_rotate_w2r(DevKeys K, const double2* __restrict__ bkf1,
                                                            const uint16_t* __restrict__ st_bara, int32_t nb,
                                                            int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                            unsigned* guard, const double2* __restrict__ gtw) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);
    double2* sT_all = reinterpret_cast<double2*>(smem + (size_t)2 * kN * 4);
    double2* sTw = sT_all + 2 * kTile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    const double2* sTp = sT_all + (wave ^ 1) * kTile;
    const int64_t item = (int64_t)blockIdx.x;
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, 128);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[128 * r + tid] = src[128 * r + tid];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    double dev_max = 0.0;
    constexpr int kRowBytes = 2 * kM * (int)sizeof(double2), kStepBytes = 2 * L * kRowBytes;
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(bkf1), (short)0, K.n * kStepBytes, 0x00020000);
    const int lane16 = lane * (int)sizeof(double2);
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc);  // LDS offset 0 of the workgroup
    const uint32_t pb = (uint32_t)wave * (kN * 4);                             // this wave's polynomial
    const int32_t* accp = acc + wave * kN;
    uint32_t* accu = reinterpret_cast<uint32_t*>(acc) + wave * kN;

    int32_t my_a = 0;  // the rotation amounts of 64 steps at a time, one per lane (a launch may be the whole rotation)
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        if (((i - i0) & 63) == 0) my_a = (i + lane < i1) ? (int32_t)bara[i + lane] : 0;
        const int32_t a = __builtin_amdgcn_readlane(my_a, (i - i0) & 63);
        if (a == 0) continue;  // workgroup-uniform
        const int bki = i * kStepBytes + wave * L * kRowBytes;  // this wave's rows of BK_i
        double2 s[2][8];
        uint32_t v0[8], v1[8];
        const uint32_t jb4 = ((uint32_t)(lane - a) & (2 * kN - 1)) << 2;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t t = jb4 + 256u * r;
            const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
            const int32_t m0 = __builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = __builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
            const uint32_t rv0 = *reinterpret_cast<const uint32_t*>(accb + o0), rv1 = *reinterpret_cast<const uint32_t*>(accb + o1);
            const uint32_t pv0 = (uint32_t)accp[64 * r + lane], pv1 = (uint32_t)accp[64 * r + lane + kM];
            v0[r] = ((rv0 ^ (uint32_t)m0) + ((dec_offset - pv0) - (uint32_t)m0)) ^ dec_offset;
            v1[r] = ((rv1 ^ (uint32_t)m1) + ((dec_offset - pv1) - (uint32_t)m1)) ^ dec_offset;
        }
        auto digit_row = [&](const int sh, const int brow, auto first) {
            constexpr bool FIRST = decltype(first)::value;
            double2 x[8], bA[8], bB[8];
            load_bk_block<true>(bA, bk_rsrc, lane16, brow, bkf1);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                              : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            __builtin_amdgcn_sched_barrier(0);
            fft512_forward<true, 1, 0>(x, sT, lane, R);
            load_bk_block<true>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[0][k] = FIRST ? cmulx<false>(x[k], bA[k])
                                : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[0][k].x)),
                                               fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[0][k].y)));
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[1][k] = FIRST ? cmulx<false>(x[k], bB[k])
                                : make_double2(fma(x[k].x, bB[k].x, fma(-x[k].y, bB[k].y, s[1][k].x)),
                                               fma(x[k].x, bB[k].y, fma(x[k].y, bB[k].x, s[1][k].y)));
        };
        digit_row(32 - BGBIT, bki, std::true_type{});
#pragma unroll 1
        for (int q = 1; q < L; q++) digit_row(32 - (q + 1) * BGBIT, bki + q * kRowBytes, std::false_type{});
        // the partial sum of the partner's output goes to the partner through this wave's tile (idle since the last transform)
        double2 y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double2 mine = wave ? s[1][k] : s[0][k], theirs = wave ? s[0][k] : s[1][k];
            sT[k * 64 + lane] = theirs;
            y[k] = mine;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double2 z = sTp[k * 64 + lane];
            y[k] = cadd(y[k], z);
        }
        __syncthreads();  // the partner has read this wave's tile before the inverse transform reuses it
        fft512_inverse<true, 0>(y, sT, lane, R);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool watched = GUARD == 1 || (GUARD == 2 && (r & 3) == 0);
            double t0, t1;
            if (r == 0) {
                if (watched) {
                    const double zx = y[0].x * (1.0 / 512.0), zy = y[0].y * (1.0 / 512.0);
                    t0 = zx + kMagic, t1 = zy + kMagic;
                    dev_max = fmax(dev_max, fabs(zx - (t0 - kMagic)));
                    dev_max = fmax(dev_max, fabs(zy - (t1 - kMagic)));
                } else {
                    t0 = fma(y[0].x, 1.0 / 512.0, kMagic), t1 = fma(y[0].y, 1.0 / 512.0, kMagic);
                }
            } else {
                const double2 z = cmulx<true>(y[r], untwist_reg(r));
                t0 = z.x + kMagic, t1 = z.y + kMagic;
                if (watched) {
                    dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic)));
                    dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic)));
                }
            }
            const int32_t j = 64 * r + lane;
            __hip_atomic_fetch_add(&accu[j], (uint32_t)__double2loint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&accu[j + kM], (uint32_t)__double2loint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        wave_sync();  // wave w reads and updates only polynomial w: nothing crosses waves here
    }
    __syncthreads();  // the epilogue below reads both polynomials with all threads
    if (GUARD) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += 128)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[128 * r + tid] = src[128 * r + tid];
    }
}

// ---- K3 (+K4), one to two gates per CU, round 3: FOUR waves per gate, the rows split 2 : 1 : 2 : 1 ----
// Launches of 257 .. 512 gate instances leave every SIMD a single wave under k_blind_rotate_w2r, and a lone wave issues a
// vector instruction only every 6-7 cycles.  Here a gate is one 256-thread workgroup (two per CU: two waves per SIMD):
//   wave 2p     ("heavy", polynomial p): digits 0 .. L-2 of accumulator polynomial p -- L-1 forward transforms, each
//               multiplied with BOTH output blocks of its BK row (two partial spectrum sums, as in k_blind_rotate_w1b);
//   wave 2p + 1 ("light"): digit L-1 the same way, then the inverse transform of OUTPUT polynomial p.
// One hand-over per step: every wave passes on the partial sums it does not invert -- through its own (idle) tile and,
// for the heavy waves' second sum, one of two extra 8 KiB slots -- barrier, the light waves add the three sums they
// receive, inverse-transform (scratch: the tile they have just emptied, which nobody else reads), round and ds_add_u32
// into "their" accumulator polynomial, barrier.  Arithmetic differs from the other kernels only in the order of exact-
// after-rounding FP64 sums.
// dynamic LDS: acc [2][1024] int32 | sT [4][kTile] double2 | X [2][8][64] double2 | tw [kTwElems] double2   (70 656 B -> 2 per CU)
template <int L, int BGBIT, int GUARD>
__global__ __launch_bounds__(256, 2) void k_blind_rotate_w4r(DevKeys K, const double2* __restrict__ bkf1,
                                                            const uint16_t* __restrict__ st_bara, int32_t nb,
                                                            int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                            unsigned* guard, const double2* __restrict__ gtw, int32_t flip_period) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);
    double2* sT_all = reinterpret_cast<double2*>(smem + (size_t)2 * kN * 4);
    double2* sX = sT_all + 4 * kTile;
    double2* sTw = sX + 2 * 8 * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    // The two workgroups that share a CU (observed: workgroup i and i + #CUs) swap the heavy and the light role within each
    // wave pair, so that every SIMD hosts one heavy and one light wave instead of two of a kind waiting for each other's phase.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) ^ (int)((blockIdx.x / (unsigned)flip_period) & 1u);
    const int pol = wave >> 1;
    const bool light = wave & 1;
    double2* sT = sT_all + wave * kTile;
    const int64_t item = (int64_t)blockIdx.x;
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, 256);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 2; r++) dst[256 * r + tid] = src[256 * r + tid];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    double dev_max = 0.0;
    constexpr int kRowBytes = 2 * kM * (int)sizeof(double2), kStepBytes = 2 * L * kRowBytes;
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(bkf1), (short)0, K.n * kStepBytes, 0x00020000);
    const int lane16 = lane * (int)sizeof(double2);
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc);
    const uint32_t pb = (uint32_t)pol * (kN * 4);
    const int32_t* accp = acc + pol * kN;
    uint32_t* accu = reinterpret_cast<uint32_t*>(acc) + pol * kN;
    const int q0 = light ? L - 1 : 0, q1 = light ? L : (L > 1 ? L - 1 : 1);  // this wave's digits [q0, q1)
    // where the partial sums go: `keep` stays (light waves: the output they invert), `give0` / `give1` are handed over
    //   wave 0: s[0] -> tile 0 (for wave 1), s[1] -> slot 0 (for wave 3)      wave 1: s[1] -> tile 1 (for wave 3)
    //   wave 2: s[1] -> tile 2 (for wave 3), s[0] -> slot 1 (for wave 1)      wave 3: s[0] -> tile 3 (for wave 1)
    double2* slot_extra = sX + pol * (8 * 64);
    const double2* in_a = light ? (pol == 0 ? sT_all + 0 * kTile : sX + 0 * (8 * 64)) : nullptr;        // wave 1: tile 0 ; wave 3: slot 0
    const double2* in_b = light ? (pol == 0 ? sX + 1 * (8 * 64) : sT_all + 1 * kTile) : nullptr;        // wave 1: slot 1 ; wave 3: tile 1
    const double2* in_c = light ? (pol == 0 ? sT_all + 3 * kTile : sT_all + 2 * kTile) : nullptr;       // wave 1: tile 3 ; wave 3: tile 2
    double2* scratch = sT_all + (pol == 0 ? 0 : 2) * kTile;  // the heavy partner's tile: after the hand-over only this light wave reads it

    int32_t my_a = 0;  // the rotation amounts of 64 steps at a time, one per lane (a launch may be the whole rotation)
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        if (((i - i0) & 63) == 0) my_a = (i + lane < i1) ? (int32_t)bara[i + lane] : 0;
        const int32_t a = __builtin_amdgcn_readlane(my_a, (i - i0) & 63);
        if (a == 0) continue;  // workgroup-uniform
        const int bki = i * kStepBytes + pol * L * kRowBytes;  // the rows of this wave's polynomial
        double2 s[2][8];
        uint32_t v0[8], v1[8];
        const uint32_t jb4 = ((uint32_t)(lane - a) & (2 * kN - 1)) << 2;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t t = jb4 + 256u * r;
            const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
            const int32_t m0 = __builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = __builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
            const uint32_t rv0 = *reinterpret_cast<const uint32_t*>(accb + o0), rv1 = *reinterpret_cast<const uint32_t*>(accb + o1);
            const uint32_t pv0 = (uint32_t)accp[64 * r + lane], pv1 = (uint32_t)accp[64 * r + lane + kM];
            v0[r] = ((rv0 ^ (uint32_t)m0) + ((dec_offset - pv0) - (uint32_t)m0)) ^ dec_offset;
            v1[r] = ((rv1 ^ (uint32_t)m1) + ((dec_offset - pv1) - (uint32_t)m1)) ^ dec_offset;
        }
        auto digit_row = [&](const int sh, const int brow, auto first) {
            constexpr bool FIRST = decltype(first)::value;
            double2 x[8], bA[8], bB[8];
            load_bk_block<true>(bA, bk_rsrc, lane16, brow, bkf1);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                              : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            __builtin_amdgcn_sched_barrier(0);
            fft512_forward<true, 1, 0>(x, sT, lane, R);
            load_bk_block<true>(bB, bk_rsrc, lane16, brow + kRowBytes / 2, bkf1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[0][k] = FIRST ? cmulx<false>(x[k], bA[k])
                                : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[0][k].x)),
                                               fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[0][k].y)));
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[1][k] = FIRST ? cmulx<false>(x[k], bB[k])
                                : make_double2(fma(x[k].x, bB[k].x, fma(-x[k].y, bB[k].y, s[1][k].x)),
                                               fma(x[k].x, bB[k].y, fma(x[k].y, bB[k].x, s[1][k].y)));
        };
        digit_row(32 - (q0 + 1) * BGBIT, bki + q0 * kRowBytes, std::true_type{});
#pragma unroll 1
        for (int q = q0 + 1; q < q1; q++) digit_row(32 - (q + 1) * BGBIT, bki + q * kRowBytes, std::false_type{});
        // hand-over: the sum for this wave's own polynomial's output goes to (or stays with) the light wave of that polynomial
        double2 y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double2 own = pol ? s[1][k] : s[0][k], other = pol ? s[0][k] : s[1][k];
            if (light) {
                y[k] = own;
                sT[k * 64 + lane] = other;          // for the other polynomial's light wave
            } else {
                sT[k * 64 + lane] = own;            // for this polynomial's light wave
                slot_extra[k * 64 + lane] = other;  // for the other polynomial's light wave
            }
        }
        __syncthreads();
        if (light) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double2 za = in_a[k * 64 + lane], zb = in_b[k * 64 + lane], zc = in_c[k * 64 + lane];
                y[k] = cadd(cadd(y[k], za), cadd(zb, zc));
            }
            fft512_inverse<true, 0>(y, scratch, lane, R);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool watched = GUARD == 1 || (GUARD == 2 && (r & 3) == 0);
                double t0, t1;
                if (r == 0) {
                    if (watched) {
                        const double zx = y[0].x * (1.0 / 512.0), zy = y[0].y * (1.0 / 512.0);
                        t0 = zx + kMagic, t1 = zy + kMagic;
                        dev_max = fmax(dev_max, fabs(zx - (t0 - kMagic)));
                        dev_max = fmax(dev_max, fabs(zy - (t1 - kMagic)));
                    } else {
                        t0 = fma(y[0].x, 1.0 / 512.0, kMagic), t1 = fma(y[0].y, 1.0 / 512.0, kMagic);
                    }
                } else {
                    const double2 z = cmulx<true>(y[r], untwist_reg(r));
                    t0 = z.x + kMagic, t1 = z.y + kMagic;
                    if (watched) {
                        dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic)));
                        dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic)));
                    }
                }
                const int32_t j = 64 * r + lane;
                __hip_atomic_fetch_add(&accu[j], (uint32_t)__double2loint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&accu[j + kM], (uint32_t)__double2loint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        __syncthreads();  // accumulator complete; every handed-over sum consumed; the tiles are scratch again
    }
    if (GUARD && light) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += 256)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 2; r++) dst[256 * r + tid] = src[256 * r + tid];
    }
}

// ---- K3 (+K4), mid-size launches: two waves per gate instance on the ONE-limb spectrum ----
// Between the latency kernel (a handful of gates) and k_blind_rotate_w1 (more gates than the chip holds one-per-SIMD-slot)
// lie launches of a few hundred to ~1 000 gates: deep circuits at small batches, cloudd's batches.  One wave per gate
// leaves SIMD slots empty there, and the step of a lone wave is a serial chain of 8 transforms.  This is
// k_blind_rotate_w2's mapping (wave w decomposes polynomial w and owns output polynomial w; spectra cross through the
// producing wave's tile; two barriers per digit row) on k_blind_rotate_w1's arithmetic: one accumulator per wave,
// 3 forward + 1 inverse transform and 6 row products per wave and step, guarded rounding.
// dynamic LDS as k_blind_rotate_w2: sT [2][kTile] double2 | tw [kTwElems] double2 | acc [2][1024] int32
template <int L, int BGBIT, bool GUARD>
__global__ __launch_bounds__(128, 2) void k_blind_rotate_w2s(DevKeys K, const double2* __restrict__ bkf1,
                                                            const uint16_t* __restrict__ st_bara, int32_t nb,
                                                            int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                            unsigned* guard, const double2* __restrict__ gtw) {
    extern __shared__ __align__(16) unsigned char smem[];
    double2* sT_all = reinterpret_cast<double2*>(smem);
    double2* sTw = sT_all + 2 * kTile;
    int32_t* acc = reinterpret_cast<int32_t*>(sTw + kTwElems);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    const double2* sTp = sT_all + (wave ^ 1) * kTile;
    const int64_t item = (int64_t)blockIdx.x;
    const uint16_t* __restrict__ bara = st_bara + (size_t)item * nb;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, 128);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[128 * r + tid] = src[128 * r + tid];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
    int32_t* accw = acc + wave * kN;  // the polynomial this wave decomposes and updates
    double dev_max = 0.0;


    const int32_t my_a = (i0 + lane < i1) ? (int32_t)bara[i0 + lane] : 0;
#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readlane(my_a, i - i0);
        if (a == 0) continue;  // workgroup-uniform
        // BK_i rows [2L][2][8][64]; this wave reads output block `wave` of every row
        const double2* __restrict__ bki = bkf1 + (size_t)i * (2 * L * 2 * kM) + (size_t)wave * kM + lane;
        double2 s[8];
        uint32_t v0[8], v1[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int32_t j = 64 * r + lane;
            v0[r] = (((uint32_t)rot_coef(accw, j, a, kN) - (uint32_t)accw[j]) + dec_offset) ^ dec_offset;
            v1[r] = (((uint32_t)rot_coef(accw, j + kM, a, kN) - (uint32_t)accw[j + kM]) + dec_offset) ^ dec_offset;
        }
        auto digit_row = [&](const int q, auto first) {
            constexpr bool FIRST = decltype(first)::value;
            const int sh = 32 - (q + 1) * BGBIT;
            const double2* __restrict__ bown = bki + (size_t)(wave * L + q) * (2 * kM);
            const double2* __restrict__ bpar = bki + (size_t)((wave ^ 1) * L + q) * (2 * kM);
            double2 x[8], bA[8], bC[8];
#pragma unroll
            for (int k = 0; k < 8; k++) bA[k] = bown[k * 64];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                              : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
            }
            __builtin_amdgcn_sched_barrier(0);
            auto req = [&]() {  // the partner row's block, requested once the transform's twiddle registers are free
#pragma unroll
                for (int k = 0; k < 8; k++) bC[k] = bpar[k * 64];
            };
            fft512_forward<true, 1, 0, decltype(req), true>(x, sT, lane, R, req);
            // hand the spectrum to the partner wave through our own (now idle) tile
#pragma unroll
            for (int k = 0; k < 8; k++) sT[k * 64 + lane] = x[k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[k] = FIRST ? cmulx<false>(x[k], bA[k])
                             : make_double2(fma(x[k].x, bA[k].x, fma(-x[k].y, bA[k].y, s[k].x)),
                                            fma(x[k].x, bA[k].y, fma(x[k].y, bA[k].x, s[k].y)));
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = sTp[k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 8; k++)
                s[k] = make_double2(fma(x[k].x, bC[k].x, fma(-x[k].y, bC[k].y, s[k].x)),
                                    fma(x[k].x, bC[k].y, fma(x[k].y, bC[k].x, s[k].y)));
            __syncthreads();  // partner has read our tile before the next transform reuses it
        };
        digit_row(0, std::true_type{});
#pragma unroll 1
        for (int q = 1; q < L; q++) digit_row(q, std::false_type{});
        fft512_inverse<true, 0>(s, sT, lane, R);
        uint32_t* accu = reinterpret_cast<uint32_t*>(accw);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double2 z = r == 0 ? make_double2(s[0].x * (1.0 / 512.0), s[0].y * (1.0 / 512.0)) : cmulx<true>(s[r], untwist_reg(r));
            const double t0 = z.x + kMagic, t1 = z.y + kMagic;
            if (GUARD) {
                dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic)));
                dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic)));
            }
            const int32_t j = 64 * r + lane;
            __hip_atomic_fetch_add(&accu[j], (uint32_t)__double2loint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&accu[j + kM], (uint32_t)__double2loint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        wave_sync();  // wave w reads and updates only polynomial w: nothing crosses waves here
    }
    __syncthreads();  // the epilogue below reads both polynomials with all threads
    if (GUARD) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += 128)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
#pragma unroll
        for (int r = 0; r < 4; r++) dst[128 * r + tid] = src[128 * r + tid];
    }
}

// ---- K3 (+K4), latency-oriented: 2L waves per gate instance ----
// For narrow levels (a single expression, the reference's own mode) the time of a level is the
// LATENCY of one blind rotation, and two waves walking 3 forward + 2 inverse transforms one after
// the other leave most of the CU idle.  Here one workgroup of 2L waves takes one gate: wave w
// (w = p*L + q) decomposes digit q of polynomial p and transforms it -- all 2L forward transforms
// at once -- and publishes the spectrum in its own tile; waves 0..3 then each own ONE spectrum
// accumulator (output polynomial w>>1, limb w&1), MAC all 2L rows into it, inverse-transform it and
// add their share to the accumulator polynomial in LDS with ds_add_u32 (addition mod 2^32
// commutes, so the two limb waves need no ordering).  Three barriers per step.
// Same arithmetic as k_blind_rotate_w2 up to the order of exact-after-rounding FP64 sums, so the
// integers it produces are identical.
// dynamic LDS: sT [2L][kTile] double2 | tw [kTwElems] double2 | acc [2][1024] int32 | bara [i1-i0] u16
// LIMBS = 1: the same kernel on the one-limb spectrum [n][2L][2][8][64] -- waves 0 and 1 own the two output polynomials,
// half the BK bytes and LDS reads per step, guarded rounding (`guard`, see k_blind_rotate_w1).
// XF / XI: which transposes of the forward / inverse transform go cross-lane instead of through LDS (bit 0 lane-high, bit 1 lane-low)
template <int L, int BGBIT, bool DIAG, int LIMBS = 2, int XF = 0, int XI = 0>
__global__ __launch_bounds__(128 * L) void k_blind_rotate_wide(DevKeys K, const double2* __restrict__ bkf,
                                                             const uint16_t* __restrict__ st_bara, int32_t nb,
                                                             int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                             unsigned long long* diag, const double2* __restrict__ gtw,
                                                             unsigned* guard) {
    constexpr int RS = 2 * LIMBS * kM;  // double2 elements per BK row
    constexpr int NW = 2 * L, NT = 64 * NW;
    extern __shared__ __align__(16) unsigned char smem[];
    double2* sT_all = reinterpret_cast<double2*>(smem);
    double2* sTw = sT_all + NW * kTile;
    int32_t* acc = reinterpret_cast<int32_t*>(sTw + kTwElems);
    uint16_t* s_bara = reinterpret_cast<uint16_t*>(acc + 2 * kN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    const int64_t item = (int64_t)blockIdx.x;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, NT);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
        const uint16_t* bara = st_bara + (size_t)item * nb;
        for (int idx = tid; idx < i1 - i0; idx += NT) s_bara[idx] = bara[i0 + idx];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;
    const int pw = wave / L, qw = wave - pw * L;  // forward role: digit qw of polynomial pw
    const int sh = 32 - (qw + 1) * BGBIT;
    const int32_t* accp = acc + pw * kN;
    const bool is_out = wave < 2 * LIMBS;         // inverse role: output polynomial wave>>1, limb wave&1 (one limb: polynomial wave)
    uint32_t* acco = reinterpret_cast<uint32_t*>(acc) + (LIMBS == 2 ? (wave >> 1) : wave) * kN;
    const int lsh = LIMBS == 2 ? (wave & 1) * 16 : 0;
    double dev_max = 0.0;
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if (DIAG) tlast = stamp();
#define IEACHE_STAMP(idx)                      \
    if (DIAG) {                                \
        const unsigned long long t_ = stamp(); \
        tsum[idx] += t_ - tlast;               \
        tlast = t_;                            \
    }

#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readfirstlane((int32_t)s_bara[i - i0]);
        if (a == 0) continue;  // workgroup-uniform
        // BK_i rows [2L][4][8][64]: this wave (as an output owner) reads block `wave` of every row.
        // One CU takes 64 B/clk from its vector-memory path, i.e. >= 3 000 cycles for the 192 KiB of a
        // step, so the loads are issued in three instalments spread over the step: rows 0,1 now (they
        // fly under the decomposition and the transform), rows 2,3 after the transform, rows 4,5 once
        // rows 0,1 are consumed.
        const double2* __restrict__ bki = bkf + (size_t)i * (2 * L * RS) + (size_t)wave * kM + lane;
        double2 bA[2][8], bB[2][8], s[8];
        if (is_out) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                bA[0][k] = bki[(size_t)0 * RS + k * 64];
                bA[1][k] = bki[(size_t)1 * RS + k * 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        int32_t lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque: keeps 16 per-coefficient LDS addresses from being hoisted (and spilled)
        // all 32 LDS reads first, then the arithmetic: left to itself the compiler waits for every read
        // before issuing the next one (16 exposed LDS round trips = 2 000 cycles per step)
        uint32_t rv0[8], rv1[8], pv0[8], pv1[8];
        const int32_t jb = (lane_o - a) & (2 * kN - 1);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int32_t j = 64 * r + lane_o;
            rv0[r] = (uint32_t)accp[(jb + 64 * r) & (kN - 1)];        // X^a * acc at j      (sign applied below)
            rv1[r] = (uint32_t)accp[(jb + 64 * r + kM) & (kN - 1)];   //            at j + 512
            pv0[r] = (uint32_t)accp[j];
            pv1[r] = (uint32_t)accp[j + kM];
        }
        __builtin_amdgcn_sched_barrier(0);
        double2 x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t n0 = 0u - (((uint32_t)(jb + 64 * r) >> 10) & 1u);        // all ones where the rotation wrapped
            const uint32_t n1 = 0u - (((uint32_t)(jb + 64 * r + kM) >> 10) & 1u);
            const uint32_t u0 = ((rv0[r] ^ n0) - n0) - pv0[r] + dec_offset;
            const uint32_t u1 = ((rv1[r] ^ n1) - n1) - pv1[r] + dec_offset;
            // digit - halfBg = sign-extended field of (u ^ (halfBg << sh))
            const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)(u0 ^ (halfBg << sh)), sh, BGBIT);
            const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)(u1 ^ (halfBg << sh)), sh, BGBIT);
            x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                          : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
        }
        IEACHE_STAMP(0)
        fft512_forward<true, XF>(x, sT, lane, R);
        IEACHE_STAMP(1)
#pragma unroll
        for (int k = 0; k < 8; k++) sT[k * 64 + lane] = x[k];  // publish
        __builtin_amdgcn_sched_barrier(0);
        if (is_out) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                bB[0][k] = bki[(size_t)2 * RS + k * 64];
                bB[1][k] = bki[(size_t)3 * RS + k * 64];
            }
        }
        IEACHE_STAMP(2)
        __syncthreads();  // A: all 2L spectra are in their tiles
        IEACHE_STAMP(3)
        if (is_out) {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = make_double2(0.0, 0.0);
#define IEACHE_MAC_ROW(row, B)                                                                          \
    {                                                                                                   \
        const double2* sp = sT_all + (row) * kTile + lane;                                              \
        _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                 \
            const double2 y = sp[k * 64];                                                               \
            s[k] = make_double2(fma(y.x, B[k].x, fma(-y.y, B[k].y, s[k].x)), fma(y.x, B[k].y, fma(y.y, B[k].x, s[k].y))); \
        }                                                                                               \
    }
            IEACHE_MAC_ROW(0, bA[0])
            IEACHE_MAC_ROW(1, bA[1])
            if (NW > 4) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    bA[0][k] = bki[(size_t)4 * RS + k * 64];
                    bA[1][k] = bki[(size_t)5 * RS + k * 64];
                }
            }
            IEACHE_MAC_ROW(2, bB[0])
            IEACHE_MAC_ROW(3, bB[1])
            if (NW > 4) {
                IEACHE_MAC_ROW(4, bA[0])
                IEACHE_MAC_ROW(5, bA[1])
            }
#undef IEACHE_MAC_ROW
        }
        IEACHE_STAMP(4)
        __syncthreads();  // B: every spectrum has been consumed, tiles are scratch again
        IEACHE_STAMP(5)
        if (is_out) {
            fft512_inverse<true, XI>(s, sT, lane, R);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double2 z = r == 0 ? make_double2(s[0].x * (1.0 / 512.0), s[0].y * (1.0 / 512.0)) : cmulx<true>(s[r], untwist_reg(r));
                const double t0 = z.x + kMagic, t1 = z.y + kMagic;
                if (LIMBS == 1) dev_max = fmax(dev_max, fmax(fabs(z.x - (t0 - kMagic)), fabs(z.y - (t1 - kMagic))));
                const uint32_t c0 = (uint32_t)__double2loint(t0) << lsh;
                const uint32_t c1 = (uint32_t)__double2loint(t1) << lsh;
                const int32_t j = 64 * r + lane;
                atomicAdd(&acco[j], c0);       // ds_add_u32; the partner limb adds its share to the same word
                atomicAdd(&acco[j + kM], c1);
            }
        }
        IEACHE_STAMP(6)
        __syncthreads();  // C: accumulator complete before the next decomposition
        IEACHE_STAMP(7)
    }
#undef IEACHE_STAMP
    if (LIMBS == 1 && guard && is_out) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > 0.0625f) atomicAdd(&guard[0], 1u);
        }
    }
    if (DIAG && diag && lane == 0 && (wave == 0 || wave == 4)) {
#pragma unroll
        for (int t = 0; t < 8; t++) atomicAdd(&diag[(wave >> 2) * 8 + t], tsum[t]);
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += NT)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
    }
}


// ---- K3 (+K4), latency-oriented, round 3: 2L waves per gate, FOUR output waves on half the rows each ----
// k_blind_rotate_wide on the one-limb spectrum leaves the six row products and the inverse transform of an output
// polynomial to ONE wave (two output waves; the other four idle for half of the step), and its tiles serve both as the
// published spectra and as the inverse transforms' scratch (barrier B).  Here waves 0..3 are output waves (one per SIMD):
// output wave (c, h) = (w & 1, w >> 1) multiplies the L published spectra of accumulator polynomial h with block c of
// their BK rows -- its L blocks are requested at the top of the step and arrive under the decomposition and the forward
// transform --, inverse-transforms that PARTIAL sum in a scratch tile of its own (no barrier B), rounds it and adds it
// into accumulator polynomial c with ds_add_u32.  Each partial sum is an integer polynomial and addition mod 2^32
// commutes, so the two halves of an output need no ordering.  Two barriers per step (spectra published / accumulator
// updated).  Same rounded integers as every other kernel here.
// dynamic LDS: acc [2][1024] int32 | sT [2L + 4][kTile] double2 | tw [kTwElems] double2 | bara [i1-i0] u16
template <int L, int BGBIT, int GUARD>
__global__ __launch_bounds__(128 * L) void k_blind_rotate_wide4(DevKeys K, const double2* __restrict__ bkf1,
                                                              const uint16_t* __restrict__ st_bara, int32_t nb,
                                                              int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                              unsigned* guard, const double2* __restrict__ gtw) {
    constexpr int NW = 2 * L, NT = 64 * NW;
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);
    double2* sT_all = reinterpret_cast<double2*>(smem + (size_t)2 * kN * 4);
    double2* sTw = sT_all + (NW + 4) * kTile;
    uint16_t* s_bara = reinterpret_cast<uint16_t*>(sTw + kTwElems);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;                  // forward scratch, then this wave's published spectrum
    double2* sTi = sT_all + (NW + (wave & 3)) * kTile;    // inverse scratch of output wave `wave`
    const int64_t item = (int64_t)blockIdx.x;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, NT);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
        const uint16_t* bara = st_bara + (size_t)item * nb;
        for (int idx = tid; idx < i1 - i0; idx += NT) s_bara[idx] = bara[i0 + idx];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;
    const int pw = wave / L, qw = wave - pw * L;  // forward role: digit qw of polynomial pw = row `wave` of BK_i
    const int sh = 32 - (qw + 1) * BGBIT;
    const bool is_out = wave < 4;
    const int oc = wave & 1, oh = (wave >> 1) & 1;  // output role: block oc of the rows of polynomial oh
    uint32_t* acco = reinterpret_cast<uint32_t*>(acc) + oc * kN;
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc);
    const uint32_t pb = (uint32_t)pw * (kN * 4);
    const int32_t* accp = acc + pw * kN;
    double dev_max = 0.0;

#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readfirstlane((int32_t)s_bara[i - i0]);
        if (a == 0) continue;  // workgroup-uniform
        // BK_i rows [2L][2][8][64]: this output wave's L blocks, all requested now
        const double2* __restrict__ bki = bkf1 + (size_t)i * (2 * L * 2 * kM) + (size_t)(oh * L) * (2 * kM) + (size_t)oc * kM + lane;
        double2 bk[L][8];
        if (is_out) {
#pragma unroll
            for (int q = 0; q < L; q++)
#pragma unroll
                for (int k = 0; k < 8; k++) bk[q][k] = bki[(size_t)q * (2 * kM) + k * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        int32_t lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque: keeps the per-coefficient LDS addresses from being hoisted out of the step loop
        const uint32_t jb4 = ((uint32_t)(lane_o - a) & (2 * kN - 1)) << 2;
        uint32_t rv0[8], rv1[8], pv0[8], pv1[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {   // all 32 LDS reads first, then the arithmetic
            const uint32_t t = jb4 + 256u * r;
            const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
            rv0[r] = *reinterpret_cast<const uint32_t*>(accb + o0);
            rv1[r] = *reinterpret_cast<const uint32_t*>(accb + o1);
            pv0[r] = (uint32_t)accp[64 * r + lane_o];
            pv1[r] = (uint32_t)accp[64 * r + lane_o + kM];
        }
        __builtin_amdgcn_sched_barrier(0);
        double2 x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t t = jb4 + 256u * r;
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
            const uint32_t u0 = (rv0[r] ^ m0) + ((dec_offset - pv0[r]) - m0);
            const uint32_t u1 = (rv1[r] ^ m1) + ((dec_offset - pv1[r]) - m1);
            // digit - halfBg = sign-extended field of (u ^ (halfBg << sh))
            const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)(u0 ^ (halfBg << sh)), sh, BGBIT);
            const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)(u1 ^ (halfBg << sh)), sh, BGBIT);
            x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                          : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
        }
        fft512_forward<true, 0>(x, sT, lane, R);
#pragma unroll
        for (int k = 0; k < 8; k++) sT[k * 64 + lane] = x[k];  // publish
        __syncthreads();  // A: all 2L spectra are in their tiles
        if (is_out) {
            double2 s[8];
#pragma unroll
            for (int q = 0; q < L; q++) {
                const double2* sp = sT_all + (oh * L + q) * kTile + lane;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double2 y = sp[k * 64];
                    s[k] = q == 0 ? cmulx<false>(y, bk[0][k])
                                  : make_double2(fma(y.x, bk[q][k].x, fma(-y.y, bk[q][k].y, s[k].x)), fma(y.x, bk[q][k].y, fma(y.y, bk[q][k].x, s[k].y)));
                }
            }
            fft512_inverse<true, 0>(s, sTi, lane, R);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool watched = GUARD == 1 || (GUARD == 2 && (r & 3) == 0);
                const double2 z = r == 0 ? make_double2(s[0].x * (1.0 / 512.0), s[0].y * (1.0 / 512.0)) : cmulx<true>(s[r], untwist_reg(r));
                const double t0 = z.x + kMagic, t1 = z.y + kMagic;
                if (watched) dev_max = fmax(dev_max, fmax(fabs(z.x - (t0 - kMagic)), fabs(z.y - (t1 - kMagic))));
                const int32_t j = 64 * r + lane;
                atomicAdd(&acco[j], (uint32_t)__double2loint(t0));  // ds_add_u32; the other half of this output adds to the same word
                atomicAdd(&acco[j + kM], (uint32_t)__double2loint(t1));
            }
        }
        __syncthreads();  // C: accumulator complete before the next decomposition; every published spectrum consumed
    }
    if (GUARD && is_out) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += NT)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
    }
}

// ---- K3 (+K4), one to two gates per CU, round 3: k_blind_rotate_wide4 built so that TWO workgroups share a CU ----
// Launches of 257 .. 512 gate instances: one workgroup of 2L waves per gate as in k_blind_rotate_wide4, but at most 168 VGPRs
// (three waves per SIMD) and 74 KB of LDS (the inverse transforms reuse the published tiles after a barrier B instead of
// scratch tiles of their own), so that two gates are resident per CU and fill each other's waits.  An output wave's L BK
// blocks are requested after its forward transform (they do not fit the register budget next to it) and arrive under
// barrier A and the other workgroup's work, two in flight at a time.
// k_blind_rotate_wide on the one-limb spectrum leaves the six row products and the inverse transform of an output
// polynomial to ONE wave (two output waves; the other four idle for half of the step), and its tiles serve both as the
// published spectra and as the inverse transforms' scratch (barrier B).  Here waves 0..3 are output waves (one per SIMD):
// output wave (c, h) = (w & 1, w >> 1) multiplies the L published spectra of accumulator polynomial h with block c of
// their BK rows -- its L blocks are requested at the top of the step and arrive under the decomposition and the forward
// transform --, inverse-transforms that PARTIAL sum in a scratch tile of its own (no barrier B), rounds it and adds it
// into accumulator polynomial c with ds_add_u32.  Each partial sum is an integer polynomial and addition mod 2^32
// commutes, so the two halves of an output need no ordering.  Two barriers per step (spectra published / accumulator
// updated).  Same rounded integers as every other kernel here.
// dynamic LDS: acc [2][1024] int32 | sT [2L][kTile] double2 | tw [kTwElems] double2 | bara [i1-i0] u16
template <int L, int BGBIT, int GUARD>
__global__ __launch_bounds__(128 * L, 3) void k_blind_rotate_wide4b(DevKeys K, const double2* __restrict__ bkf1,
                                                              const uint16_t* __restrict__ st_bara, int32_t nb,
                                                              int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                              unsigned* guard, const double2* __restrict__ gtw) {
    constexpr int NW = 2 * L, NT = 64 * NW;
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);
    double2* sT_all = reinterpret_cast<double2*>(smem + (size_t)2 * kN * 4);
    double2* sTw = sT_all + NW * kTile;
    uint16_t* s_bara = reinterpret_cast<uint16_t*>(sTw + kTwElems);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;                  // forward scratch, then this wave's published spectrum
    const int64_t item = (int64_t)blockIdx.x;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, NT);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
        const uint16_t* bara = st_bara + (size_t)item * nb;
        for (int idx = tid; idx < i1 - i0; idx += NT) s_bara[idx] = bara[i0 + idx];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;
    const int pw = wave / L, qw = wave - pw * L;  // forward role: digit qw of polynomial pw = row `wave` of BK_i
    const int sh = 32 - (qw + 1) * BGBIT;
    const bool is_out = wave < 4;
    const int oc = wave & 1, oh = (wave >> 1) & 1;  // output role: block oc of the rows of polynomial oh
    uint32_t* acco = reinterpret_cast<uint32_t*>(acc) + oc * kN;
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc);
    const uint32_t pb = (uint32_t)pw * (kN * 4);
    const int32_t* accp = acc + pw * kN;
    double dev_max = 0.0;

#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readfirstlane((int32_t)s_bara[i - i0]);
        if (a == 0) continue;  // workgroup-uniform
        // BK_i rows [2L][2][8][64]: this output wave's L blocks, all requested now
        const double2* __restrict__ bki = bkf1 + (size_t)i * (2 * L * 2 * kM) + (size_t)(oh * L) * (2 * kM) + (size_t)oc * kM + lane;
        int32_t lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque: keeps the per-coefficient LDS addresses from being hoisted out of the step loop
        const uint32_t jb4 = ((uint32_t)(lane_o - a) & (2 * kN - 1)) << 2;
        uint32_t rv0[8], rv1[8], pv0[8], pv1[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {   // all 32 LDS reads first, then the arithmetic
            const uint32_t t = jb4 + 256u * r;
            const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
            rv0[r] = *reinterpret_cast<const uint32_t*>(accb + o0);
            rv1[r] = *reinterpret_cast<const uint32_t*>(accb + o1);
            pv0[r] = (uint32_t)accp[64 * r + lane_o];
            pv1[r] = (uint32_t)accp[64 * r + lane_o + kM];
        }
        __builtin_amdgcn_sched_barrier(0);
        double2 x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t t = jb4 + 256u * r;
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
            const uint32_t u0 = (rv0[r] ^ m0) + ((dec_offset - pv0[r]) - m0);
            const uint32_t u1 = (rv1[r] ^ m1) + ((dec_offset - pv1[r]) - m1);
            // digit - halfBg = sign-extended field of (u ^ (halfBg << sh))
            const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)(u0 ^ (halfBg << sh)), sh, BGBIT);
            const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)(u1 ^ (halfBg << sh)), sh, BGBIT);
            x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                          : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
        }
        fft512_forward<true, 0>(x, sT, lane, R);
#pragma unroll
        for (int k = 0; k < 8; k++) sT[k * 64 + lane] = x[k];  // publish
        __builtin_amdgcn_sched_barrier(0);
        double2 bka[8], bkb[8];  // two of this output wave's L BK blocks in flight at a time (all L do not fit 168 registers)
        if (is_out) {
#pragma unroll
            for (int k = 0; k < 8; k++) bka[k] = bki[k * 64];
        }
        __syncthreads();  // A: all 2L spectra are in their tiles
        double2 s[8];
        if (is_out) {
#pragma unroll
            for (int q = 0; q < L; q++) {
                double2 (&cur)[8] = (q & 1) ? bkb : bka;
                double2 (&nxt)[8] = (q & 1) ? bka : bkb;
                if (q + 1 < L) {
#pragma unroll
                    for (int k = 0; k < 8; k++) nxt[k] = bki[(size_t)(q + 1) * (2 * kM) + k * 64];
                }
                const double2* sp = sT_all + (oh * L + q) * kTile + lane;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double2 y = sp[k * 64];
                    s[k] = q == 0 ? cmulx<false>(y, cur[k])
                                  : make_double2(fma(y.x, cur[k].x, fma(-y.y, cur[k].y, s[k].x)), fma(y.x, cur[k].y, fma(y.y, cur[k].x, s[k].y)));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // B: every published spectrum consumed; the tiles are scratch again
        if (is_out) {
            fft512_inverse<true, 0>(s, sT, lane, R);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool watched = GUARD == 1 || (GUARD == 2 && (r & 3) == 0);
                const double2 z = r == 0 ? make_double2(s[0].x * (1.0 / 512.0), s[0].y * (1.0 / 512.0)) : cmulx<true>(s[r], untwist_reg(r));
                const double t0 = z.x + kMagic, t1 = z.y + kMagic;
                if (watched) dev_max = fmax(dev_max, fmax(fabs(z.x - (t0 - kMagic)), fabs(z.y - (t1 - kMagic))));
                const int32_t j = 64 * r + lane;
                atomicAdd(&acco[j], (uint32_t)__double2loint(t0));  // ds_add_u32; the other half of this output adds to the same word
                atomicAdd(&acco[j + kM], (uint32_t)__double2loint(t1));
            }
        }
        __syncthreads();  // C: accumulator complete before the next decomposition; every published spectrum consumed
    }
    if (GUARD && is_out) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += NT)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
    }
}

// ---- K3 (+K4), latency-oriented on the ONE-limb spectrum: 2L waves per gate, every wave a whole row ----
// k_blind_rotate_wide hands all 2L spectra to four output waves (two barriers, 192 KiB of LDS reads, 192 KiB of BK through
// one CU per step).  The inverse transform is linear and every row's product digit_row (*) BK_row is itself an integer
// polynomial, so here wave w keeps its spectrum in registers, multiplies it with BOTH output blocks of its own BK row,
// inverse-transforms the two products itself (interleaved) and adds the rounded coefficients into the accumulator with
// ds_add_u32 (addition mod 2^32 commutes).  No spectrum crosses waves; two barriers per step (accumulator read / written);
// 96 KiB of BK per step, requested before the decomposition.  2L forward + 4L inverse transforms instead of 2L + 2, on
// six waves that would otherwise wait for each other.  Each partial product is 1/2L of the full sum, so the rounding
// margin of section 2 only grows; the guard is the same.
// MEASURED (n = 630, one gate per CU): 3.9-4.1 ms per blind rotation against 3.35-3.67 ms for k_blind_rotate_wide -- six
// waves on four SIMDs put two whole rows (~1 000 vector instructions each) on two of them, and that serial vector work
// (8 k cycles per step) is longer than the hand-overs it removes.  Kept selectable (br_variant 22 / 23), not used by default.
// dynamic LDS: sT [2L][kTile] double2 | tw [kTwElems] double2 | acc [2][1024] int32 | bara [i1-i0] u16
template <int L, int BGBIT, bool GUARD>
__global__ __launch_bounds__(128 * L) void k_blind_rotate_wide1(DevKeys K, const double2* __restrict__ bkf1,
                                                              const uint16_t* __restrict__ st_bara, int32_t nb,
                                                              int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                              unsigned* guard, const double2* __restrict__ gtw) {
    constexpr int NW = 2 * L, NT = 64 * NW;
    extern __shared__ __align__(16) unsigned char smem[];
    double2* sT_all = reinterpret_cast<double2*>(smem);
    double2* sTw = sT_all + NW * kTile;
    int32_t* acc = reinterpret_cast<int32_t*>(sTw + kTwElems);
    uint16_t* s_bara = reinterpret_cast<uint16_t*>(acc + 2 * kN);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2* sT = sT_all + wave * kTile;
    const int64_t item = (int64_t)blockIdx.x;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    load_twiddles(sTw, gtw, tid, NT);
    const LaneRoots R = make_roots(sTw, lane);
    {
        const int4* src = reinterpret_cast<const int4*>(gacc);
        int4* dst = reinterpret_cast<int4*>(acc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
        const uint16_t* bara = st_bara + (size_t)item * nb;
        for (int idx = tid; idx < i1 - i0; idx += NT) s_bara[idx] = bara[i0 + idx];
    }
    __syncthreads();

    constexpr uint32_t halfBg = 1u << (BGBIT - 1);
    uint32_t dec_offset = 0;
#pragma unroll
    for (int q = 1; q <= L; q++) dec_offset += halfBg << (32 - q * BGBIT);
    constexpr double kMagic = 6755399441055744.0;
    const int pw = wave / L, qw = wave - pw * L;  // digit qw of polynomial pw = row `wave` of BK_i
    const int sh = 32 - (qw + 1) * BGBIT;
    const int32_t* accp = acc + pw * kN;
    uint32_t* accu = reinterpret_cast<uint32_t*>(acc);
    double dev_max = 0.0;

#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readfirstlane((int32_t)s_bara[i - i0]);
        if (a == 0) continue;  // workgroup-uniform
        // BK_i rows [2L][2][8][64]: this wave's row, both output blocks, requested before anything else
        const double2* __restrict__ bki = bkf1 + (size_t)i * (2 * L * 2 * kM) + (size_t)wave * (2 * kM) + lane;
        double2 s0[8], s1[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            s0[k] = bki[k * 64];
            s1[k] = bki[(8 + k) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        int32_t lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque: keeps 16 per-coefficient LDS addresses from being hoisted (and spilled)
        uint32_t rv0[8], rv1[8], pv0[8], pv1[8];
        const int32_t jb = (lane_o - a) & (2 * kN - 1);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int32_t j = 64 * r + lane_o;
            rv0[r] = (uint32_t)accp[(jb + 64 * r) & (kN - 1)];
            rv1[r] = (uint32_t)accp[(jb + 64 * r + kM) & (kN - 1)];
            pv0[r] = (uint32_t)accp[j];
            pv1[r] = (uint32_t)accp[j + kM];
        }
        __builtin_amdgcn_sched_barrier(0);
        double2 x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t n0 = 0u - (((uint32_t)(jb + 64 * r) >> 10) & 1u);  // all ones where the rotation wrapped
            const uint32_t n1 = 0u - (((uint32_t)(jb + 64 * r + kM) >> 10) & 1u);
            const uint32_t u0 = ((rv0[r] ^ n0) - n0) - pv0[r] + dec_offset;
            const uint32_t u1 = ((rv1[r] ^ n1) - n1) - pv1[r] + dec_offset;
            const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)(u0 ^ (halfBg << sh)), sh, BGBIT);
            const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)(u1 ^ (halfBg << sh)), sh, BGBIT);
            x[r] = r == 0 ? make_double2((double)e0, (double)e1)
                          : cmulx<false>(make_double2((double)e0, (double)e1), twist_reg(r));
        }
        __syncthreads();  // A: every wave has read the accumulator; from here on it may be added to
        fft512_forward<true, 1>(x, sT, lane, R);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double2 b0 = s0[k], b1 = s1[k];
            s0[k] = cmulx<false>(x[k], b0);
            s1[k] = cmulx<false>(x[k], b1);
        }
        fft512_inverse_pair<true>(s0, s1, sT, lane, R);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double2 z0 = r == 0 ? make_double2(s0[0].x * (1.0 / 512.0), s0[0].y * (1.0 / 512.0)) : cmulx<true>(s0[r], untwist_reg(r));
            const double2 z1 = r == 0 ? make_double2(s1[0].x * (1.0 / 512.0), s1[0].y * (1.0 / 512.0)) : cmulx<true>(s1[r], untwist_reg(r));
            const double t00 = z0.x + kMagic, t01 = z0.y + kMagic, t10 = z1.x + kMagic, t11 = z1.y + kMagic;
            if (GUARD) {
                dev_max = fmax(dev_max, fmax(fabs(z0.x - (t00 - kMagic)), fabs(z0.y - (t01 - kMagic))));
                dev_max = fmax(dev_max, fmax(fabs(z1.x - (t10 - kMagic)), fabs(z1.y - (t11 - kMagic))));
            }
            const int32_t j = 64 * r + lane;
            atomicAdd(&accu[j], (uint32_t)__double2loint(t00));  // ds_add_u32: the 2L waves add their shares in any order
            atomicAdd(&accu[j + kM], (uint32_t)__double2loint(t01));
            atomicAdd(&accu[kN + j], (uint32_t)__double2loint(t10));
            atomicAdd(&accu[kN + j + kM], (uint32_t)__double2loint(t11));
        }
        __syncthreads();  // B: accumulator complete before the next decomposition
    }
    if (GUARD) {
        float m = (float)dev_max;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __builtin_nontemporal_load(&guard[1])) atomicMax(&guard[1], bits);
            if (m > kGuardLimit) atomicAdd(&guard[0], 1u);
        }
    }
    if (ext) {
        Torus32* u = ext + (size_t)item * (kN + 4);
        for (int32_t j = tid; j <= kN; j += NT)
            u[j] = j == 0 ? acc[0] : (j == kN ? acc[kN] : (int32_t)(0u - (uint32_t)acc[kN - j]));
    } else {
        const int4* src = reinterpret_cast<const int4*>(acc);
        int4* dst = reinterpret_cast<int4*>(gacc);
        for (int idx = tid; idx < 2 * kN / 4; idx += NT) dst[idx] = src[idx];
    }
}

}  // namespace

// N=1024, k=1 with either libtfhe parameter set: l=3/Bgbit=7 (>= v1.1, "128-bit") or l=2/Bgbit=10
// (v1.0 and the paper's 78 MiB keys).  Exactness margin for the latter: 4 rows x 1024 x 512 x 2^15 < 2^37.
bool supported(const Params& p) {
    return p.N == kN && p.k == 1 && ((p.l == 3 && p.Bgbit == 7) || (p.l == 2 && p.Bgbit == 10)) && p.n <= 4096;
}

// The one-limb kernels round sums of up to 2l x N x 2^(Bgbit-1) x 2^31: 2^49.6 for l=3 / Bgbit=7, where the measured
// rounding error is 35x below the guard's limit.  For l=2 / Bgbit=10 the worst case is 2^52 and the typical error 6.5x
// larger -- inside 0.5 but no longer clear of the limit -- so that set stays on the two-limb kernels.
bool one_limb_supported(const Params& p) { return supported(p) && p.l == 3 && p.Bgbit == 7; }

size_t spectrum_elems(const Params& p) { return (size_t)p.n * p.kpl() * 4 * kM; }
size_t spectrum1_elems(const Params& p) { return (size_t)p.n * p.kpl() * 2 * kM; }
size_t lds_bytes_w1() { return (size_t)(kW1Gates * kTile + kTwElems) * sizeof(double2) + (size_t)kW1Gates * 2 * kN * 4; }
int gates_per_workgroup_w1() { return kW1Gates; }

size_t lds_bytes(const Params& p) {
    (void)p;
    return (size_t)(2 * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4;
}

int32_t bara_stride(const Params& p) { return (p.n + 7) & ~7; }

size_t lds_bytes_wide(const Params& p) {
    return (size_t)(2 * p.l * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)bara_stride(p) * 2;
}

size_t state_bytes_per_item(const Params& p) { return (size_t)bara_stride(p) * 2 + (size_t)2 * kN * 4; }

size_t twiddle_table_elems() { return kTwElems; }

void build_twiddle_table(double2* d_tw, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_twiddle_table, dim3(1), dim3(128), 0, stream, d_tw);
}

void prepare_spectrum(const Params& p, const Torus32* d_bk_raw, double2* d_bkf, hipStream_t stream) {
    const size_t npoly = (size_t)p.n * p.kpl() * 2;
    hipLaunchKernelGGL(k_bk_to_spectrum_w64, dim3((unsigned)npoly), dim3(64), 0, stream, d_bk_raw, d_bkf);
}

void prepare_spectrum1(const Params& p, const Torus32* d_bk_raw, double2* d_bkf1, hipStream_t stream) {
    const size_t npoly = (size_t)p.n * p.kpl() * 2;
    hipLaunchKernelGGL(k_bk_to_spectrum_w64_1, dim3((unsigned)npoly), dim3(64), 0, stream, d_bk_raw, d_bkf1);
}

static int32_t device_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}

// diagnostic build (IEACHE_BR_VARIANT=1): per-segment s_memtime sums, printed per launch() call
static unsigned long long* diag_buf() {
    static unsigned long long* p = nullptr;
    if (!p) {
        if (hipMalloc(&p, 16 * sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(p, 0, 16 * sizeof(unsigned long long)) != hipSuccess) {
            p = nullptr;
            throw std::runtime_error("hipMalloc failed for the blind-rotation diagnostic buffer");
        }
    }
    return p;
}
static void diag_report(hipStream_t stream, int64_t items, int32_t nsteps, int variant) {
    unsigned long long h[16];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, diag_buf(), sizeof h, hipMemcpyDeviceToHost);
    (void)hipMemset(diag_buf(), 0, sizeof h);
    static const char* names_w2[8] = {"decompose", "build+fwdFFT(x3)", "tile+ownBK+MAC(x3)", "barrier1(x3)", "partner+BK+MAC(x3)",
                                      "barrier2(x3)", "invFFT x2+update", "end barrier"};
    static const char* names_wide[8] = {"head+decompose", "fwdFFT", "publish+BK issue", "barrier A", "MAC rows",
                                        "barrier B", "invFFT+update", "barrier C"};  // "wave 1" = wave 4 (no output role)
    const char* const* names = variant >= kVariantWide ? names_wide : names_w2;
    const double denom = (double)items * (nsteps > 0 ? nsteps : 1);
    for (int w = 0; w < 2; w++) {
        double tot = 0;
        for (int t = 0; t < 8; t++) tot += (double)h[w * 8 + t];
        fprintf(stderr, "[br-diag] wave %d: %.0f memtime ticks per step:", w, tot / denom);
        for (int t = 0; t < 8; t++) fprintf(stderr, " %s=%.0f", names[t], (double)h[w * 8 + t] / denom);
        fprintf(stderr, "\n");
    }
}

template <int L, int BGBIT>
static void launch_slice(int variant, dim3 grid, dim3 blk, size_t lds, hipStream_t stream, const DevKeys& K, const double2* d_bkf,
                         const uint16_t* st_bara, int32_t nb, int32_t* st_acc, int32_t i0, int32_t i1, Torus32* e,
                         const double2* gtw) {
    unsigned long long* const nodiag = nullptr;
    if (variant == kVariantWide || variant == kVariantWide + 1) {
        static const bool attr_set = [] {  // > 64 KiB of dynamic LDS has to be allowed explicitly
            return hipFuncSetAttribute((const void*)k_blind_rotate_wide<L, BGBIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024) == hipSuccess &&
                   hipFuncSetAttribute((const void*)k_blind_rotate_wide<L, BGBIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024) == hipSuccess;
        }();
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_wide");
        const size_t lds_wide = (size_t)(2 * L * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)nb * 2;
        if (variant == kVariantWide)
            hipLaunchKernelGGL((k_blind_rotate_wide<L, BGBIT, false>), grid, dim3(128 * L), lds_wide, stream, K, d_bkf, st_bara, nb, st_acc, i0, i1, e, nodiag, gtw, (unsigned*)nullptr);
        else
            hipLaunchKernelGGL((k_blind_rotate_wide<L, BGBIT, true>), grid, dim3(128 * L), lds_wide, stream, K, d_bkf, st_bara, nb, st_acc, i0, i1, e, diag_buf(), gtw, (unsigned*)nullptr);
        return;
    }
    if (variant == kVariantExactOneWave || variant == kVariantExactOneWave + 1) {  // round 4: one wave per gate on the two-limb spectrum
        const int64_t items = (int64_t)grid.x;
        const dim3 g1((unsigned)((items + kW1Gates - 1) / kW1Gates)), b1(64 * kW1Gates);
#define IEACHE_X1(MODE)                                                                                                         \
    {                                                                                                                           \
        static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_x1<L, BGBIT, MODE>,                        \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_w1()) == hipSuccess; \
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_x1"); \
        hipLaunchKernelGGL((k_blind_rotate_x1<L, BGBIT, MODE>), g1, b1, lds_bytes_w1(), stream, K, d_bkf, st_bara, nb, st_acc, items, \
                           i0, i1, e, gtw);                                                                                     \
    }
        if (variant == kVariantExactOneWave) IEACHE_X1(0) else IEACHE_X1(2)  // + 1: BK blocks re-requested at block granularity (A/B partner)
#undef IEACHE_X1
        return;
    }
    switch (variant) {
        case 12: hipLaunchKernelGGL((k_blind_rotate_w2<L, BGBIT, false, true>), grid, blk, lds, stream, K, d_bkf, st_bara, nb, st_acc, i0, i1, e, nodiag, gtw); break;  // every transpose through LDS (round 1's default)
        // default since round 2: the forward transforms' first (lane-high) transpose cross-lane, everything else through LDS
        default: hipLaunchKernelGGL((k_blind_rotate_w2<L, BGBIT, false, true, 1, 0, false, true>), grid, blk, lds, stream, K, d_bkf, st_bara, nb, st_acc, i0, i1, e, nodiag, gtw); break;
    }
}

// workgroups i and i + period are taken to share a CU: the device's CU count (IEACHE_W4R_FLIP overrides; a huge value = never flip)
static int32_t w4r_flip_period() {
    static const int32_t v = [] {
        if (const char* e = getenv("IEACHE_W4R_FLIP")) return atoi(e) > 0 ? atoi(e) : 1 << 30;
        return device_cus();
    }();
    return v;
}

template <int L, int BGBIT>
static void launch_slice_w1(int sub, int64_t items, hipStream_t stream, const DevKeys& K, const double2* d_bkf1,
                            const uint16_t* st_bara, int32_t nb, int32_t* st_acc, int32_t i0, int32_t i1, Torus32* e,
                            unsigned* guard, const double2* gtw) {
    const dim3 grid((unsigned)((items + kW1Gates - 1) / kW1Gates)), blk(64 * kW1Gates);
#define IEACHE_W1(...)                                                                                                          \
    {                                                                                                                           \
        static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_w1<L, BGBIT, __VA_ARGS__>,                 \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_w1()) == hipSuccess; \
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_w1"); \
        hipLaunchKernelGGL((k_blind_rotate_w1<L, BGBIT, __VA_ARGS__>), grid, blk, lds_bytes_w1(), stream, K, d_bkf1, st_bara, nb, \
                           st_acc, items, i0, i1, e, guard, gtw);                                                               \
    }
    if (sub == 16) {  // ... with the s_memtime phase stamps (diagnostic)
        const size_t lds_wide = (size_t)(2 * L * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)nb * 2;
        static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_wide<L, BGBIT, true, 1>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_wide<1 limb, diag>");
        hipLaunchKernelGGL((k_blind_rotate_wide<L, BGBIT, true, 1>), dim3((unsigned)items), dim3(128 * L), lds_wide, stream, K, d_bkf1, st_bara,
                           nb, st_acc, i0, i1, e, diag_buf(), gtw, guard);
        return;
    }
    if (sub >= 11 && sub <= 15) {  // the latency kernel (2L waves per gate, spectra handed to the output waves) on the one-limb spectrum
        const size_t lds_wide = (size_t)(2 * L * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)nb * 2;
#define IEACHE_WIDE1(XF, XI)                                                                                                       \
    {                                                                                                                              \
        static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_wide<L, BGBIT, false, 1, XF, XI>,             \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;    \
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_wide<1 limb>"); \
        hipLaunchKernelGGL((k_blind_rotate_wide<L, BGBIT, false, 1, XF, XI>), dim3((unsigned)items), dim3(128 * L), lds_wide, stream, K, \
                           d_bkf1, st_bara, nb, st_acc, i0, i1, e, (unsigned long long*)nullptr, gtw, guard);                      \
    }
        switch (sub) {  // transposes cross-lane instead of through LDS: forward / inverse, lane-high (1), both (3)
            case 12: IEACHE_WIDE1(1, 0) break;
            case 13: IEACHE_WIDE1(3, 0) break;
            case 14: IEACHE_WIDE1(1, 1) break;
            case 15: IEACHE_WIDE1(3, 3) break;
            default: IEACHE_WIDE1(0, 0) break;
        }
#undef IEACHE_WIDE1
        return;
    }
    if (sub == 9 || sub == 10) {  // 2L waves per gate, every wave a whole row of the one-limb spectrum (latency); 10 = no guard arithmetic
        static const bool attr_set =
            hipFuncSetAttribute((const void*)k_blind_rotate_wide1<L, BGBIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
            hipFuncSetAttribute((const void*)k_blind_rotate_wide1<L, BGBIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_wide1");
        const size_t lds_wide = (size_t)(2 * L * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)nb * 2;
        if (sub == 9)
            hipLaunchKernelGGL((k_blind_rotate_wide1<L, BGBIT, true>), dim3((unsigned)items), dim3(128 * L), lds_wide, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        else
            hipLaunchKernelGGL((k_blind_rotate_wide1<L, BGBIT, false>), dim3((unsigned)items), dim3(128 * L), lds_wide, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        return;
    }
    if (sub == 7 || sub == 8) {  // two waves per gate on the one-limb spectrum (mid-size launches); 8 = without the guard arithmetic
        const dim3 g2((unsigned)items), b2(128);
        const size_t lds2 = (size_t)(2 * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4;
        if (sub == 7)
            hipLaunchKernelGGL((k_blind_rotate_w2s<L, BGBIT, true>), g2, b2, lds2, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        else
            hipLaunchKernelGGL((k_blind_rotate_w2s<L, BGBIT, false>), g2, b2, lds2, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        return;
    }
#define IEACHE_W1B(...)                                                                                                         \
    {                                                                                                                           \
        static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_w1b<L, BGBIT, __VA_ARGS__>,                \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_w1()) == hipSuccess; \
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_w1b"); \
        hipLaunchKernelGGL((k_blind_rotate_w1b<L, BGBIT, __VA_ARGS__>), grid, blk, lds_bytes_w1(), stream, K, d_bkf1, st_bara, nb, \
                           st_acc, items, i0, i1, e, guard, gtw);                                                               \
    }
    if (sub == 30 || sub == 31) {  // round 3: four waves per gate, rows 2:1:2:1 (launches of one to two gates per CU); 31 = guard on every coefficient
        const size_t lds4 = (size_t)(4 * kTile + 2 * 8 * 64 + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4;
        static const bool attr_set =
            hipFuncSetAttribute((const void*)k_blind_rotate_w4r<L, BGBIT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
            hipFuncSetAttribute((const void*)k_blind_rotate_w4r<L, BGBIT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_w4r");
        if (sub == 30)
            hipLaunchKernelGGL((k_blind_rotate_w4r<L, BGBIT, 2>), dim3((unsigned)items), dim3(256), lds4, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw, w4r_flip_period());
        else
            hipLaunchKernelGGL((k_blind_rotate_w4r<L, BGBIT, 1>), dim3((unsigned)items), dim3(256), lds4, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw, w4r_flip_period());
        return;
    }
    if (sub == 28) {  // round 3: k_blind_rotate_wide4 built for two workgroups per CU (launches of one to two gates per CU)
        const size_t lds_wide = (size_t)(2 * L * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)nb * 2;
        static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_wide4b<L, BGBIT, 2>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_wide4b");
        hipLaunchKernelGGL((k_blind_rotate_wide4b<L, BGBIT, 2>), dim3((unsigned)items), dim3(128 * L), lds_wide, stream, K, d_bkf1, st_bara, nb,
                           st_acc, i0, i1, e, guard, gtw);
        return;
    }
    if (sub == 25 || sub == 26) {  // round 3: latency kernel with four output waves on half the rows each; 26 = guard on every coefficient
        const size_t lds_w4 = (size_t)((2 * L + 4) * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4 + (size_t)nb * 2;
        static const bool attr_set =
            hipFuncSetAttribute((const void*)k_blind_rotate_wide4<L, BGBIT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
            hipFuncSetAttribute((const void*)k_blind_rotate_wide4<L, BGBIT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_wide4");
        if (sub == 25)
            hipLaunchKernelGGL((k_blind_rotate_wide4<L, BGBIT, 2>), dim3((unsigned)items), dim3(128 * L), lds_w4, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        else
            hipLaunchKernelGGL((k_blind_rotate_wide4<L, BGBIT, 1>), dim3((unsigned)items), dim3(128 * L), lds_w4, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        return;
    }
    if (sub == 23 || sub == 24) {  // round 3: two waves per gate, rows split (mid-size launches); 24 = guard on every coefficient
        const dim3 g2((unsigned)items), b2(128);
        const size_t lds2 = (size_t)(2 * kTile + kTwElems) * sizeof(double2) + (size_t)2 * kN * 4;
        if (sub == 23)
            hipLaunchKernelGGL((k_blind_rotate_w2r<L, BGBIT, 2>), g2, b2, lds2, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        else
            hipLaunchKernelGGL((k_blind_rotate_w2r<L, BGBIT, 1>), g2, b2, lds2, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, gtw);
        return;
    }
    switch (sub) {  // round 3: k_blind_rotate_w1b
        case 18: IEACHE_W1B(2, 0) return;   // guard on one coefficient in four
        case 19: IEACHE_W1B(1, 0) return;   // guard on every coefficient
        case 20: IEACHE_W1B(2, 1) return;   // polynomial 1's forward transposes through LDS
        case 21: IEACHE_W1B(2, 2) return;   // every second row's
        case 22: IEACHE_W1B(0, 0) return;   // no guard arithmetic (measurement)
        case 32: IEACHE_W1B(2, 4) return;   // polynomial 0's forward transforms with BOTH transposes cross-lane (no LDS round trip), polynomial 1's lane-low through LDS
        case 33: IEACHE_W1B(2, 5) return;   // all six
        case 36: {  // phase stamps (diagnostic): decomposition / digits+twist / forward transform / BK + products / inverse pair / update
            static const bool attr_set = hipFuncSetAttribute((const void*)k_blind_rotate_w1b<L, BGBIT, 2, 0, 0, true>,
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_w1()) == hipSuccess;
            if (!attr_set) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_blind_rotate_w1b<diag>");
            hipLaunchKernelGGL((k_blind_rotate_w1b<L, BGBIT, 2, 0, 0, true>), grid, blk, lds_bytes_w1(), stream, K, d_bkf1, st_bara, nb, st_acc, items, i0, i1,
                               e, guard, gtw, diag_buf());
            return;
        }
        case 37: IEACHE_W1B(2, 0, 0, false, true) return;   // L2 prefetch of the next step's BK blocks, one slice per wave
        case 34: IEACHE_W1B(2, 0, 1) return;   // first inter-pass twiddles from the global table instead of LDS
        case 35: IEACHE_W1B(2, 0, 2) return;   // both sets
        case 29: IEACHE_W1B(2, 3) return;   // rows software-pipelined: the next row's digits / twist under this row's last transpose
        default: break;
    }
#undef IEACHE_W1B
    switch (sub) {  // > 64 KiB of dynamic LDS has to be allowed explicitly, per instantiation
        case 1: IEACHE_W1(false) break;          // no guard arithmetic (measurement)
        case 2: IEACHE_W1(true, 0) break;        // forward transposes through LDS
        case 3: IEACHE_W1(true, 3) break;        // both forward transposes cross-lane
        case 4: IEACHE_W1(true, 1, 1) break;     // both BK blocks of a row requested before its transform
        case 5: IEACHE_W1(true, 1, 2) break;     // the second one from inside the transform
        case 6: IEACHE_W1(true, 1, 3) break;
        case 17: IEACHE_W1(true, 1, 0, false) break;  // BK blocks through global_load instead of buffer_load
        default: IEACHE_W1(true) break;
    }
#undef IEACHE_W1
}

int32_t default_variant() {
    static const int32_t v = getenv("IEACHE_BR_VARIANT") ? atoi(getenv("IEACHE_BR_VARIANT")) : 0;
    return v;
}

int32_t default_slice() {
    static const int32_t s = getenv("IEACHE_BR_SLICE") ? atoi(getenv("IEACHE_BR_SLICE")) : 16;
    return s > 0 ? (s < 64 ? s : 64) : 16;  // <= 64: one rotation amount per lane
}

int launch(const Params& p, const DevKeys& K, const double2* d_bkf, const double2* d_bkf1, unsigned* guard, const WorkDesc& W,
           int64_t items, void* state, Torus32* ext, int32_t steps, Torus32* dbg_acc, int32_t slice, int32_t variant,
           const double2* d_twiddles, hipStream_t stream) {
    const bool one_limb = variant >= kVariantOneLimb && variant <= kVariantOneLimb + 40;
    if (one_limb && (!d_bkf1 || !guard)) throw std::runtime_error("one-limb blind rotation without its spectrum / guard word");
    int launches = 0;
    const dim3 grid((unsigned)items), blk(128);
    // IEACHE_BR_LDS_PAD=<bytes>: measurement aid -- extra dynamic LDS per workgroup lowers the number of
    // resident workgroups per CU (35.8 KB -> 4; +6 KB -> 3; +18 KB -> 2), i.e. waves per SIMD, at unchanged code
    static const size_t lds_pad = getenv("IEACHE_BR_LDS_PAD") ? (size_t)atol(getenv("IEACHE_BR_LDS_PAD")) : 0;
    const size_t lds = lds_bytes(p) + lds_pad;
    const int32_t nb = bara_stride(p);
    // state block: [items][2][1024] int32 accumulators, then [items][nb] u16 rotation amounts
    int32_t* st_acc = reinterpret_cast<int32_t*>(state);
    uint16_t* st_bara = reinterpret_cast<uint16_t*>(st_acc + (size_t)items * 2 * kN);
    hipLaunchKernelGGL(k_br_prologue, grid, blk, 0, stream, K, W, st_bara, nb, st_acc);
    const int32_t nsteps = steps < 0 ? p.n : (steps < p.n ? steps : p.n);
    // the wide kernel keeps a slice's rotation amounts in LDS, so a slice may be the whole rotation
    const int32_t max_slice = (variant == kVariantWide || variant == kVariantWide + 1 || (variant >= kVariantWideOneLimb && variant <= kVariantWideOneLimb + 7) ||
                               variant == kVariantOneLimb + 25 || variant == kVariantOneLimb + 26 || variant == kVariantOneLimb + 28 ||
                               variant == kVariantOneLimbTwoWaves || variant == kVariantOneLimbTwoWaves + 1 ||
                               variant == kVariantOneLimbFourWaves || variant == kVariantOneLimbFourWaves + 1) ? nb : 64;
    const int32_t S = (slice >= 1 && slice <= max_slice) ? slice : default_slice();
    for (int32_t i0 = 0; i0 < nsteps; i0 += S) {
        const int32_t i1 = i0 + S < nsteps ? i0 + S : nsteps;
        Torus32* e = (i1 == nsteps) ? ext : nullptr;  // the last slice extracts instead of storing the accumulator
        launches++;
        if (one_limb) {
            if (p.l == 3)
                launch_slice_w1<3, 7>(variant - kVariantOneLimb, items, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, d_twiddles);
            else
                launch_slice_w1<2, 10>(variant - kVariantOneLimb, items, stream, K, d_bkf1, st_bara, nb, st_acc, i0, i1, e, guard, d_twiddles);
        } else if (p.l == 3)
            launch_slice<3, 7>(variant, grid, blk, lds, stream, K, d_bkf, st_bara, nb, st_acc, i0, i1, e, d_twiddles);
        else
            launch_slice<2, 10>(variant, grid, blk, lds, stream, K, d_bkf, st_bara, nb, st_acc, i0, i1, e, d_twiddles);
    }
    if (variant == kVariantOneLimb + 36) {
        unsigned long long h[16];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, diag_buf(), sizeof h, hipMemcpyDeviceToHost);
        (void)hipMemset(diag_buf(), 0, sizeof h);
        static const char* names[6] = {"decomposition (x2)", "digits+cvt+twist (x6)", "forward transform (x6)", "2nd BK block + products (x6)", "inverse pair", "round+update"};
        const double denom = (double)items * (nsteps > 0 ? nsteps : 1) / 2.0;  // each of the two slots collects half of the waves
        for (int w = 0; w < 2; w++) {
            double tot = 0;
            for (int t = 0; t < 6; t++) tot += (double)h[w * 8 + t];
            fprintf(stderr, "[br-diag w1b] waves %d mod 2: %.0f memtime ticks per step:", w, tot / denom);
            for (int t = 0; t < 6; t++) fprintf(stderr, " %s=%.0f", names[t], (double)h[w * 8 + t] / denom);
            fprintf(stderr, "\n");
        }
    }
    if (variant == 1 || variant == 4 || variant == kVariantWide + 1 || variant == kVariantWideOneLimb + 7) diag_report(stream, items, nsteps, variant == kVariantWideOneLimb + 7 ? kVariantWide + 1 : variant);
    if (nsteps == 0 && ext) {
        // degenerate (steps == 0): extraction straight from the initial accumulator
        if (p.l == 3)
            launch_slice<3, 7>(12, grid, blk, lds, stream, K, d_bkf, st_bara, nb, st_acc, 0, 0, ext, d_twiddles);
        else
            launch_slice<2, 10>(12, grid, blk, lds, stream, K, d_bkf, st_bara, nb, st_acc, 0, 0, ext, d_twiddles);
    }
    if (dbg_acc)
        (void)hipMemcpyAsync(dbg_acc, st_acc, (size_t)items * 2 * kN * 4, hipMemcpyDeviceToDevice, stream);
    return launches;
}

}  // namespace w64
}  // namespace ieache
