// Blind rotation for launches of at most one gate per CU, 4L = 12 waves per gate (round 4): k_blind_rotate_wide12.
//
// k_blind_rotate_wide4 gives each of the 2L digit rows ONE wave and a 512-point transform of 8 points per lane: six waves on
// four SIMDs (two SIMDs carry two transforms, two carry one) and then four lone output waves, each issuing one instruction
// every 6-7 cycles.  A CMux step is a dependency chain, so with one gate per CU the only way to fill the SIMDs is to cut the
// step's transforms into more, shorter pieces.  Here every polynomial of C[X]/(X^512 - i) (the folded negacyclic ring) is
// split into its even and odd coefficients, c(X) = c_e(Y) + X c_o(Y) with Y = X^2, Y^256 = i:
//     (a b)_e = a_e b_e + Y a_o b_o        (a b)_o = a_e b_o + a_o b_e
// and each half goes through a 256-point transform of FOUR points per lane (radix 4 x 4 x 4 x 4, all three register <-> lane
// transposes cross-lane: v_permlane32/16_swap, v_cndmask_b32_dpp).  Forward phase: 4L waves, one per (digit row, parity) --
// three per SIMD, each with half of wide4's decomposition and a transform of a third of its instructions.  Output phase:
// eight waves, one per (output polynomial c, row half, output parity): 2L point-wise products with the key's half spectra
// (KA and KB, precomputed: k_bk_to_spectrum_w12; BK_i is copied into LDS by LDS-DMA at the top of the step), ONE 256-point inverse, rounding, ds_add_u32 into the accumulator --
// every partial sum is an integer polynomial, so the two row halves of an output need no ordering (as in wide4).
// Two workgroup barriers per step.  Same rounded integers as every other kernel (tests: variants 40 / 41).
// wide12_model.py (this directory) states the index maps in numpy and checks them against the negacyclic product.
// ATTIC (measured and lost: profiles/r4_narrow_ab.txt).  Included by blind_rotate_attic.hip inside namespace ieache::w64::{anonymous};
// in the product build of commit 3e9995d it was included by blind_rotate_w64.hip the same way, with br_variant 40 / 41 / 42.
#pragma once

constexpr int kQ = 256;  // points of a half transform

// per-lane twiddles of the three inter-pass multiplications (13 complex values, loop-invariant registers)
//   t1[k] = theta^lane W256^(lane k), theta = e^{i pi/512}: the lane part of the half ring's twist rides on the first set
//   t2[k-1] = W64^(k (lane & 15))        t3[k-1] = W16^(k (lane & 3))
struct Roots256 {
    double2 t1[4], t2[3], t3[3];
};
__device__ __forceinline__ Roots256 make_roots256(int lane) {
    Roots256 R;
    double s, c;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        sincospi((double)(lane * (1 + 4 * k)) / 512.0, &s, &c);
        R.t1[k] = make_double2(c, s);
    }
#pragma unroll
    for (int k = 1; k < 4; k++) {
        sincospi((double)(k * (lane & 15)) / 32.0, &s, &c);
        R.t2[k - 1] = make_double2(c, s);
        sincospi((double)(k * (lane & 3)) / 8.0, &s, &c);
        R.t3[k - 1] = make_double2(c, s);
    }
    return R;
}

// radix 4 over the register index: u[k] = sum_a (+-i)^(a k) v[a]   (16 additions)
template <bool INV>
__device__ __forceinline__ void bfly4(double2 (&v)[4]) {
    const double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = csub(v[1], v[3]);
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    const double2 p = make_double2(t1.x - t3.y, t1.y + t3.x), m = make_double2(t1.x + t3.y, t1.y - t3.x);  // t1 +- i t3
    v[1] = INV ? m : p;
    v[3] = INV ? p : m;
}

// One double2 of register `lo` against the same of register `hi`, across lane bit B = 2, 1, 0 (fft512.h has 5, 4, 3):
// new_hi = set ? hi : lo[partner], new_lo = set ? hi[partner] : lo, set = lane bit B.  row_ror:n hands lane l the value of lane
// (l - n) mod 16 of its row, so the clear lanes' partner l + 4 is row_ror:12 and the set lanes' partner l - 4 is row_ror:4.
#define IEACHE_SWAP4_DPP(NAME, MASK, CTRL_UP, CTRL_DOWN)                                                                        \
    __device__ __forceinline__ void NAME(unsigned(&a)[4], unsigned(&b)[4]) {                                                   \
        unsigned na0, na1, na2, na3, nb0, nb1, nb2, nb3;                                                                       \
        asm volatile("s_mov_b32 vcc_lo, " MASK "\n\t"                                                                          \
                     "s_mov_b32 vcc_hi, " MASK "\n\t"                                                                          \
                     "s_nop 1\n\t"                                                                                             \
                     "v_cndmask_b32_dpp %4, %8, %12, vcc " CTRL_UP " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_cndmask_b32_dpp %5, %9, %13, vcc " CTRL_UP " row_mask:0xf bank_mask:0xf\n\t"                           \
                     "v_cndmask_b32_dpp %6, %10, %14, vcc " CTRL_UP " row_mask:0xf bank_mask:0xf\n\t"                          \
                     "v_cndmask_b32_dpp %7, %11, %15, vcc " CTRL_UP " row_mask:0xf bank_mask:0xf\n\t"                          \
                     "s_not_b64 vcc, vcc\n\t"                                                                                  \
                     "v_cndmask_b32_dpp %0, %12, %8, vcc " CTRL_DOWN " row_mask:0xf bank_mask:0xf\n\t"                         \
                     "v_cndmask_b32_dpp %1, %13, %9, vcc " CTRL_DOWN " row_mask:0xf bank_mask:0xf\n\t"                         \
                     "v_cndmask_b32_dpp %2, %14, %10, vcc " CTRL_DOWN " row_mask:0xf bank_mask:0xf\n\t"                        \
                     "v_cndmask_b32_dpp %3, %15, %11, vcc " CTRL_DOWN " row_mask:0xf bank_mask:0xf"                            \
                     : "=&v"(na0), "=&v"(na1), "=&v"(na2), "=&v"(na3), "=&v"(nb0), "=&v"(nb1), "=&v"(nb2), "=&v"(nb3)          \
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])                  \
                     : "vcc");                                                                                                 \
        a[0] = na0, a[1] = na1, a[2] = na2, a[3] = na3;                                                                        \
        b[0] = nb0, b[1] = nb1, b[2] = nb2, b[3] = nb3;                                                                        \
    }
IEACHE_SWAP4_DPP(swap4_lane_bit2, "0xf0f0f0f0", "row_ror:12", "row_ror:4")
IEACHE_SWAP4_DPP(swap4_lane_bit1, "0xcccccccc", "quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]")
IEACHE_SWAP4_DPP(swap4_lane_bit0, "0xaaaaaaaa", "quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]")
#undef IEACHE_SWAP4_DPP

// register bit M (1 or 2) <-> lane bit B
template <int M, int B>
__device__ __forceinline__ void bitswap4(double2 (&x)[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r & M) continue;
        unsigned a[4] = {(unsigned)__double2loint(x[r].x), (unsigned)__double2hiint(x[r].x), (unsigned)__double2loint(x[r].y),
                         (unsigned)__double2hiint(x[r].y)};
        unsigned b[4] = {(unsigned)__double2loint(x[r | M].x), (unsigned)__double2hiint(x[r | M].x), (unsigned)__double2loint(x[r | M].y),
                         (unsigned)__double2hiint(x[r | M].y)};
        if constexpr (B >= 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) swap_dwords<B>(a[q], b[q]);
        } else if constexpr (B == 3) {
            swap4_row_ror8(a, b);
        } else if constexpr (B == 2) {
            swap4_lane_bit2(a, b);
        } else if constexpr (B == 1) {
            swap4_lane_bit1(a, b);
        } else {
            swap4_lane_bit0(a, b);
        }
        x[r] = make_double2(__hiloint2double((int)a[1], (int)a[0]), __hiloint2double((int)a[3], (int)a[2]));
        x[r | M] = make_double2(__hiloint2double((int)b[1], (int)b[0]), __hiloint2double((int)b[3], (int)b[2]));
    }
}
// the two bits of the register index <-> lane bits (HI, HI - 1)
template <int HI>
__device__ __forceinline__ void xpose4(double2 (&x)[4]) {
    bitswap4<2, HI>(x);
    bitswap4<1, HI - 1>(x);
}

// Forward half transform.
//   in : v[a] = x[64 a + lane], untwisted (the register part theta^(64 a) = e^{i pi a/8} is applied here, the lane part rides on t1)
//   out: v[g] = F[k(g, lane)],  k = (lane >> 4) + 4 ((lane >> 2) & 3) + 16 (lane & 3) + 64 g,  F[k] = sum_n x[n] theta^n W256^(n k)
template <class MID = NoHook>
__device__ __forceinline__ void fwd256(double2 (&v)[4], const Roots256& R, MID mid = MID()) {
    v[1] = make_double2(kCos8 * fma(-kTan8, v[1].y, v[1].x), kCos8 * fma(kTan8, v[1].x, v[1].y));           // e^{i pi/8}
    v[2] = make_double2(kR * (v[2].x - v[2].y), kR * (v[2].x + v[2].y));                                    // e^{i pi/4}
    v[3] = make_double2(kCos8 * fma(kTan8, v[3].x, -v[3].y), kCos8 * fma(kTan8, v[3].y, v[3].x));           // e^{3 i pi/8} = i e^{-i pi/8}
    bfly4<false>(v);  // over a -> ka
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = cmulx<false>(v[k], R.t1[k]);
    xpose4<5>(v);     // register = b, lane = (ka, c, d)
    if (!std::is_same<MID, NoHook>::value) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
    }
    bfly4<false>(v);  // over b -> kb
#pragma unroll
    for (int k = 1; k < 4; k++) v[k] = cmulx<false>(v[k], R.t2[k - 1]);
    xpose4<3>(v);     // register = c, lane = (ka, kb, d)
    bfly4<false>(v);  // over c -> kc
#pragma unroll
    for (int k = 1; k < 4; k++) v[k] = cmulx<false>(v[k], R.t3[k - 1]);
    xpose4<1>(v);     // register = d, lane = (ka, kb, kc)
    bfly4<false>(v);  // over d -> kd
}

// Its mirror (unnormalised: 256 x), the lane part of the untwist included:
//   out: v[a] = 256 x[64 a + lane] e^{i pi a/8}  -- the register part of the untwist is left to the caller (untwist256)
__device__ __forceinline__ void inv256(double2 (&v)[4], const Roots256& R) {
    bfly4<true>(v);
    xpose4<1>(v);
#pragma unroll
    for (int k = 1; k < 4; k++) v[k] = cmulx<true>(v[k], R.t3[k - 1]);
    bfly4<true>(v);
    xpose4<3>(v);
#pragma unroll
    for (int k = 1; k < 4; k++) v[k] = cmulx<true>(v[k], R.t2[k - 1]);
    bfly4<true>(v);
    xpose4<5>(v);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = cmulx<true>(v[k], R.t1[k]);
    bfly4<true>(v);
}
// v[a] e^{-i pi a/8} up to one real factor per register, which the rounding FMA applies (round_coef's gain)
__device__ __forceinline__ double untwist256_gain(int a) { return a == 0 ? 1.0 : (a == 2 ? kR : kCos8); }
__device__ __forceinline__ void untwist256(double2 (&v)[4]) {
    v[1] = make_double2(fma(kTan8, v[1].y, v[1].x), fma(-kTan8, v[1].x, v[1].y));   // e^{-i pi/8} / cos
    v[2] = make_double2(v[2].x + v[2].y, v[2].y - v[2].x);                          // e^{-i pi/4} / R
    v[3] = make_double2(fma(kTan8, v[3].x, v[3].y), fma(kTan8, v[3].y, -v[3].x));   // e^{-3 i pi/8} = -i e^{i pi/8}, / cos
}

// ---- key preparation: BK polynomial -> the two half spectra of this kernel ----
// bkw layout: [n][2L rows][c = 2][KA, KB][g = 4][lane = 64] double2, each scaled by 1/256 (the inverse's normalisation):
// KA / KB = half transforms of the even / odd coefficients.  One CMux step's block BK_i is 32 L KiB, contiguous.  61.9 MB at n = 630.
__global__ __launch_bounds__(64) void k_bk_to_spectrum_w12(const Torus32* bk_raw, double2* bkw) {
    const int lane = threadIdx.x;
    const Roots256 R = make_roots256(lane);
    const size_t poly = blockIdx.x;  // (i * 2L + row) * 2 + c
    const Torus32* src = bk_raw + poly * kN;
    double2* dst = bkw + poly * (2 * kQ) + lane;
    constexpr double inv = 1.0 / 256.0;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        double2 v[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int j = 128 * a + 2 * lane + h;
            v[a] = make_double2((double)src[j], (double)src[j + kM]);
        }
        fwd256(v, R);
#pragma unroll
        for (int g = 0; g < 4; g++) dst[(h * 4 + g) * 64] = make_double2(v[g].x * inv, v[g].y * inv);
    }
}

// dynamic LDS: acc [2][1024] int32 | sp [2L rows][2 parities][256] double2 | sbk = BK_i [2L][2][2][256] double2 | bara [i1-i0] u16
// (8 + 16 L + 32 L KiB: 152 KiB for L = 3 -- one workgroup per CU, which is this kernel's regime)
// BK_i goes from L2 into LDS by LDS-DMA (global_load_lds_dwordx4: no registers, 8 requests of 1 KiB per wave, issued at the top of
// the step and landed under the forward phase).  Read into registers by the eight output waves themselves it is 24 requests per
// wave and 192 KiB per step (both output parities need KA and KB): the CU's vector-memory path was busy for ~5000 of a step's
// ~11 600 cycles and the step took 27 % LONGER than wide4's (profiles/r4_wide12.txt).
template <int L, int BGBIT, int GUARD, bool DIAG = false, bool PREFETCH = true>
__global__ __launch_bounds__(256 * L) void k_blind_rotate_wide12(DevKeys K, const double2* __restrict__ bkw,
                                                               const uint16_t* __restrict__ st_bara, int32_t nb,
                                                               int32_t* st_acc, int32_t i0, int32_t i1, Torus32* ext,
                                                               unsigned* guard, unsigned long long* diag) {
    constexpr int NW = 4 * L, NT = 64 * NW;
    constexpr int kStepElems = 2 * L * 2 * 2 * kQ;  // double2 elements of one BK_i
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* acc = reinterpret_cast<int32_t*>(smem);
    double2* sp = reinterpret_cast<double2*>(smem + (size_t)2 * kN * 4);
    double2* sbk = sp + 4 * L * kQ;
    uint32_t* s_sink = reinterpret_cast<uint32_t*>(sbk + kStepElems);  // 64 words nobody reads: where the L2 prefetch's bytes go
    uint16_t* s_bara = reinterpret_cast<uint16_t*>(s_sink + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t item = (int64_t)blockIdx.x;
    int32_t* gacc = st_acc + (size_t)item * 2 * kN;
    const Roots256 R = make_roots256(lane);
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
    // forward role: parity h of digit qw of accumulator polynomial pw (row = pw L + qw of BK_i)
    const int row = wave >> 1, h = wave & 1;
    const int pw = row / L, qw = row - pw * L;
    const int sh = 32 - (qw + 1) * BGBIT;
    double2* spw = sp + (size_t)wave * kQ + lane;  // this wave's published half spectrum [g][lane]
    const unsigned char* accb = reinterpret_cast<const unsigned char*>(acc);
    const uint32_t pb = (uint32_t)pw * (kN * 4);
    const int32_t* accp = acc + pw * kN + h;
    // output role (waves 0..7): parity on0 of output polynomial oc, from the L rows of accumulator polynomial oeta:
    //   even = sum A KA + Y sum B KB,  odd = sum A KB + sum B KA   (A / B: the even / odd half spectrum of a digit row)
    const bool is_out = wave < 8;
    const int oc = wave & 1, on0 = (wave >> 1) & 1, oeta = (wave >> 2) & 1;
    uint32_t* acco = reinterpret_cast<uint32_t*>(acc) + oc * kN + on0;
    const double2* spo = sp + (size_t)(oeta * L) * 2 * kQ + lane;
    const double2* sbko = sbk + (size_t)((oeta * L) * 2 + oc) * 2 * kQ + lane;
    const int arr1 = on0, arr2 = 1 - on0;
    double2 Z[4];  // Y_k = theta W256^k at this lane's four points for the even parity, 1 for the odd one
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int k = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3) + 64 * g;
        double sn, cs;
        sincospi((double)(1 + 4 * k) / 512.0, &sn, &cs);
        Z[g] = on0 ? make_double2(1.0, 0.0) : make_double2(cs, sn);
    }
    // this wave's share of the BK_i copy: 1 KiB pieces wave, wave + NW, ... of its 32 L
    const __attribute__((address_space(3))) unsigned char* sbk3 =
        (const __attribute__((address_space(3))) unsigned char*)(__attribute__((address_space(3))) void*)sbk;
    double dev_max = 0.0;
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    if (DIAG) tlast = stamp();
#define IEACHE_STAMP12(idx)                    \
    if (DIAG) {                                \
        const unsigned long long t_ = stamp(); \
        tsum[idx] += t_ - tlast;               \
        tlast = t_;                            \
    }

#pragma unroll 1
    for (int32_t i = i0; i < i1; i++) {
        const int32_t a = __builtin_amdgcn_readfirstlane((int32_t)s_bara[i - i0]);
        if (a == 0) continue;  // workgroup-uniform
        {
            const double2* __restrict__ bki = bkw + (size_t)i * kStepElems + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int piece = wave + NW * j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bki + piece * 64),
                                                 (__attribute__((address_space(3))) void*)(sbk3 + piece * 1024), 16, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        int32_t lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque: keeps the per-coefficient LDS addresses from being hoisted out of the step loop
        // coefficient j = 128 r + 2 lane + h (real part) and j + 512 (imaginary part) of X^a acc_pw - acc_pw
        const uint32_t jb4 = ((uint32_t)(2 * lane_o + h - a) & (2 * kN - 1)) << 2;
        uint32_t rv0[4], rv1[4], pv0[4], pv1[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t t = jb4 + 512u * r;
            const uint32_t o0 = (t & 4092u) | pb, o1 = o0 ^ 2048u;
            rv0[r] = *reinterpret_cast<const uint32_t*>(accb + o0);
            rv1[r] = *reinterpret_cast<const uint32_t*>(accb + o1);
            pv0[r] = (uint32_t)accp[128 * r + 2 * lane_o];
            pv1[r] = (uint32_t)accp[128 * r + 2 * lane_o + kM];
        }
        __builtin_amdgcn_sched_barrier(0);
        double2 v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t t = jb4 + 512u * r;
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)t, 12, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int32_t)(t + 2048u), 12, 1);
            const uint32_t u0 = (rv0[r] ^ m0) + ((dec_offset - pv0[r]) - m0);
            const uint32_t u1 = (rv1[r] ^ m1) + ((dec_offset - pv1[r]) - m1);
            // digit - halfBg = sign-extended field of (u ^ (halfBg << sh))
            const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)(u0 ^ (halfBg << sh)), sh, BGBIT);
            const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)(u1 ^ (halfBg << sh)), sh, BGBIT);
            v[r] = make_double2((double)e0, (double)e1);
        }
        IEACHE_STAMP12(0)
        fwd256(v, R);
        IEACHE_STAMP12(1)
#pragma unroll
        for (int g = 0; g < 4; g++) spw[g * 64] = v[g];  // publish
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of BK_i have landed
        IEACHE_STAMP12(2)
        __syncthreads();  // A: all 4L half spectra and BK_i are in LDS
        IEACHE_STAMP12(3)
        // BK_{i+1} into L2 a step ahead: one word of each of its 128-byte lines, by the waves that have no output role (every
        // wave when all have one), through LDS-DMA into a sink so that no register waits for the data.  A step's block comes from
        // the Infinity Cache or HBM -- no other CU of the XCD has read it earlier -- and without this its copy above lands ~4000
        // cycles after the request, longer than the forward phase that is meant to cover it.
        if (PREFETCH) {
            constexpr int kLines = kStepElems / 8, PW = NW > 8 ? NW - 8 : NW, PER = kLines / (PW * 64);
            static_assert(PER * PW * 64 == kLines, "lines of a BK block per prefetching lane");
            if (wave >= NW - PW && i + 1 < K.n) {
                const unsigned char* nxt = reinterpret_cast<const unsigned char*>(bkw + (size_t)(i + 1) * kStepElems) +
                                           (size_t)((wave - (NW - PW)) * 64 + lane) * 128;
#pragma unroll
                for (int t = 0; t < PER; t++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nxt + (size_t)t * (PW * 64 * 128)),
                                                     (__attribute__((address_space(3))) void*)s_sink, 4, 0, 0);
            }
        }
        if (is_out) {
            double2 s1[4], s2[4];
#pragma unroll
            for (int q = 0; q < L; q++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const double2 ya = spo[(size_t)(2 * q) * kQ + g * 64], yb = spo[(size_t)(2 * q + 1) * kQ + g * 64];
                    const double2 ka = sbko[(size_t)q * (4 * kQ) + arr1 * kQ + g * 64], kb = sbko[(size_t)q * (4 * kQ) + arr2 * kQ + g * 64];
                    s1[g] = q == 0 ? cmulx<false>(ya, ka)
                                   : make_double2(fma(ya.x, ka.x, fma(-ya.y, ka.y, s1[g].x)), fma(ya.x, ka.y, fma(ya.y, ka.x, s1[g].y)));
                    s2[g] = q == 0 ? cmulx<false>(yb, kb)
                                   : make_double2(fma(yb.x, kb.x, fma(-yb.y, kb.y, s2[g].x)), fma(yb.x, kb.y, fma(yb.y, kb.x, s2[g].y)));
                }
            }
            double2 s[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                s[g] = make_double2(fma(s2[g].x, Z[g].x, fma(-s2[g].y, Z[g].y, s1[g].x)), fma(s2[g].x, Z[g].y, fma(s2[g].y, Z[g].x, s1[g].y)));
            IEACHE_STAMP12(4)
            inv256(s, R);
            untwist256(s);
            IEACHE_STAMP12(5)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool watched = GUARD == 1 || (GUARD == 2 && r == 0);
                const uint32_t d0 = round_coef(s[r].x, untwist256_gain(r), watched, dev_max), d1 = round_coef(s[r].y, untwist256_gain(r), watched, dev_max);
                const int32_t j = 128 * r + 2 * lane;
                atomicAdd(&acco[j], d0);  // ds_add_u32; the other row half of this output adds to the same word
                atomicAdd(&acco[j + kM], d1);
            }
        }
        IEACHE_STAMP12(6)
        __syncthreads();  // C: accumulator complete before the next decomposition; every published spectrum and BK_i consumed
        IEACHE_STAMP12(7)
    }
#undef IEACHE_STAMP12
    if (DIAG && diag && lane == 0 && (wave == 0 || wave == 8)) {
#pragma unroll
        for (int t = 0; t < 8; t++) atomicAdd(&diag[(wave >> 3) * 8 + t], tsum[t]);
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
