// ATTIC -- not part of the default build (make attic compiles it as a syntax / code-generation check only).
// Blind-rotation kernels of rounds 1-3 that were measured and lost to the ones in ../blind_rotate_w64.hip; every one produced
// the same bits.  See README.md in this directory for the br_variant numbers they had and the profile that records each result.
#include "../blind_rotate_w64.h"
#include "../fft512.h"

namespace ieache {
namespace w64 {
using namespace dev;
namespace {
constexpr int kW1Gates = 4;
constexpr float kGuardLimit = 0.0625f;

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
template <int L, int BGBIT, bool GUARD, int XLANE = 1, int EARLYB = 0>
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
            load_bk_block(bA, bk_rsrc, lane16, brow);                  // -> output polynomial 0
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int32_t e0 = __builtin_amdgcn_sbfe((int32_t)v0[r], sh, BGBIT);  // v_bfe_i32
                const int32_t e1 = __builtin_amdgcn_sbfe((int32_t)v1[r], sh, BGBIT);
                x[r] = make_double2((double)e0, (double)e1);  // untwisted: the first radix-8 pass applies e^{i pi r/16} itself
            }
            if (EARLYB == 1) {
                load_bk_block(bB, bk_rsrc, lane16, brow + kRowBytes / 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (EARLYB >= 2) {
                // the second block is requested inside the transform, into the registers its twiddles leave
                // (2: after the first inter-pass twiddles, 3: after the second)
                auto req = [&]() {
                    load_bk_block(bB, bk_rsrc, lane16, brow + kRowBytes / 2);
                };
                fft512_forward<true, XLANE, decltype(req), EARLYB == 3>(x, sT, lane, R, req);
            } else {
                fft512_forward<true, XLANE>(x, sT, lane, R);
            }
            if (EARLYB == 0) {
                load_bk_block(bB, bk_rsrc, lane16, brow + kRowBytes / 2);  // -> output polynomial 1
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
                const double2 z = make_double2(s[c][r].x * untwist_gain(r), s[c][r].y * untwist_gain(r));
                const double t0 = z.x + kMagic52, t1 = z.y + kMagic52;
                if (GUARD) {
                    dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic52)));
                    dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic52)));
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
                x[r] = make_double2((double)e0, (double)e1);  // untwisted: the first radix-8 pass applies e^{i pi r/16} itself
            }
            __builtin_amdgcn_sched_barrier(0);
            auto req = [&]() {  // the partner row's block, requested once the transform's twiddle registers are free
#pragma unroll
                for (int k = 0; k < 8; k++) bC[k] = bpar[k * 64];
            };
            fft512_forward<true, 1, decltype(req), true>(x, sT, lane, R, req);
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
        fft512_inverse<true>(s, sT, lane, R);
        uint32_t* accu = reinterpret_cast<uint32_t*>(accw);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double2 z = make_double2(s[r].x * untwist_gain(r), s[r].y * untwist_gain(r));
            const double t0 = z.x + kMagic52, t1 = z.y + kMagic52;
            if (GUARD) {
                dev_max = fmax(dev_max, fabs(z.x - (t0 - kMagic52)));
                dev_max = fmax(dev_max, fabs(z.y - (t1 - kMagic52)));
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
            x[r] = make_double2((double)e0, (double)e1);  // untwisted: the first radix-8 pass applies e^{i pi r/16} itself
        }
        fft512_forward<true>(x, sT, lane, R);
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
            fft512_inverse<true>(s, sT, lane, R);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool watched = GUARD == 1 || (GUARD == 2 && (r & 3) == 0);
                const double2 z = make_double2(s[r].x * untwist_gain(r), s[r].y * untwist_gain(r));
                const double t0 = z.x + kMagic52, t1 = z.y + kMagic52;
                if (watched) dev_max = fmax(dev_max, fmax(fabs(z.x - (t0 - kMagic52)), fabs(z.y - (t1 - kMagic52))));
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
            x[r] = make_double2((double)e0, (double)e1);  // untwisted: the first radix-8 pass applies e^{i pi r/16} itself
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
            const double2 z0 = make_double2(s0[r].x * untwist_gain(r), s0[r].y * untwist_gain(r));
            const double2 z1 = make_double2(s1[r].x * untwist_gain(r), s1[r].y * untwist_gain(r));
            const double t00 = z0.x + kMagic52, t01 = z0.y + kMagic52, t10 = z1.x + kMagic52, t11 = z1.y + kMagic52;
            if (GUARD) {
                dev_max = fmax(dev_max, fmax(fabs(z0.x - (t00 - kMagic52)), fabs(z0.y - (t01 - kMagic52))));
                dev_max = fmax(dev_max, fmax(fabs(z1.x - (t10 - kMagic52)), fabs(z1.y - (t11 - kMagic52))));
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

// ---- round 4: k_blind_rotate_wide12 (12 waves per gate, even / odd half transforms of four points per lane) ----
#include "wide12.h"

}  // namespace

// code-generation check (make attic): one instantiation of each
extern const void* const attic_kernels[7];
const void* const attic_kernels[7] = {(const void*)k_blind_rotate_w1<3, 7, true>, (const void*)k_blind_rotate_w2s<3, 7, true>,
                                     (const void*)k_blind_rotate_wide4b<3, 7, 2>, (const void*)k_blind_rotate_wide1<3, 7, true>,
                                     (const void*)k_blind_rotate_wide12<3, 7, 2>, (const void*)k_blind_rotate_wide12<2, 10, 2>,
                                     (const void*)k_bk_to_spectrum_w12};

}  // namespace w64
}  // namespace ieache
