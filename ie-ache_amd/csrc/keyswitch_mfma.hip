// K5 for large launches as an int8 matrix product on the MFMA pipe.
//
//   out = (0, b') - sum_{i < N} sum_{j < t} KSK[i][j][digit_j(a'_i)]        (int32, wraparound; lweKeySwitch)
//
// The walk kernels (keyswitch_sliced.hip) pay ~6 vector / scalar instructions per gate and position to pick "its" row and
// subtract it, and are bound by instruction issue.  Written as a product it is the one place on this
// path where a matrix unit fits:
//
//   C[g][c] = sum_k A[g][k] * B[k][c],   k = (i, j, d),  A[g][(i,j,d)] = [digit_j(a'_i of gate g) == d]   (one-hot, int8 0/1)
//                                                        B[(i,j,d)][(c, limb)] = byte `limb` of KSK[i][j][d][c]  (int8)
//
// with the 32-bit key words cut into four BALANCED byte limbs (v = b0 + 2^8 b1 + 2^16 b2 + 2^24 b3 mod 2^32, each in
// [-128, 127]).  Every limb column accumulates at most N t = 8 192 products of magnitude <= 128 -- |sum| <= 2^20, exact in
// the MFMA's int32 accumulators -- and out = (0, b') - sum_limb (C_limb << 8 limb) mod 2^32 is the integer definition of the
// key switch: bit-identical to the walk kernels, whatever the order of summation.  The d = 0 rows of a key-switch key are
// zero, so a position whose digit is 0 needs no special case: its one-hot byte multiplies zeros.
//
// One K-step of the 32x32x32 int8 MFMA is exactly one coefficient i: 8 positions j x 4 digit values d.  A wave owns 128 gate
// instances (four 32-row tiles) x 32 output coefficients x 4 limbs = 16 accumulator tiles (256 registers) and walks K alone;
// nothing crosses waves.  Per K-step it reads 4 KiB of B -- stored at key load in exactly the operand order of the
// instruction, so each load is one coalesced 1 KiB global_load_dwordx4 per wave (all waves of the chip that work on the same
// coefficient block read the same stream while it is L2-hot) -- and builds its four A operands from a 256-entry table in
// LDS: the byte holding four 2-bit digits of a'_i -> four one-hot dwords.  The digits come from a small transposing
// pre-pass ([coefficient][gate] order, so that the 32 gates of a tile load as one row).
// Which lane / byte of an operand register is "k" does not matter as long as A and B agree: both use
// k = 16 (lane / 32) + byte, i.e. positions j = 4 (lane / 32) + (byte / 4), digit value d = byte % 4.
#include "keyswitch_mfma.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <stdexcept>

namespace ieache {
namespace ksm {

using namespace dev;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// coefficients of B fragments in flight per wave: 8 (round 4; 4 until then: 0.658 -> 0.610 ms per 8 192 gates, profiles/r4_keyswitch_ahead.txt).
// 4 or 8: the walk of a K split is a multiple of 8 coefficients for every supported N (N % 64 == 0, splits <= 8).
#ifndef IEACHE_KS_AHEAD
#define IEACHE_KS_AHEAD 8
#endif
constexpr int kWaveGates = 128;   // four 32-row tiles
constexpr int kWgWaves = 4;
constexpr int kWgGates = kWaveGates * kWgWaves;

// ---- key preparation: padded KSK [N][8][4][stride] int32 -> B fragments [N][ncb][4 limbs][64 lanes][16 bytes] ----
__global__ __launch_bounds__(256) void k_ksm_prepare(const int32_t* __restrict__ ksk, int8_t* __restrict__ limbs, int32_t stride, int32_t ncb) {
    const int32_t i = blockIdx.x, cb = blockIdx.y;
    int8_t* dst = limbs + ((size_t)i * ncb + cb) * 4096;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {  // (lane, byte) of one fragment; the four limbs of a key word together
        const int lane = idx >> 4, e = idx & 15;
        const int c = cb * 32 + (lane & 31), j = 4 * (lane >> 5) + (e >> 2), d = e & 3;
        // d = 0 rows are never read by lweKeySwitch ("if the digit is not 0: subtract"); here a zero digit's one-hot byte
        // multiplies them, so they are zero by construction whatever the key file held there
        int64_t r = (c < stride && d != 0) ? (int64_t)ksk[(((size_t)i * 8 + j) * 4 + d) * stride + c] : 0;
#pragma unroll
        for (int limb = 0; limb < 4; limb++) {
            const int8_t b = (int8_t)(r & 0xFF);  // balanced: the remainder in [-128, 127]
            dst[limb * 1024 + idx] = b;
            r = (r - b) >> 8;                     // exact division
        }
    }
}

// ---- per launch: digits of a'_i = u.a_i + 2^15, transposed to [i / 4][gate] (four coefficients x 16 bits per word) ----
__global__ __launch_bounds__(256) void k_ksm_digits(const Torus32* __restrict__ ext, unsigned long long* __restrict__ dig4, int64_t items,
                                                    int64_t gpad, int32_t N) {
    __shared__ uint16_t tile[64][66];
    const int64_t g0 = (int64_t)blockIdx.y * 64;
    const int32_t i0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int gy = ty + 4 * r;
        const int64_t g = g0 + gy;
        tile[gy][tx] = g < items ? (uint16_t)(((uint32_t)ext[(size_t)g * (N + 4) + i0 + tx] + 0x8000u) >> 16) : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int q = ty + 4 * r;  // local i / 4
        const unsigned long long v = (unsigned long long)tile[tx][4 * q] | ((unsigned long long)tile[tx][4 * q + 1] << 16) |
                                     ((unsigned long long)tile[tx][4 * q + 2] << 32) | ((unsigned long long)tile[tx][4 * q + 3] << 48);
        dig4[(size_t)(i0 / 4 + q) * gpad + g0 + tx] = v;
    }
}

// ---- per launch: output rows start as (0, ..., 0, b'); the row addresses are kept for the product's epilogue ----
__global__ __launch_bounds__(256) void k_ksm_init(DevKeys K, WorkDesc W, const Torus32* __restrict__ ext, Torus32* flat_out,
                                                  Torus32** out_ptr) {
    const int64_t item = (int64_t)blockIdx.x;
    const int32_t n = K.n, stride = K.stride;
    Torus32* out = flat_out ? flat_out + (size_t)item * stride : resolve(W, W.item0 + item, stride).out;
    if (threadIdx.x == 0) out_ptr[item] = out;
    const Torus32 b = ext[(size_t)item * (K.N + 4) + K.N];
    for (int32_t q = threadIdx.x; q < stride; q += 256) out[q] = q == n ? b : 0;
}

// ---- the product ----
// grid: one workgroup (4 independent waves) per (coefficient block, K split, block of 512 gates); the order is an L2 matter, see below.
__global__ __launch_bounds__(64 * kWgWaves) void k_ksm_gemm(const v4i* __restrict__ limbs, const unsigned long long* __restrict__ dig4,
                                                            Torus32* const* __restrict__ out_ptr, int64_t items, int64_t gpad, int64_t gblocks,
                                                            int32_t N, int32_t ncb, int32_t stride, int32_t ksplit, int32_t xcd_map) {
    __shared__ v4i lut[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // byte = four 2-bit digits, the first in the top bits -> one dword per digit with a 1 in byte `digit`
        v4i e;
#pragma unroll
        for (int q = 0; q < 4; q++) e[q] = 1 << (8 * ((tid >> (6 - 2 * q)) & 3));
        lut[tid] = e;
    }
    __syncthreads();
    // Which (coefficient block, K split) stream and which block of 512 gates this workgroup takes.  Default: streams
    // fastest, so that every XCD (workgroups are dealt round-robin over them) walks all streams and each pulls the whole
    // 84 MB through its L2 (measured: L2 hit rate 59 %, 0.55 GB of fetch per 8 192 gates).  xcd_map (IEACHE_KS_XCD=1) gives
    // each XCD its own eighth of the streams instead, walked by all its CUs together -- fewer bytes, but MEASURED SLOWER
    // (0.77 against 0.66 ms per 8 192 gates: thirty-two CUs hammering the same few L2 channels); kept as the A/B partner.
    const int32_t streams = ncb * ksplit;
    int32_t st;
    int64_t gblk;
    if (xcd_map && streams % 8 == 0) {
        const int32_t x = blockIdx.x & 7, j = blockIdx.x >> 3;
        st = x + 8 * (int32_t)(j / gblocks);
        gblk = j % gblocks;
    } else {
        st = blockIdx.x % streams;
        gblk = blockIdx.x / streams;
    }
    const int32_t cb = st % ncb, ks = st / ncb;
    const int64_t gbase = gblk * kWgGates + (int64_t)wave * kWaveGates;
    if (gbase >= items) return;  // whole wave past the launch (after the only barrier)
    const int m = lane & 31, grp = lane >> 5;
    const int32_t i4_0 = (N / 4 / ksplit) * ks, i4_1 = i4_0 + N / 4 / ksplit;

    v16i acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int l = 0; l < 4; l++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[t][l][v] = 0;

    const unsigned long long* dg = dig4 + gbase + m;
    const int sh0 = 8 * (1 - grp);  // this lane group's byte of a coefficient's 16 digit bits: j = 0..3 the top byte, j = 4..7 the low one
    unsigned long long d64[4], d64n[4];
#pragma unroll
    for (int t = 0; t < 4; t++) d64n[t] = dg[(size_t)i4_0 * gpad + t * 32];
    // B fragments are requested kAhead coefficients ahead (one wave per SIMD has nothing else to hide an L2 miss behind):
    // buffer c holds coefficient 4 i4 + c (kAhead = 8: of two consecutive i4) and is refilled with the coefficient kAhead further
    // on right after its products are issued
    constexpr int kAhead = IEACHE_KS_AHEAD;   // 4 or 8
    constexpr int kGroups = kAhead / 4;       // i4 values per trip of the loop: every K split must hold a whole number of trips (split_ok)
    static_assert(kAhead == 4 || kAhead == 8, "IEACHE_KS_AHEAD: B fragments are requested 4 or 8 coefficients ahead");
    v4i bq[kAhead][4];
    const v4i* bp0 = limbs + (size_t)cb * 256 + lane;
#pragma unroll
    for (int c = 0; c < kAhead; c++)
#pragma unroll
        for (int l = 0; l < 4; l++) bq[c][l] = bp0[(size_t)(4 * i4_0 + c) * ncb * 256 + l * 64];
#pragma unroll 1
    for (int32_t i4 = i4_0; i4 < i4_1; i4 += kGroups) {
#pragma unroll
        for (int gq = 0; gq < kGroups; gq++) {
#pragma unroll
            for (int t = 0; t < 4; t++) d64[t] = d64n[t];
            const bool more_d = i4 + gq + 1 < i4_1;
            if (more_d) {
#pragma unroll
                for (int t = 0; t < 4; t++) d64n[t] = dg[(size_t)(i4 + gq + 1) * gpad + t * 32];
            }
            const bool more = i4 + gq + kGroups < i4_1;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                v4i a[4];
#pragma unroll
                for (int t = 0; t < 4; t++) a[t] = lut[(unsigned)(d64[t] >> (16 * c + sh0)) & 0xFFu];
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int l = 0; l < 4; l++) acc[t][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t], bq[4 * gq + c][l], acc[t][l], 0, 0, 0);
                if (more) {
#pragma unroll
                    for (int l = 0; l < 4; l++) bq[4 * gq + c][l] = bp0[(size_t)(4 * (i4 + gq + kGroups) + c) * ncb * 256 + l * 64];
                }
            }
        }
    }
    // epilogue: D[row][col] of a 32x32 tile sits in register v of lane: col = lane % 32, row = 8 (v / 4) + 4 (lane / 32) + v % 4
    const int32_t coef = cb * 32 + m;
    if (coef >= stride) return;
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int64_t g = gbase + t * 32 + 8 * (v >> 2) + 4 * grp + (v & 3);
            if (g < items) {
                const uint32_t s = (uint32_t)acc[t][0][v] + ((uint32_t)acc[t][1][v] << 8) + ((uint32_t)acc[t][2][v] << 16) + ((uint32_t)acc[t][3][v] << 24);
                if (s) atomicAdd(reinterpret_cast<unsigned*>(out_ptr[g]) + coef, 0u - s);
            }
        }
    }
}

}  // namespace

// A K split is usable when it divides the N / 4 digit groups and leaves every split a whole number (>= 1) of the loop's trips
// of IEACHE_KS_AHEAD / 4 groups: the B fragments are preloaded a whole trip ahead, so a split shorter than a trip would multiply
// the NEXT split's fragments by stale digits and preload past the end of the limb table.
bool split_ok(const Params& p, int32_t ksplit) {
    constexpr int32_t groups_per_trip = IEACHE_KS_AHEAD / 4;
    if (ksplit < 1 || (p.N / 4) % ksplit != 0) return false;
    const int32_t per_split = p.N / 4 / ksplit;
    return per_split >= groups_per_trip && per_split % groups_per_trip == 0;
}

bool supported(const Params& p) { return p.ks_t == 8 && p.ks_basebit == 2 && p.k == 1 && p.N % 64 == 0 && p.N >= 64; }

static int32_t coef_blocks(const Params& p) { return (p.lwe_stride() + 31) / 32; }
static int64_t padded_items(int64_t items) { return (items + kWgGates - 1) / kWgGates * kWgGates; }

size_t limb_matrix_bytes(const Params& p) { return (size_t)p.N * coef_blocks(p) * 4096; }

size_t digit_scratch_bytes(const Params& p, int64_t items) {
    const int64_t gpad = padded_items(items);
    return (size_t)(p.N / 4) * gpad * 8 + (size_t)gpad * sizeof(Torus32*);
}

void prepare(const Params& p, const int32_t* d_ksk_padded, int8_t* d_limbs, hipStream_t stream) {
    hipLaunchKernelGGL(k_ksm_prepare, dim3((unsigned)p.N, (unsigned)coef_blocks(p)), dim3(256), 0, stream, d_ksk_padded, d_limbs,
                       p.lwe_stride(), coef_blocks(p));
}

int launch(const Params& p, const DevKeys& K, const WorkDesc& W, int64_t items, const Torus32* ext, Torus32* flat_out,
           const int8_t* d_limbs, void* d_digits, int32_t ksplit, int32_t cus, hipStream_t stream) {
    if (items <= 0) return 0;
    if (cus <= 0) cus = 256;
    const int64_t gpad = padded_items(items);
    const int32_t ncb = coef_blocks(p);
    unsigned long long* dig4 = reinterpret_cast<unsigned long long*>(d_digits);
    Torus32** out_ptr = reinterpret_cast<Torus32**>(dig4 + (size_t)(p.N / 4) * gpad);
    const int64_t gblocks = gpad / kWgGates;
    if (ksplit <= 0) {
        // One workgroup (4 waves, all 512 registers each) per CU at a time: W = gblocks * ncb workgroups take ceil(W k / CUs)
        // rounds of 1 / k of the walk each, plus a fixed cost per split (table build, one more pass of atomic adds).
        // Measured at n = 630 (profiles/r3_keyswitch_mfma.txt): 512 gates k = 8, 1 024 k = 4, 2 304 k = 2, 8 192 k = 4, 16 384 k = 2.
        const int64_t W0 = gblocks * ncb;
        double best = 0;
        for (int32_t k = 1; k <= 8; k *= 2) {
            if (!split_ok(p, k)) break;
            const double cost = (double)((W0 * k + cus - 1) / cus) / k + 0.02 * k;
            if (ksplit <= 0 || cost < best) {
                best = cost;
                ksplit = k;
            }
        }
    }
    if (!split_ok(p, ksplit)) throw std::invalid_argument("key-switch K split does not leave every split a whole trip of the product's loop");
    static const int32_t xcd_map = getenv("IEACHE_KS_XCD") ? atoi(getenv("IEACHE_KS_XCD")) : 0;  // measurement aid, see k_ksm_gemm
    hipLaunchKernelGGL(k_ksm_digits, dim3((unsigned)(p.N / 64), (unsigned)(gpad / 64)), dim3(256), 0, stream, ext, dig4, items, gpad, p.N);
    hipLaunchKernelGGL(k_ksm_init, dim3((unsigned)items), dim3(256), 0, stream, K, W, ext, flat_out, out_ptr);
    hipLaunchKernelGGL(k_ksm_gemm, dim3((unsigned)(gblocks * ncb * ksplit)), dim3(64 * kWgWaves), 0, stream,
                       reinterpret_cast<const v4i*>(d_limbs), dig4, out_ptr, items, gpad, gblocks, p.N, ncb, p.lwe_stride(), ksplit, xcd_map);
    return 3;
}

}  // namespace ksm
}  // namespace ieache
