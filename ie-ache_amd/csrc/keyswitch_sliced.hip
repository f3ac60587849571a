// K5, gate-batched and sliced: LWE key switch for launches of thousands of gate instances,
// libtfhe's t = 8, basebit = 2 decomposition (lweKeySwitchTranslate_fromArray).
//
//   out = (0, b') - sum_{i < N} sum_{j < t} KSK[i][j][digit_j(a'_i)]        (int32, wraparound)
//
// What bounds this at scale is (a) how many instructions it takes one gate to pick "its" row of a
// position and subtract it, and (b) how many KSK bytes each workgroup drags through the caches.
//  (a) A workgroup owns G = 32 gate instances; each lane owns one int4 column of all 32
//      accumulators (128 VGPRs).  The three candidate rows of a position sit in a 16-register
//      table laid out component-major ({0, r1.x, r2.x, r3.x}, {0, r1.y, ...}, ...), so a gate's digit
//      (wave-uniform, in an SGPR) IS the register offset: one s_bfe_u32 + one s_set_gpr_idx_idx +
//      four v_sub_u32 with an indexed source per gate and position.  The compiler's own code for
//      the same thing is 19 instructions (it re-arms the index mode around every element), and it
//      cannot be told to keep a table in fixed registers across statements, so the walk over the
//      positions is one inline-assembly block with hand-assigned registers.
//  (b) The walk is cut into slices of a few dozen coefficients i, one launch per slice, every
//      workgroup of a launch walking the SAME slice: the rows of a slice (a few MB) are then read
//      from HBM / Infinity Cache once per XCD and served to all other workgroups by the L2, exactly
//      like the blind rotation's BK blocks.  Partial sums live in the output rows between launches.
// Subtraction mod 2^32 commutes, so the result is bit-identical to the other key-switch kernels.
#include "keyswitch_sliced.h"

#include <hip/hip_runtime.h>

#include <stdexcept>

namespace ieache {
namespace kss {

using namespace dev;

namespace {

typedef int v32i __attribute__((ext_vector_type(32)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// Register plan of the assembly block for G gate instances per workgroup (nothing else may live
// there while it runs); TB = 32 + 4G, stages S0..S3 = TB + 16 + 12k, DG = S3 + 12:
//   v[32 : 32+4G)   accumulators: gate g, component c at v[32 + 4g + c]
//   v[TB : TB+16)   row table, component-major: v[TB + 4c + d] = component c of row d (d = 0: zero)
//   v[Sk : Sk+12)   rows (r1, r2, r3) of the positions = k mod 4, fetched four positions ahead
//   v[DG : DG+G/2)  packed digits of one coefficient for the G gates, as read from LDS
//   s[80:81]        address of the current prefetch position's row 0;  s82 remaining coefficients
//   s[84 : 84+G/2)  packed digits: s[84 + g/2] holds gate g in bits [16(g&1), +16), digit j at bit 2j
//   s83            scratch;  s79 saved M0
// G = 32 needs 240 registers (2 waves per SIMD); G = 16, 8 and 4 need 168, 132 and 114 (3, 3 and 4 waves
// per SIMD) and are for smaller launches, where more workgroups matter more than fewer row fetches per gate.
#define KS_SUB4(GATE, TB)                                                               \
    "v_sub_u32 v[32+4*(" #GATE ")+0], v[32+4*(" #GATE ")+0], v[" #TB "+0]\n\t"         \
    "v_sub_u32 v[32+4*(" #GATE ")+1], v[32+4*(" #GATE ")+1], v[" #TB "+4]\n\t"         \
    "v_sub_u32 v[32+4*(" #GATE ")+2], v[32+4*(" #GATE ")+2], v[" #TB "+8]\n\t"         \
    "v_sub_u32 v[32+4*(" #GATE ")+3], v[32+4*(" #GATE ")+3], v[" #TB "+12]\n\t"
#define KS_GATE(GATE, J, TB)                                                                      \
    "s_bfe_u32 s83, s[84+((" #GATE ")/2)], (((" #GATE ")&1)*16+2*" #J ")|0x20000\n\t"          \
    "s_set_gpr_idx_idx s83\n\t" KS_SUB4(GATE, TB)
#define KS_GATES8(B, J, TB)                                                                         \
    KS_GATE(B + 0, J, TB) KS_GATE(B + 1, J, TB) KS_GATE(B + 2, J, TB) KS_GATE(B + 3, J, TB)         \
    KS_GATE(B + 4, J, TB) KS_GATE(B + 5, J, TB) KS_GATE(B + 6, J, TB) KS_GATE(B + 7, J, TB)
#define KS_GATES_G4(J, TB) KS_GATE(0, J, TB) KS_GATE(1, J, TB) KS_GATE(2, J, TB) KS_GATE(3, J, TB)
#define KS_GATES_G8(J, TB) KS_GATES8(0, J, TB)
#define KS_GATES_G16(J, TB) KS_GATES8(0, J, TB) KS_GATES8(8, J, TB)
#define KS_GATES_G32(J, TB) KS_GATES8(0, J, TB) KS_GATES8(8, J, TB) KS_GATES8(16, J, TB) KS_GATES8(24, J, TB)
// one position: wait for its rows, move them into the table, refill the stage with position + 4,
// then let every gate subtract the row its digit selects
#define KS_POSITION(J, S, TB, GATES)                                                           \
    "s_waitcnt vmcnt(9)\n\t"                                                                   \
    "v_mov_b32 v[" #TB "+1], v[" #S "+0]\n\t v_mov_b32 v[" #TB "+5], v[" #S "+1]\n\t"        \
    "v_mov_b32 v[" #TB "+9], v[" #S "+2]\n\t v_mov_b32 v[" #TB "+13], v[" #S "+3]\n\t"       \
    "v_mov_b32 v[" #TB "+2], v[" #S "+4]\n\t v_mov_b32 v[" #TB "+6], v[" #S "+5]\n\t"        \
    "v_mov_b32 v[" #TB "+10], v[" #S "+6]\n\t v_mov_b32 v[" #TB "+14], v[" #S "+7]\n\t"      \
    "v_mov_b32 v[" #TB "+3], v[" #S "+8]\n\t v_mov_b32 v[" #TB "+7], v[" #S "+9]\n\t"        \
    "v_mov_b32 v[" #TB "+11], v[" #S "+10]\n\t v_mov_b32 v[" #TB "+15], v[" #S "+11]\n\t"    \
    "global_load_dwordx4 v[" #S "+0:" #S "+3], %[off1], s[80:81]\n\t"                         \
    "global_load_dwordx4 v[" #S "+4:" #S "+7], %[off2], s[80:81]\n\t"                         \
    "global_load_dwordx4 v[" #S "+8:" #S "+11], %[off3], s[80:81]\n\t"                        \
    "s_add_u32 s80, s80, %[step]\n\t s_addc_u32 s81, s81, 0\n\t"                               \
    "s_set_gpr_idx_on s83, gpr_idx(SRC1)\n\t" GATES(J, TB) "s_set_gpr_idx_off\n\t"
#define KS_RFL4(DG, K)                                                                             \
    "v_readfirstlane_b32 s[84+" #K "+0], v[" #DG "+" #K "+0]\n\t v_readfirstlane_b32 s[84+" #K "+1], v[" #DG "+" #K "+1]\n\t" \
    "v_readfirstlane_b32 s[84+" #K "+2], v[" #DG "+" #K "+2]\n\t v_readfirstlane_b32 s[84+" #K "+3], v[" #DG "+" #K "+3]\n\t"
// the whole walk over the coefficients of a slice; DIGITS reads one coefficient's digits into s[84...]
#define KS_LOAD3(S)                                                                             \
    "global_load_dwordx4 v[" #S "+0:" #S "+3], %[off1], s[80:81]\n\t"                         \
    "global_load_dwordx4 v[" #S "+4:" #S "+7], %[off2], s[80:81]\n\t"                         \
    "global_load_dwordx4 v[" #S "+8:" #S "+11], %[off3], s[80:81]\n\t"                        \
    "s_add_u32 s80, s80, %[step]\n\t s_addc_u32 s81, s81, 0\n\t"
#define KS_WALK(TB, S0, S1, S2, S3, GATES, DIGITS, LDSTEP)                                      \
    "s_mov_b32 s79, m0\n\t" /* the index mode below writes M0 */                                \
    "s_mov_b64 s[80:81], %[rb]\n\t"                                                             \
    "s_mov_b32 s82, %[ni]\n\t"                                                                  \
    "v_mov_b32 v[" #TB "+0], 0\n\t v_mov_b32 v[" #TB "+4], 0\n\t v_mov_b32 v[" #TB "+8], 0\n\t v_mov_b32 v[" #TB "+12], 0\n\t" \
    KS_LOAD3(S0) KS_LOAD3(S1) KS_LOAD3(S2) KS_LOAD3(S3)                                          \
    "1:\n\t" DIGITS                                                                              \
    "v_add_u32 %[lds], " #LDSTEP ", %[lds]\n\t"                                                  \
    "s_nop 3\n\t" /* VALU wrote the SGPRs the s_bfe_u32 below reads */                           \
    KS_POSITION(0, S0, TB, GATES) KS_POSITION(1, S1, TB, GATES) KS_POSITION(2, S2, TB, GATES) KS_POSITION(3, S3, TB, GATES) \
    KS_POSITION(4, S0, TB, GATES) KS_POSITION(5, S1, TB, GATES) KS_POSITION(6, S2, TB, GATES) KS_POSITION(7, S3, TB, GATES) \
    "s_sub_u32 s82, s82, 1\n\t"                                                                  \
    "s_cmp_lg_u32 s82, 0\n\t"                                                                    \
    "s_cbranch_scc1 1b\n\t"                                                                      \
    "s_waitcnt vmcnt(0)\n\t" /* the four positions fetched past the slice (the key buffer is padded for them) */ \
    "s_mov_b32 m0, s79\n\t"
#define KS_DIGITS_G4(DG)  "ds_read_b64 v[" #DG ":" #DG "+1], %[lds]\n\t s_waitcnt lgkmcnt(0)\n\t"                         \
                          "v_readfirstlane_b32 s84, v[" #DG "+0]\n\t v_readfirstlane_b32 s85, v[" #DG "+1]\n\t"
#define KS_DIGITS_G8(DG)  "ds_read_b128 v[" #DG ":" #DG "+3], %[lds]\n\t s_waitcnt lgkmcnt(0)\n\t" KS_RFL4(DG, 0)
#define KS_DIGITS_G16(DG) "ds_read_b128 v[" #DG ":" #DG "+3], %[lds]\n\t ds_read_b128 v[" #DG "+4:" #DG "+7], %[lds] offset:16\n\t" \
                          "s_waitcnt lgkmcnt(0)\n\t" KS_RFL4(DG, 0) KS_RFL4(DG, 4)
#define KS_DIGITS_G32(DG) "ds_read_b128 v[" #DG ":" #DG "+3], %[lds]\n\t ds_read_b128 v[" #DG "+4:" #DG "+7], %[lds] offset:16\n\t" \
                          "ds_read_b128 v[" #DG "+8:" #DG "+11], %[lds] offset:32\n\t ds_read_b128 v[" #DG "+12:" #DG "+15], %[lds] offset:48\n\t" \
                          "s_waitcnt lgkmcnt(0)\n\t" KS_RFL4(DG, 0) KS_RFL4(DG, 4) KS_RFL4(DG, 8) KS_RFL4(DG, 12)
#define KS_SGPR_CLOBBERS "memory", "scc", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", \
                         "s94", "s95", "s96", "s97", "s98", "s99"

// LDS: dw [i1 - i0][G] u16 (packed digits)
template <int G>
__global__ __launch_bounds__(256) void k_keyswitch_sliced(DevKeys K, WorkDesc W, const Torus32* ext, Torus32* flat_out,
                                                          int64_t items, int32_t i0, int32_t i1) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int32_t N = K.N, n = K.n, stride = K.stride;
    uint16_t* dw = reinterpret_cast<uint16_t*>(smem);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int64_t item0 = (int64_t)blockIdx.x * G;
    const int32_t gcount = (int32_t)(items - item0 < G ? items - item0 : G);
    const int32_t ni = i1 - i0;
    constexpr uint32_t prec_offset = 1u << (32 - (1 + 16));
    // digits of a'_i, i in [i0, i1), of every gate: digit j at bits [2j, 2j + 2)
    for (int32_t idx = tid; idx < ni * G; idx += nthreads) {
        const int32_t g = idx % G, ii = idx / G;
        uint32_t packed = 0;
        if (g < gcount) {
            const uint32_t a = (uint32_t)ext[(size_t)(item0 + g) * (N + 4) + i0 + ii] + prec_offset;
#pragma unroll
            for (int32_t j = 0; j < 8; j++) packed |= ((a >> (30 - 2 * j)) & 3u) << (2 * j);
        }
        dw[ii * G + g] = (uint16_t)packed;
    }
    const int32_t nvec = stride >> 2;
    const int32_t col = tid;  // one int4 column per thread
    const bool active = col < nvec;
    const int32_t ccol = active ? col : 0;  // idle lanes shadow column 0
    // accumulators: the first slice starts from (0, b'), later slices continue from the output rows
    v32i acc[(G + 7) / 8];
#pragma unroll
    for (int q = 0; q < (G + 7) / 8; q++)
#pragma unroll
        for (int e = 0; e < 32; e++) acc[q][e] = 0;
    auto out_row = [&](int g) -> Torus32* {
        const int64_t it = item0 + g;
        return flat_out ? flat_out + (size_t)it * stride : resolve(W, W.item0 + it, stride).out;
    };
    if (i0 == 0) {
        if (col == (n >> 2)) {  // the column holding b'
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g < gcount) {
                    const int32_t bp = ext[(size_t)(item0 + g) * (N + 4) + N];
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if ((n & 3) == c) acc[g >> 3][4 * (g & 7) + c] = bp;
                }
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g < gcount) {
                const int4 v = reinterpret_cast<const int4*>(out_row(g))[ccol];
                acc[g >> 3][4 * (g & 7) + 0] = v.x;
                acc[g >> 3][4 * (g & 7) + 1] = v.y;
                acc[g >> 3][4 * (g & 7) + 2] = v.z;
                acc[g >> 3][4 * (g & 7) + 3] = v.w;
            }
        }
    }
    __syncthreads();

    // rows [pos][d], pos = i * 8 + j, are consecutive: row d of position pos starts at (pos * 4 + d) * stride
    const Torus32* rowbase = K.ksk + (size_t)i0 * 8 * 4 * stride;
    const uint32_t rowbytes = (uint32_t)stride * 4u;
    uint32_t off1 = (uint32_t)ccol * 16u + rowbytes, off2 = off1 + rowbytes, off3 = off2 + rowbytes;
    const uint32_t step = 4u * rowbytes;
    uint32_t lds_addr = (uint32_t)(uintptr_t)dw;  // LDS byte address (the low 32 bits of a __shared__ pointer)
    // scratch register ranges are claimed through pinned dummy outputs
#define KS_IO [lds] "+v"(lds_addr) : [rb] "s"(rowbase), [ni] "s"(ni), [off1] "v"(off1), [off2] "v"(off2), [off3] "v"(off3), [step] "s"(step) : KS_SGPR_CLOBBERS
    if constexpr (G == 32) {
        v32i t0, t1;
        v16i t2;
        asm volatile(KS_WALK(160, 176, 188, 200, 212, KS_GATES_G32, KS_DIGITS_G32(224), 64)
                     : "+{v[32:63]}"(acc[0]), "+{v[64:95]}"(acc[1]), "+{v[96:127]}"(acc[2]), "+{v[128:159]}"(acc[3]),
                       "=&{v[160:191]}"(t0), "=&{v[192:223]}"(t1), "=&{v[224:239]}"(t2), KS_IO);
    } else if constexpr (G == 16) {
        v32i t0, t1;
        v8i t2;
        asm volatile(KS_WALK(96, 112, 124, 136, 148, KS_GATES_G16, KS_DIGITS_G16(160), 32)
                     : "+{v[32:63]}"(acc[0]), "+{v[64:95]}"(acc[1]), "=&{v[96:127]}"(t0), "=&{v[128:159]}"(t1), "=&{v[160:167]}"(t2), KS_IO);
    } else if constexpr (G == 8) {
        v32i t0, t1;
        v4i t2;
        asm volatile(KS_WALK(64, 80, 92, 104, 116, KS_GATES_G8, KS_DIGITS_G8(128), 16)
                     : "+{v[32:63]}"(acc[0]), "=&{v[64:95]}"(t0), "=&{v[96:127]}"(t1), "=&{v[128:131]}"(t2), KS_IO);
    } else {
        static_assert(G == 4, "G is 4, 8, 16 or 32");
        // four gates = 16 accumulator registers (v[32:47]); the row table gets its own operand (v[48:63])
        // instead of living in the unused half of a 32-register accumulator vector
        v16i a4, tb;
        v32i t0;
        v16i t1;
        v2i t2;
#pragma unroll
        for (int e = 0; e < 16; e++) a4[e] = acc[0][e];
        asm volatile(KS_WALK(48, 64, 76, 88, 100, KS_GATES_G4, KS_DIGITS_G4(112), 8)
                     : "+{v[32:47]}"(a4), "=&{v[48:63]}"(tb), "=&{v[64:95]}"(t0), "=&{v[96:111]}"(t1), "=&{v[112:113]}"(t2), KS_IO);
#pragma unroll
        for (int e = 0; e < 16; e++) acc[0][e] = a4[e];
    }
#undef KS_IO
    if (active) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g < gcount)
                reinterpret_cast<int4*>(out_row(g))[col] = make_int4(acc[g >> 3][4 * (g & 7) + 0], acc[g >> 3][4 * (g & 7) + 1],
                                                                  acc[g >> 3][4 * (g & 7) + 2], acc[g >> 3][4 * (g & 7) + 3]);
        }
    }
}

}  // namespace

bool supported(const Params& p) {
    // one int4 column per lane of at most 4 waves; libtfhe's key-switch decomposition
    return p.ks_t == 8 && p.ks_basebit == 2 && p.lwe_stride() / 4 <= 256 && p.N % 8 == 0;
}

int32_t max_slice() { return 1024; }  // digits of a slice in LDS: slice * G * 2 bytes <= 64 KiB

template <int G>
static void launch_g(const DevKeys& K, const WorkDesc& W, int64_t items, const Torus32* ext, Torus32* flat_out, int32_t i0, int32_t i1,
                     int nld, hipStream_t stream) {
    static const bool attr = [] {
        return hipFuncSetAttribute((const void*)k_keyswitch_sliced<G>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) ==
               hipSuccess;
    }();
    if (!attr) throw std::runtime_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for k_keyswitch_sliced");
    hipLaunchKernelGGL(k_keyswitch_sliced<G>, dim3((unsigned)((items + G - 1) / G)), dim3(64 * nld), (size_t)(i1 - i0) * G * 2, stream,
                       K, W, ext, flat_out, items, i0, i1);
}

int launch(const Params& p, const DevKeys& K, const WorkDesc& W, int64_t items, const Torus32* ext, Torus32* flat_out,
           int32_t slice, int32_t gates_per_wg, hipStream_t stream) {
    const int nld = (p.lwe_stride() / 4 + 63) / 64;
    const int32_t nco = p.N * p.k;
    if (slice < 1 || slice > max_slice()) slice = max_slice();
    if (slice > nco) slice = nco;
    // gates per workgroup: fewer row fetches per gate with 32, more workgroups in flight with 16 / 8
    int g = gates_per_wg;
    if (g != 4 && g != 8 && g != 16 && g != 32)  // measured (profiles/r1_v8_kernel_microbench.txt): 1024 -> 4, 2048-4096 -> 8, 8192 -> 16
        g = items >= 14336 ? 32 : (items >= 5120 ? 16 : (items >= 1536 ? 8 : 4));
    int launches = 0;
    for (int32_t i0 = 0; i0 < nco; i0 += slice) {
        const int32_t i1 = i0 + slice < nco ? i0 + slice : nco;
        if (g == 32)
            launch_g<32>(K, W, items, ext, flat_out, i0, i1, nld, stream);
        else if (g == 16)
            launch_g<16>(K, W, items, ext, flat_out, i0, i1, nld, stream);
        else if (g == 8)
            launch_g<8>(K, W, items, ext, flat_out, i0, i1, nld, stream);
        else
            launch_g<4>(K, W, items, ext, flat_out, i0, i1, nld, stream);
        launches++;
    }
    return launches;
}

}  // namespace kss
}  // namespace ieache
