// Two-waves-per-gate blind rotation specialised for N=1024, k=1 and libtfhe's two default
// gate-bootstrapping sets: l=3, Bgbit=7 (>= v1.1) and l=2, Bgbit=10 (v1.0).  See blind_rotate_w64.hip.
#pragma once
#include "device_common.h"

namespace ieache {
namespace w64 {

bool supported(const Params& p);
// ... and the parameter sets whose sums leave the one-limb transform enough FP64 headroom (l=3, Bgbit=7)
bool one_limb_supported(const Params& p);
// number of double2 elements of the BK spectrum in this kernel's layout
size_t spectrum_elems(const Params& p);
size_t lds_bytes(const Params& p);
int32_t bara_stride(const Params& p);
// raw BK [n][2l][2][N] int32 (device) -> two-limb spectrum [n][2l][4][8][64] double2
void prepare_spectrum(const Params& p, const Torus32* d_bk_raw, double2* d_bkf, hipStream_t stream);
// bytes of blind-rotation state (accumulator + rotation amounts) one gate instance keeps in HBM between slices
size_t state_bytes_per_item(const Params& p);
// K0..K4 for `items` gate instances: prologue, then the CMux steps in slices of S steps per launch
// (`slice`: 1..64 for the one-wave-per-gate kernels, up to the whole rotation for k_blind_rotate_w2r / _w4r / _wide*, which
// reload their per-lane rotation amounts every 64 steps or keep them in LDS; out of range or 0 = default 16 or
// IEACHE_BR_SLICE; the evaluator passes 64 / the whole rotation for launches whose gates are all resident at once).
// state: items * state_bytes_per_item() bytes of scratch.
// ext rows of N+4 int32 (may be null), dbg_acc [items][2][N] (may be null; when set, pass ext = null).
// Returns the number of k_blind_rotate_w2 launches issued.
// d_bkf1 / guard: the one-limb spectrum and the two-word guard record of k_blind_rotate_w1 (may be null for the
// two-limb variants).
int launch(const Params& p, const dev::DevKeys& K, const double2* d_bkf, const double2* d_bkf1, unsigned* guard,
           const dev::WorkDesc& W, int64_t items, void* state, Torus32* ext, int32_t steps, Torus32* dbg_acc, int32_t slice,
           int32_t variant, const double2* d_twiddles, hipStream_t stream);
// one-limb form (k_blind_rotate_w1): raw BK -> spectrum [n][2l][2][8][64] double2
size_t spectrum1_elems(const Params& p);
void prepare_spectrum1(const Params& p, const Torus32* d_bk_raw, double2* d_bkf1, hipStream_t stream);
size_t lds_bytes_w1();
int gates_per_workgroup_w1();
// the kernels' twiddle table (twiddle_table_elems() double2 in device memory), built once per context
size_t twiddle_table_elems();
void build_twiddle_table(double2* d_tw, hipStream_t stream);
// Kernel variants ("br_variant" / IEACHE_BR_VARIANT; all produce identical bits).  0 lets the EVALUATOR choose by launch
// size (evaluator.hip: <= one gate per CU -> 38, <= 2 per CU -> 43, <= 5 per CU -> 36, above -> 31; "exact_fft": 7 / 0); passed to launch()
// itself, 0 is the two-limb two-wave kernel.
//   two limbs (exact by construction), two waves per gate -- k_blind_rotate_w2:
//     0  wave-local sync, the forward transforms' lane-high transpose cross-lane (v_permlane*_swap / DPP), every other
//        transpose and the paired inverse through LDS       12  every transpose through LDS (round 1's default)
//     1  12 with s_memtime phase stamps on stderr            2  LDS transposes with workgroup barriers
//     3  every transpose cross-lane   4  3 with stamps       5 / 6  only the lane-high / lane-low transposes cross-lane
//     10 forward-transform LDS stores interleaved with the twiddle multiplies that feed them (no gain)
//     11 with round 1's (unneeded) workgroup barrier at the end of every CMux step
//   two limbs, 2L waves per gate -- k_blind_rotate_wide (latency; any slice length up to n):   7   (8 with stamps)
//   one limb with the rounding guard, one wave per gate -- k_blind_rotate_w1 (wide launches):
//     13 default   14 without the guard arithmetic   15 / 16 forward transposes both through LDS / both cross-lane
//     17 / 18 / 19 second BK block of a row requested before its transform / after its first / second twiddles
//     30 BK blocks through global_load instead of buffer_load (-2.7 %)
//   one limb, two waves per gate -- k_blind_rotate_w2s (mid-size launches):   20   (21 without the guard arithmetic)
//   one limb, 2L waves per gate (latency):
//     22 / 23 every wave a whole row, no hand-over -- k_blind_rotate_wide1 (measured slower; 23 without guard arithmetic)
//     24 k_blind_rotate_wide on the one-limb spectrum (narrow launches)   25-28 its transposes cross-lane (slower)
//     29 24 with phase stamps
//   round 3 (all on the one-limb spectrum; guard on one rounded coefficient in four unless noted):
//     31 k_blind_rotate_w1b: k_blind_rotate_w1 with the index / sign arithmetic of the decomposition rewritten -- the
//        default of wide launches   32 guard on every coefficient   33 / 34 some forward transposes through LDS   35 no guard
//     36 k_blind_rotate_w2r: two waves per gate, the ROWS of BK_i split between them, one hand-over per step -- the default
//        of mid-size launches   37 guard on every coefficient
//     38 k_blind_rotate_wide4: 2L waves per gate, four output waves on half the rows each, no barrier B -- the default of
//        narrow launches   39 guard on every coefficient
//     43 k_blind_rotate_w4r: four waves per gate, rows split 2 : 1 : 2 : 1, one hand-over per step -- the default of launches
//        of one to two gates per CU (4.4 ms against w2r's 5.2)   44 guard on every coefficient
//     measured and NOT faster (kept as the A/B partners): 41 k_blind_rotate_wide4b (wide4 built for two workgroups per CU:
//     5.7 ms at 512 gates against w2r's 5.2; the same build of k_blind_rotate_wide spilled 186 registers and was dropped),
//     42 k_blind_rotate_w1b with the rows software-pipelined (next row's digits under this row's last transpose: -1 %),
//     45 / 46 k_blind_rotate_w1b with both transposes of three / all six forward transforms cross-lane (two-instruction
//     v_cndmask_b32_dpp exchanges, no LDS round trip: -2 % / -5.5 %), 47 / 48 k_blind_rotate_w1b taking the first / both twiddle
//     sets from the global table through the buffer path instead of LDS (105 of 341 LDS instructions per step: -2 % / -4 %),
//     50 k_blind_rotate_w1b with every wave touching one 8 KiB slice of the NEXT step's BK blocks per step (L2 prefetch: -1.3 %)
//     49 k_blind_rotate_w1b with s_memtime phase stamps on stderr (diagnostic)
int32_t default_variant();
constexpr int32_t kVariantWide = 7;
constexpr int32_t kVariantExactOneWave = 9;  // k_blind_rotate_x1 (round 4): two limbs, one wave per gate
constexpr int32_t kVariantOneLimb = 13;
constexpr int32_t kVariantOneLimbDefault = 31;  // k_blind_rotate_w1b, guard on one coefficient in four (round 3)
constexpr int32_t kVariantOneLimbTwoWaves = 36;     // k_blind_rotate_w2r (round 3; round 2's k_blind_rotate_w2s = 20)
constexpr int32_t kVariantOneLimbFourWaves = 43;    // k_blind_rotate_w4r (round 3): launches of one to two gates per CU
constexpr int32_t kVariantWideOneLimb = 22;
constexpr int32_t kVariantWideHandoverOneLimb = 38;  // k_blind_rotate_wide4 (round 3; round 2's k_blind_rotate_wide on one limb = 24)

}  // namespace w64
}  // namespace ieache
