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
// (`slice`: 1..64 for the one- and two-limb one-wave-per-gate kernels and k_blind_rotate_w2, up to the whole rotation for k_blind_rotate_w2r / _w4r / _wide*, which
// reload their per-lane rotation amounts every 64 steps or keep them in LDS; out of range or 0 = default 16 or
// IEACHE_BR_SLICE; the evaluator passes 64 / the whole rotation for launches whose gates are all resident at once).
// state: items * state_bytes_per_item() bytes of scratch.
// ext rows of N+4 int32 (may be null), dbg_acc [items][2][N] (may be null; when set, pass ext = null).
// Returns the number of k_blind_rotate_w2 launches issued.
// d_bkf1 / guard: the one-limb spectrum and the two-word guard record of k_blind_rotate_w1 (may be null for the
// two-limb variants).
// Rotation of roles for mid-size launches (blind_rotate_w64.hip: launch_mixed_phases): k <= 4 subsets of the items on k streams
// (streams[0] = the launch's own), tw of them at a time on the two-waves-per-gate kernel for s2 steps while the others take s1
// (<= 64 x any) steps on the one-wave-per-gate kernel; `cycles` rounds of k phases, the rest of the rotation by the ordinary
// slice loop.  ev: k events (no timing).  sync: a barrier across the streams at every phase boundary.
struct MixPlan {
    int k = 0, tw = 0;
    int32_t s1 = 0, s2 = 0, cycles = 0;
    int32_t tail_s1 = 0, tail_s2 = 0;  // a last, shortened round (0 = none)
    int wg = 4;  // gate instances per workgroup of the one-wave kernel's launches (LDS: with two two-wave gates per CU, 4 or 2 x 2 fit)
    bool sync = true;
    hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};
int launch(const Params& p, const dev::DevKeys& K, const double2* d_bkf, const double2* d_bkf1, unsigned* guard,
           const dev::WorkDesc& W, int64_t items, void* state, Torus32* ext, int32_t steps, Torus32* dbg_acc, int32_t slice,
           int32_t variant, const double2* d_twiddles, hipStream_t stream, int wg_gates = 0, const MixPlan* mix = nullptr);
// wg_gates: gate instances per workgroup of the two one-wave-per-gate kernels (k_blind_rotate_w1b, guard on one coefficient
// in four, and k_blind_rotate_x1): 1 .. 4, 0 = 4.  Four gates share a workgroup only for the twiddle table; fewer per
// workgroup let a launch that does not fill the chip spread evenly over the CUs (LDS: 4 -> 2 workgroups per CU, 3 -> 2, 2 -> 3, 1 -> 6).
// one-limb form (k_blind_rotate_w1): raw BK -> spectrum [n][2l][2][8][64] double2
size_t spectrum1_elems(const Params& p);
void prepare_spectrum1(const Params& p, const Torus32* d_bk_raw, double2* d_bkf1, hipStream_t stream);
size_t lds_bytes_w1(int wg_gates = 4);
int gates_per_workgroup_w1();
// the kernels' twiddle table (twiddle_table_elems() double2 in device memory), built once per context
size_t twiddle_table_elems();
void build_twiddle_table(double2* d_tw, hipStream_t stream);
// Kernel variants ("br_variant" / IEACHE_BR_VARIANT; all produce identical bits).  0 lets the EVALUATOR choose by launch size
// (evaluator.hip: <= one gate per CU -> 38, <= 2 per CU -> 43, <= 5 per CU -> 36, above -> 31; "exact_fft": 7 / 0 / 9);
// passed to launch() itself, 0 is the two-limb two-wave kernel.
//   two limbs (exact by construction):
//     9  k_blind_rotate_x1: one wave per gate (round 4; wide launches)
//     0  k_blind_rotate_w2: two waves per gate, split by output polynomial  12  every transpose through LDS (round 1)
//     7  k_blind_rotate_wide: 2L waves per gate (latency; any slice length up to n)   8  with s_memtime phase stamps on stderr
//   one limb with the rounding guard (on one rounded coefficient in four unless noted):
//     31 k_blind_rotate_w1b: one wave per gate (wide launches)   32 guard on every coefficient   35 no guard (measurement)   49 phase stamps
//     36 k_blind_rotate_w2r: two waves per gate, rows split (2 .. 5 gates per CU)      37 guard on every coefficient
//     43 k_blind_rotate_w4r: four waves per gate, rows 2:1:2:1 (1 .. 2 gates per CU)  44 guard on every coefficient
//     38 k_blind_rotate_wide4: 2L waves per gate, four output waves (<= 1 gate per CU) 39 guard on every coefficient
//     24 k_blind_rotate_wide on the one-limb spectrum (round 2's latency kernel, the A/B partner of 38)
// Every other number of rounds 1-3 (k_blind_rotate_w1, _w2s, _wide1, _wide4b and the template flags that lost their A/B) is
// refused; attic/README.md maps them to the profile that records each measurement.
int32_t default_variant();
bool variant_known(int32_t v);
bool variant_one_limb(int32_t v);  // takes the one-limb spectrum and the guard record (the sampled audit applies)
constexpr int32_t kVariantWide = 7;
constexpr int32_t kVariantExactOneWave = 9;         // k_blind_rotate_x1 (round 4): two limbs, one wave per gate
constexpr int32_t kVariantTwoWavesLds = 12;
constexpr int32_t kVariantWideOneLimb = 24;
constexpr int32_t kVariantOneLimbDefault = 31;      // k_blind_rotate_w1b, guard on one coefficient in four (round 3)
constexpr int32_t kVariantOneLimbStamps = 49;
constexpr int32_t kVariantOneLimbTwoWaves = 36;     // k_blind_rotate_w2r (round 3)
constexpr int32_t kVariantOneLimbFourWaves = 43;    // k_blind_rotate_w4r (round 3): launches of one to two gates per CU
constexpr int32_t kVariantWideHandoverOneLimb = 38;  // k_blind_rotate_wide4 (round 3)

}  // namespace w64
}  // namespace ieache
