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
// (`slice`, 1..64; 0 = default 16 or IEACHE_BR_SLICE).  state: items * state_bytes_per_item() bytes of scratch.
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
// kernel variant: 0 = default (IEACHE_BR_VARIANT overrides): wave-local sync, the forward transforms' lane-high transpose
// cross-lane (v_permlane*_swap / DPP), every other transpose -- and the paired inverse -- through LDS (+0.8 % over all-LDS,
// which is variant 12); 1 = all-LDS with
// s_memtime diagnostics printed to stderr; 2 = LDS transposes with workgroup barriers; 3 = cross-lane
// (DPP / v_permlane*_swap) transposes; 4 = 3 with diagnostics; 5 / 6 = only the lane-high / lane-low transpose cross-lane;
// 10 = forward-transform LDS stores interleaved with the twiddle multiplies that feed them (measured: no gain);
// 11 = with round 1's (unneeded) workgroup barrier at the end of every CMux step.
// All produce identical bits.
int32_t default_variant();
// 7 = 2L waves per gate (k_blind_rotate_wide): lower latency per gate, for launches of few gates;
// takes any slice length up to n (one launch for the whole rotation)
constexpr int32_t kVariantWide = 7;
// 13 = one wave per gate on the one-limb spectrum with the rounding guard (k_blind_rotate_w1; the evaluator's default for
// wide launches), 14 = the same without the guard arithmetic (measurement only).  Bit-identical to the two-limb kernels
// as long as the guard stays silent.
constexpr int32_t kVariantOneLimb = 13;
// 20 = two waves per gate on the one-limb spectrum (k_blind_rotate_w2s; the evaluator's choice for mid-size launches),
// 21 = the same without the guard arithmetic
constexpr int32_t kVariantOneLimbTwoWaves = 20;
// 22 = 2L waves per gate, each wave one whole row of the one-limb spectrum (k_blind_rotate_wide1: the latency kernel's
// one-limb form; any slice length up to n), 23 = the same without the guard arithmetic
constexpr int32_t kVariantWideOneLimb = 22;
// 24 = k_blind_rotate_wide itself on the one-limb spectrum (two output waves instead of four; the evaluator's choice for
// launches of at most one gate per CU)
constexpr int32_t kVariantWideHandoverOneLimb = 24;

}  // namespace w64
}  // namespace ieache
