// Wave-per-gate blind rotation specialised for N=1024, k=1, l=3, Bgbit=7
// (libtfhe's default 128-bit gate-bootstrapping set).  See blind_rotate_w64.hip.
#pragma once
#include "device_common.h"

namespace ieache {
namespace w64 {

bool supported(const Params& p);
// number of double2 elements of the BK spectrum in this kernel's layout
size_t spectrum_elems(const Params& p);
size_t lds_bytes(const Params& p);
// raw BK [n][2l][2][N] int32 (device) -> two-limb spectrum [n][2l][4][8][64] double2
void prepare_spectrum(const Params& p, const Torus32* d_bk_raw, double2* d_bkf, hipStream_t stream);
// K0..K4 for `items` gate instances; ext rows of N+4 int32 (may be null), dbg_acc [items][2][N] (may be null)
void launch(const Params& p, const dev::DevKeys& K, const double2* d_bkf, const dev::WorkDesc& W, int64_t items,
            Torus32* ext, int32_t steps, Torus32* dbg_acc, hipStream_t stream);

}  // namespace w64
}  // namespace ieache
