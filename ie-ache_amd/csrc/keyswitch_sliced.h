// Gate-batched, sliced LWE key switch for large launches (keyswitch_sliced.hip).
#pragma once
#include "device_common.h"

namespace ieache {
namespace kss {

// libtfhe's key-switch decomposition (t = 8, basebit = 2) with rows of at most 256 int4 columns
bool supported(const Params& p);
// largest slice (coefficients i per launch) the digit buffer in LDS allows
int32_t max_slice();
// out rows = key switch of `items` extracted samples (ext rows of N+4 int32).  The walk over the N
// coefficients may be cut into launches of `slice` coefficients each (0 = the whole walk in one launch);
// partial sums are kept in the output rows in between.  gates_per_wg: 4, 8, 16 or 32 (0 = by launch size).
// The key buffer must be readable 16 rows past its end (prefetch).  Returns the number of launches.
int launch(const Params& p, const dev::DevKeys& K, const dev::WorkDesc& W, int64_t items, const Torus32* ext,
           Torus32* flat_out, int32_t slice, int32_t gates_per_wg, hipStream_t stream);

}  // namespace kss
}  // namespace ieache
