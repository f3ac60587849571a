// LWE key switch of large launches as an int8 matrix product on the MFMA pipe (keyswitch_mfma.hip).
#pragma once
#include "device_common.h"

namespace ieache {
namespace ksm {

// libtfhe's key-switch decomposition t = 8, basebit = 2: one coefficient = 8 positions x 4 digit values = one K-step of 32
bool supported(const Params& p);
// a K split the product accepts: divides the N / 4 digit groups and leaves each split a whole number of its loop's trips
// (two groups per trip with B fragments eight coefficients ahead): N = 1024 -> 1 .. 128, N = 64 -> 1 .. 8
bool split_ok(const Params& p, int32_t ksplit);
// bytes of the byte-limb form of the key-switch key (built once per key load): N * ceil(stride / 32) * 4096
size_t limb_matrix_bytes(const Params& p);
// bytes of digit scratch for launches of up to `items` gate instances
size_t digit_scratch_bytes(const Params& p, int64_t items);
// padded KSK [N][t][base][stride] int32 (device) -> limb matrix in MFMA operand order
void prepare(const Params& p, const int32_t* d_ksk_padded, int8_t* d_limbs, hipStream_t stream);
// out rows = key switch of `items` extracted samples (ext rows of N+4 int32); bit-identical to the other key-switch
// kernels.  d_digits: digit_scratch_bytes(p, items) bytes of scratch.  ksplit: the walk over the N coefficients cut into
// this many workgroups per (gate block, coefficient block), partial sums meeting through atomic adds (1, 2, 4, 8; 0 = by
// launch size and the device's `cus` compute units).  Returns the number of kernel launches.
int launch(const Params& p, const dev::DevKeys& K, const dev::WorkDesc& W, int64_t items, const Torus32* ext, Torus32* flat_out,
           const int8_t* d_limbs, void* d_digits, int32_t ksplit, int32_t cus, hipStream_t stream);

}  // namespace ksm
}  // namespace ieache
