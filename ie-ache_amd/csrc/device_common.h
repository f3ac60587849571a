// Device-side structures and helpers shared by the evaluator's kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "circuit.h"
#include "params.h"

namespace ieache {
namespace dev {

struct DevKeys {
    int32_t n, N, M, logM, l, Bgbit, kpl, ks_t, ks_basebit, ks_base, stride;
    uint32_t dec_offset;
    const double2* bkf;   // [n][kpl][2][2 limbs][M], bit-reversed spectrum order
    const int32_t* ksk;   // [N][t][base][stride]
    const double2* twist; // exp(i*pi*j/N), j < M
    const double2* wtab;  // exp(-2*pi*i*j/M), j < M/2
};

// Where the gate instances of one launch live.
struct WorkDesc {
    const DevGate* gates;  // circuit mode when non-null
    int32_t g0, ng;
    Torus32* store;
    int32_t n_slots;
    const Torus32* flat_a;  // flat mode: rows [item]
    const Torus32* flat_b;
    const Torus32* flat_c;  // third operand of bootsMUX (flat_type == kFlatMux)
    Torus32* flat_out;
    int32_t flat_type;
    int64_t item0;
};

// flat_type of the two blind rotations of bootsMUX(a,b,c) (boot-gates.cpp): item 2g is
// (0,-1/8) + a + b, item 2g+1 is (0,-1/8) - a + c; neither is key-switched on its own
constexpr int32_t kFlatMux = 16;

struct GateInst {
    const Torus32* a;
    const Torus32* b;
    Torus32* out;
    int32_t sa, sb;  // signed multipliers (0 = operand is the constant, handled via cst)
    uint32_t cst;
};

__device__ __forceinline__ void gate_coeffs(int32_t type, int32_t& k, uint32_t& cst) {
    // boot-gates.cpp: AND (0,-1/8)+ca+cb ; XOR (0,1/4)+2(ca+cb) ; OR (0,1/8)+ca+cb ; NAND (0,1/8)-ca-cb
    switch (type) {
        case GATE_AND: k = 1; cst = 0xE0000000u; break;
        case GATE_XOR: k = 2; cst = 0x40000000u; break;
        case GATE_OR: k = 1; cst = 0x20000000u; break;
        default: k = -1; cst = 0x20000000u; break;  // NAND
    }
}

__device__ __forceinline__ GateInst resolve(const WorkDesc& W, int64_t item, int32_t stride) {
    GateInst g;
    int32_t type, k;
    if (W.gates) {
        const int64_t b = item / W.ng;
        const DevGate d = W.gates[W.g0 + (int32_t)(item % W.ng)];
        Torus32* base = W.store + (size_t)b * W.n_slots * stride;
        type = d.type;
        gate_coeffs(type, k, g.cst);
        g.a = d.a_slot >= 0 ? base + (size_t)d.a_slot * stride : nullptr;
        g.b = d.b_slot >= 0 ? base + (size_t)d.b_slot * stride : nullptr;
        g.out = base + (size_t)d.out_slot * stride;
        g.sa = d.a_neg ? -k : k;
        g.sb = d.b_neg ? -k : k;
        // a constant operand is (0, -1/8): only its b term contributes
        if (!g.a) g.cst += (uint32_t)g.sa * 0xE0000000u;
        if (!g.b) g.cst += (uint32_t)g.sb * 0xE0000000u;
    } else if (W.flat_type == kFlatMux) {
        const int64_t gi = item >> 1;
        const bool second = item & 1;
        g.cst = 0xE0000000u;
        g.a = W.flat_a + (size_t)gi * stride;
        g.b = (second ? W.flat_c : W.flat_b) + (size_t)gi * stride;
        g.out = nullptr;
        g.sa = second ? -1 : 1;
        g.sb = 1;
    } else {
        type = W.flat_type;
        gate_coeffs(type, k, g.cst);
        g.a = W.flat_a + (size_t)item * stride;
        g.b = W.flat_b ? W.flat_b + (size_t)item * stride : nullptr;
        g.out = W.flat_out + (size_t)item * stride;
        g.sa = k;
        g.sb = k;
        if (type < 0) {  // raw bootstrap of the row in flat_a (debug hook)
            g.sa = 1;
            g.sb = 0;
            g.cst = 0;
            g.b = nullptr;
        }
    }
    return g;
}

__device__ __forceinline__ uint32_t combined_coef(const GateInst& g, int32_t i, int32_t n) {
    uint32_t v = 0;
    if (g.a) v += (uint32_t)g.sa * (uint32_t)g.a[i];
    if (g.b) v += (uint32_t)g.sb * (uint32_t)g.b[i];
    if (i == n) v += g.cst;
    return v;
}

// libtfhe modSwitchFromTorus32(phase, 2N) for power-of-two N: (phase + 2^(31-log2(2N))) >> (32-log2(2N))
__device__ __forceinline__ int32_t modswitch2N(uint32_t phase, int32_t log2N2) {
    return (int32_t)((phase + (1u << (31 - log2N2))) >> (32 - log2N2));
}

// coefficient i of X^a * p  (mod X^N+1), a in [0,2N)
__device__ __forceinline__ int32_t rot_coef(const int32_t* p, int32_t i, int32_t a, int32_t N) {
    // branch-free: one load plus a conditional negate (a ?: on two loads compiles to
    // divergent branches with a full LDS wait inside each)
    const int32_t idx = (i - a) & (2 * N - 1);
    const uint32_t v = (uint32_t)p[idx & (N - 1)];
    return (int32_t)((idx & N) ? 0u - v : v);
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmul_conj(double2 a, double2 b) {  // a * conj(b)
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 c) {  // a*b + c
    return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}


}  // namespace dev
}  // namespace ieache
