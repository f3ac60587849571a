// Parameter set and raw key containers of the evaluator.
//
// Mirrors what the reference obtains from libtfhe's
// TFheGateBootstrappingParameterSet (Keygen/keygen.c:22-23,
// Cloud/cloud.c:666-669).  Values always come from the key header; the
// defaults below are libtfhe >= 1.1 "128-bit" (SURVEY.md App. A).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ieache {

using Torus32 = int32_t;

struct Params {
    int32_t n = 630;          // LWE dimension
    int32_t N = 1024;         // TLWE ring degree
    int32_t k = 1;            // TLWE mask polynomials
    int32_t l = 3;            // TGSW decomposition length
    int32_t Bgbit = 7;        // TGSW decomposition base bits
    int32_t ks_t = 8;         // key-switch length
    int32_t ks_basebit = 2;   // key-switch base bits
    double lwe_alpha_min = 3.0517578125e-05;  // 2^-15, fresh LWE / KS noise
    double lwe_alpha_max = 0.012467;
    double tlwe_alpha_min = 2.98023223876953125e-08;  // 2^-25, BK noise
    double tlwe_alpha_max = 0.012467;

    int32_t kpl() const { return (k + 1) * l; }
    int32_t ks_base() const { return 1 << ks_basebit; }
    int32_t lwe_stride() const { return (n + 1 + 3) & ~3; }  // row stride in int32, 16-B aligned
    size_t bk_count() const { return (size_t)n * kpl() * (k + 1) * N; }
    size_t ksk_count() const { return (size_t)k * N * ks_t * ks_base() * (n + 1); }
    bool supported() const {
        return k == 1 && N >= 16 && N <= 1024 && (N & (N - 1)) == 0 && n >= 1 && l >= 1 &&
               l * Bgbit <= 32 && Bgbit >= 1 && ks_t >= 1 && ks_basebit >= 1 &&
               ks_t * ks_basebit < 32 && ks_basebit <= 4;
    }
};

// 1/8, the gate message amplitude (libtfhe modSwitchToTorus32(1,8))
constexpr Torus32 kMU = 0x20000000;

// Cloud (evaluation) key as raw arrays, libtfhe order:
//   bk  [n][(k+1)l][k+1][N]   TGSW rows
//   ksk [kN][t][base][n+1]    LWE samples, (a.., b)
struct CloudKeyData {
    Params p;
    std::vector<Torus32> bk;
    std::vector<Torus32> ksk;
};

// Secret key set: LWE key bits, TLWE key bits, and its cloud key.
struct SecretKeyData {
    Params p;
    std::vector<int32_t> lwe_key;   // [n]   in {0,1}
    std::vector<int32_t> tlwe_key;  // [k*N] in {0,1}
    CloudKeyData cloud;             // may be empty (bk/ksk size 0) for metadata-only keys
};

}  // namespace ieache
