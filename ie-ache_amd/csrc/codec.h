// Codec for the three files the Cloud evaluator touches plus the key files it
// reads: cloud.key, nbit.key / secret.key, cloud.data, answer.data
// (Cloud/cloud.c:656-663, 703-766, 809-855, 899-917).
//
// These are libtfhe's tfhe_io.cpp serialisations.  libtfhe is NOT in the
// reference tree, so the layout below is restated from the upstream format
//  and pinned only by the sizes the reference leaks:
//   * one LweSample = 4n+16 bytes -> 2536 B at n=630: the hard-coded failure
//     threshold 162304 = 64 x 2536 (Cloud/dragonfly_cipher_cloud.py:1295);
//   * key-switch key dumps all `base` entries per (i,j) and one variance
//     double for the whole key (AC058.pdf p.2 "78.25 MByte" = 16 384 000 +
//     65 667 072 B for the n=500 set).
// Everything is isolated here so a real libtfhe file can correct it later
// without touching the evaluator.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "params.h"

namespace ieache {

// type tags libtfhe writes in front of binary sections
constexpr int32_t kLweSampleUid = 42;
constexpr int32_t kLweKeyUid = 43;
constexpr int32_t kLweKeySwitchKeyUid = 200;
constexpr int32_t kLweBootstrappingKeyUid = 201;
constexpr int32_t kTGswKeyUid = 202;

struct CodecError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- LweSample stream (cloud.data / answer.data) ----
// On disk: int32 uid(42) | int32 a[n] | int32 b | double current_variance.
inline size_t lwe_sample_bytes(int32_t n) { return (size_t)4 * n + 16; }
// Reads `count` samples into rows of (n+1) int32; variances (optional) into var[count].
void read_lwe_samples(FILE* f, int32_t n, size_t count, Torus32* out, double* var = nullptr);
void write_lwe_samples(FILE* f, int32_t n, size_t count, const Torus32* rows, size_t row_stride,
                       const double* var = nullptr);

// ---- parameter header (text sections) ----
void write_params(FILE* f, const Params& p);
Params read_params(FILE* f);

// ---- key files ----
void write_cloud_key(FILE* f, const CloudKeyData& ck);
void read_cloud_key(FILE* f, CloudKeyData* ck);
void write_secret_key(FILE* f, const SecretKeyData& sk);
// with_cloud=false skips over the cloud-key body without keeping it
void read_secret_key(FILE* f, SecretKeyData* sk, bool with_cloud = false);

// convenience: whole files
void save_cloud_key(const std::string& path, const CloudKeyData& ck);
void load_cloud_key(const std::string& path, CloudKeyData* ck);
void save_secret_key(const std::string& path, const SecretKeyData& sk);
void load_secret_key(const std::string& path, SecretKeyData* sk, bool with_cloud = false);

}  // namespace ieache
