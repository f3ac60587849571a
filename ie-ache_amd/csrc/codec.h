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
// Writer: libtfhe's order as best known (tfhe_io.cpp write_tfheGateBootstrappingParameters ->
// GATEBOOTSPARAMS, LWEPARAMS, then write_tGswParams, which emits its TLWEPARAMS before TGSWPARAMS).
void write_params(FILE* f, const Params& p);

// ---- key files ----
// Writers (the layout this build's tools produce; fixtures are written with it).
void write_cloud_key(FILE* f, const CloudKeyData& ck);
void write_secret_key(FILE* f, const SecretKeyData& sk);
void save_cloud_key(const std::string& path, const CloudKeyData& ck);
void save_secret_key(const std::string& path, const SecretKeyData& sk);

// Readers (SURVEY App. B "codec strategy").  libtfhe is not in the reference tree and no file
// written by it exists here, so the reader does not assume one layout:
//   * text sections ("-----BEGIN <TITLE>-----" ... "-----END <TITLE>-----") are located by TITLE,
//     wherever they stand and in any order; properties are read by name;
//   * what is left is binary.  Its layout is chosen among enumerated hypotheses -- type tags
//     present or not, no / one / per-sample variance doubles, key-switch entries with or without
//     the never-read d = 0 rows, key-switch key before or after the bootstrapping key, secret key
//     bits before or after the cloud-key body -- by solving for the exact remaining byte count,
//     then checking every tag the hypothesis expects and that secret-key words are bits.
// The hypothesis that matched is kept in last_key_layout() (and reported by the C ABI).
// with_cloud=false locates the cloud-key body without keeping it (nbit.key: metadata only).
Params load_params(const std::string& path);
void load_cloud_key(const std::string& path, CloudKeyData* ck);
void load_secret_key(const std::string& path, SecretKeyData* sk, bool with_cloud = false);
const std::string& last_key_layout();

}  // namespace ieache
