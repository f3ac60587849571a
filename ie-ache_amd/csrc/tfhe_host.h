// Host-side (CPU) TFHE primitives the Cloud path needs around the GPU
// evaluator: key generation, LWE encryption/decryption of single bits, and
// the trivial gates.  These are what the reference gets from libtfhe in
// Keygen/keygen.c:22-51 (new_random_gate_bootstrapping_secret_keyset),
// Client1/alice.c:116-149 (bootsSymEncrypt) and Cloud/cloud.c:709-713,
// 822-824 (bootsSymDecrypt / bootsSymEncrypt of the metadata words).
// None of this is on the bootstrapped-gate hot path.
#pragma once
#include "params.h"

namespace ieache {

// Two generators behind one interface:
//  * seeded: xoshiro256** seeded through splitmix64 from a list of 32-bit seed words (the
//    reference seeds libtfhe with {314,1592,657} / {314,1592,888}, Keygen/keygen.c:30,34; our
//    generator is documented, not libtfhe's).  Reproducible, NOT cryptographic: for test
//    vectors and for mirroring keygen.c's fixed seeds only.
//  * Rng::secure(): a ChaCha20 key stream keyed from the kernel (getrandom(2)).  Used wherever
//    fresh ciphertexts leave the process (the metadata words main() re-encrypts under the nbit
//    key, cloud.c:822-824, 837-839) and by the tools when no seed is given.
class Rng {
public:
    Rng(const uint32_t* seed_words, int count, uint64_t stream = 0);
    explicit Rng(uint64_t seed) : Rng(nullptr, 0, seed) {}
    static Rng secure();  // throws std::runtime_error when the kernel gives no entropy
    uint64_t next();
    Torus32 uniform_torus32() { return (Torus32)(uint32_t)(next() >> 32); }
    int32_t bit() { return (int32_t)(next() >> 63); }
    double uniform01();  // (0,1]
    double gaussian(double sigma);
    // libtfhe gaussian32(0, sigma): noise on the torus
    Torus32 gaussian_torus32(double sigma);

private:
    void chacha_refill();
    uint64_t s_[4];
    bool have_spare_ = false;
    double spare_ = 0;
    bool chacha_ = false;
    uint32_t key_[8] = {0}, nonce_[3] = {0};
    uint64_t counter_ = 0;
    uint64_t buf_[8];
    int buf_pos_ = 8;
};

// Full key generation (secret + cloud key).  with_cloud=false skips BK/KSK.
// nseed >= 0: reproducible from the seed words (keygen.c:30,34 uses fixed ones);
// nseed < 0: all randomness from Rng::secure().
void keygen(const Params& p, const uint32_t* seed_words, int nseed, SecretKeyData* out,
            bool with_cloud = true);

// bootsSymEncrypt / bootsSymDecrypt on raw samples (int32[n+1]).
void lwe_encrypt_bit(const Params& p, const int32_t* lwe_key, int bit, Rng& rng, Torus32* out);
Torus32 lwe_phase(const Params& p, const int32_t* lwe_key, const Torus32* sample);
inline int lwe_decrypt_bit(const Params& p, const int32_t* lwe_key, const Torus32* sample) {
    return lwe_phase(p, lwe_key, sample) > 0;
}

// libtfhe modSwitchFromTorus32 / modSwitchToTorus32
int32_t modswitch_from_torus32(Torus32 phase, int32_t Msize);
Torus32 modswitch_to_torus32(int32_t mu, int32_t Msize);

}  // namespace ieache
