// See tfhe_host.h.  CPU-side key generation and bit encryption for the
// evaluator's tools and the metadata handling of the `cloud` shim.
#include "tfhe_host.h"

#include <sys/random.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace ieache {

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

Rng::Rng(const uint32_t* seed_words, int count, uint64_t stream) {
    uint64_t x = 0x243F6A8885A308D3ULL ^ (stream * 0xD1342543DE82EF95ULL);
    for (int i = 0; i < count; i++) {
        x ^= seed_words[i];
        (void)splitmix64(x);
    }
    for (int i = 0; i < 4; i++) s_[i] = splitmix64(x);
}

// ChaCha20 block function (RFC 8439 section 2.3), 64-bit block counter in words 12-13
void Rng::chacha_refill() {
    auto rotl32 = [](uint32_t v, int c) { return (v << c) | (v >> (32 - c)); };
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) st[4 + i] = key_[i];
    st[12] = (uint32_t)counter_;
    st[13] = (uint32_t)(counter_ >> 32);
    st[14] = nonce_[0];
    st[15] = nonce_[1];
    uint32_t x[16];
    memcpy(x, st, sizeof x);
    auto qr = [&](int a, int b, int c2, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c2] += x[d]; x[b] = rotl32(x[b] ^ x[c2], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c2] += x[d]; x[b] = rotl32(x[b] ^ x[c2], 7);
    };
    for (int r = 0; r < 10; r++) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) x[i] += st[i];
    memcpy(buf_, x, sizeof buf_);
    buf_pos_ = 0;
    counter_++;
}

Rng Rng::secure() {
    Rng r((uint64_t)0);
    unsigned char seed[44];
    size_t got = 0;
    while (got < sizeof seed) {
        const ssize_t k = getrandom(seed + got, sizeof seed - got, 0);
        if (k < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error("getrandom() failed: no kernel entropy for fresh encryptions");
        }
        got += (size_t)k;
    }
    memcpy(r.key_, seed, 32);
    memcpy(r.nonce_, seed + 32, 12);
    r.chacha_ = true;
    r.counter_ = 0;
    r.buf_pos_ = 8;
    return r;
}

uint64_t Rng::next() {
    if (chacha_) {
        if (buf_pos_ >= 8) chacha_refill();
        return buf_[buf_pos_++];
    }
    const uint64_t result = rotl(s_[1] * 5, 7) * 9;
    const uint64_t t = s_[1] << 17;
    s_[2] ^= s_[0];
    s_[3] ^= s_[1];
    s_[1] ^= s_[2];
    s_[0] ^= s_[3];
    s_[2] ^= t;
    s_[3] = rotl(s_[3], 45);
    return result;
}

double Rng::uniform01() { return ((next() >> 11) + 1) * (1.0 / 9007199254740992.0); }

double Rng::gaussian(double sigma) {
    if (have_spare_) {
        have_spare_ = false;
        return spare_ * sigma;
    }
    const double u = uniform01(), v = uniform01();
    const double r = std::sqrt(-2.0 * std::log(u)), th = 6.283185307179586476925 * v;
    spare_ = r * std::sin(th);
    have_spare_ = true;
    return r * std::cos(th) * sigma;
}

Torus32 Rng::gaussian_torus32(double sigma) {
    // libtfhe dtot32: fractional part scaled to 2^32
    const double d = gaussian(sigma);
    const double frac = d - std::nearbyint(d);
    return (Torus32)(uint32_t)(int64_t)std::llrint(frac * 4294967296.0);
}

int32_t modswitch_from_torus32(Torus32 phase, int32_t Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    const uint64_t half = interv / 2;
    const uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + half;
    return (int32_t)(phase64 / interv);
}

Torus32 modswitch_to_torus32(int32_t mu, int32_t Msize) {
    const uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    const uint64_t phase64 = (uint64_t)(int64_t)mu * interv;
    return (Torus32)(phase64 >> 32);
}

void lwe_encrypt_bit(const Params& p, const int32_t* lwe_key, int bit, Rng& rng, Torus32* out) {
    uint32_t b = (uint32_t)rng.gaussian_torus32(p.lwe_alpha_min) + (uint32_t)(bit ? kMU : -kMU);
    for (int32_t i = 0; i < p.n; i++) {
        const Torus32 a = rng.uniform_torus32();
        out[i] = a;
        if (lwe_key[i]) b += (uint32_t)a;
    }
    out[p.n] = (Torus32)b;
}

Torus32 lwe_phase(const Params& p, const int32_t* lwe_key, const Torus32* sample) {
    uint32_t acc = (uint32_t)sample[p.n];
    for (int32_t i = 0; i < p.n; i++)
        if (lwe_key[i]) acc -= (uint32_t)sample[i];
    return (Torus32)acc;
}

// b += a * s mod (X^N+1) for a binary key polynomial s
static void add_mul_binary_key(int32_t N, uint32_t* b, const Torus32* a, const int32_t* s) {
    for (int32_t i = 0; i < N; i++) {
        if (!s[i]) continue;
        // X^i * a : coefficient j of a lands on j+i, negated past N
        for (int32_t j = 0; j < N - i; j++) b[j + i] += (uint32_t)a[j];
        for (int32_t j = N - i; j < N; j++) b[j + i - N] -= (uint32_t)a[j];
    }
}

// Threads for the two key-generation loops: OpenMP's default is every hardware thread of the host, but a
// container is often allowed far less CPU time than that (cgroup cpu.max), and hundreds of threads on a
// 16-CPU share only contend.
static int host_threads() {
    int n = 1;
#ifdef _OPENMP
    n = omp_get_num_procs();
#endif
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = (atol(quota) + period - 1) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        fclose(f);
    }
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}

void keygen(const Params& p, const uint32_t* seed_words, int nseed, SecretKeyData* out,
            bool with_cloud) {
    out->p = p;
    // nseed < 0: every stream is its own kernel-keyed ChaCha20 generator (nothing reproducible)
    const bool secure = nseed < 0;
    auto make_rng = [&](uint64_t stream) { return secure ? Rng::secure() : Rng(seed_words, nseed, stream); };
    Rng krng = make_rng(0);
    out->lwe_key.resize(p.n);
    for (auto& b : out->lwe_key) b = krng.bit();
    out->tlwe_key.resize((size_t)p.k * p.N);
    for (auto& b : out->tlwe_key) b = krng.bit();
    out->cloud.p = p;
    out->cloud.bk.clear();
    out->cloud.ksk.clear();
    if (!with_cloud) return;

    const int n_threads = host_threads();
    (void)n_threads;
    const int32_t N = p.N, k = p.k, l = p.l, kpl = p.kpl();
    // Bootstrapping key: BK_i = TGSW_enc(lwe_key[i]) under the TLWE key.
    // One RNG stream per i so the loop can run in any order / in parallel.
    out->cloud.bk.assign(p.bk_count(), 0);
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (int32_t i = 0; i < p.n; i++) {
        Rng rng = make_rng(1 + (uint64_t)i);
        Torus32* bki = out->cloud.bk.data() + (size_t)i * kpl * (k + 1) * N;
        for (int32_t row = 0; row < kpl; row++) {
            Torus32* a = bki + (size_t)row * (k + 1) * N;  // polys a_0..a_{k-1}, then b
            uint32_t* b = (uint32_t*)(a + (size_t)k * N);
            for (int32_t j = 0; j < N; j++) b[j] = (uint32_t)rng.gaussian_torus32(p.tlwe_alpha_min);
            for (int32_t c = 0; c < k; c++) {
                for (int32_t j = 0; j < N; j++) a[(size_t)c * N + j] = rng.uniform_torus32();
                add_mul_binary_key(N, b, a + (size_t)c * N, out->tlwe_key.data() + (size_t)c * N);
            }
        }
        // + mu * H : gadget on the constant coefficient of poly `bloc` in row bloc*l+q
        for (int32_t bloc = 0; bloc <= k; bloc++)
            for (int32_t q = 0; q < l; q++) {
                const uint32_t h = 1u << (32 - (q + 1) * p.Bgbit);
                uint32_t* coef = (uint32_t*)(bki + ((size_t)(bloc * l + q) * (k + 1) + bloc) * N);
                coef[0] += (uint32_t)out->lwe_key[i] * h;
            }
    }
    // Key-switch key: KSK[i][j][d] = LWE_enc(d * tlwe_key[i] / base^(j+1)) under the n-key
    const int32_t base = p.ks_base(), n = p.n;
    out->cloud.ksk.assign(p.ksk_count(), 0);
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int32_t i = 0; i < k * N; i++) {
        Rng rng = make_rng((uint64_t)1 << 32 | (uint64_t)i);
        for (int32_t j = 0; j < p.ks_t; j++)
            for (int32_t d = 1; d < base; d++) {  // d = 0 stays all-zero: never read
                Torus32* s = out->cloud.ksk.data() + (((size_t)i * p.ks_t + j) * base + d) * (n + 1);
                uint32_t b = (uint32_t)rng.gaussian_torus32(p.lwe_alpha_min) +
                             (uint32_t)out->tlwe_key[i] * (uint32_t)d *
                                 (1u << (32 - (j + 1) * p.ks_basebit));
                for (int32_t q = 0; q < n; q++) {
                    const Torus32 a = rng.uniform_torus32();
                    s[q] = a;
                    if (out->lwe_key[q]) b += (uint32_t)a;
                }
                s[n] = (Torus32)b;
            }
    }
}

}  // namespace ieache
