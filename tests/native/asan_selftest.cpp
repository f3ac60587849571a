// Host-side code (product C++ without the GPU parts + the C oracle) under
// AddressSanitizer / UBSan.  GPU sanitizers are not available on the pool, so this is
// where memory errors in the codec, the circuit builder, keygen and the oracle would show.
// Built and run by tests/test_sanitizers_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../ie-ache_amd/csrc/circuit.h"
#include "../../ie-ache_amd/csrc/codec.h"
#include "../../ie-ache_amd/csrc/tfhe_host.h"
#include "../../oracle/tfhe_oracle.h"

using namespace ieache;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    Params p;
    p.n = 7;
    p.N = 32;
    const uint32_t seed[3] = {5, 6, 7};
    SecretKeyData sk;
    keygen(p, seed, 3, &sk, true);
    CHECK(sk.cloud.bk.size() == p.bk_count() && sk.cloud.ksk.size() == p.ksk_count());

    // codec round trips
    save_secret_key(tmp + "/asan_secret.key", sk);
    save_cloud_key(tmp + "/asan_cloud.key", sk.cloud);
    SecretKeyData sk2;
    load_secret_key(tmp + "/asan_secret.key", &sk2, true);
    CHECK(sk2.lwe_key == sk.lwe_key && sk2.tlwe_key == sk.tlwe_key && sk2.cloud.bk == sk.cloud.bk && sk2.cloud.ksk == sk.cloud.ksk);
    CloudKeyData ck2;
    load_cloud_key(tmp + "/asan_cloud.key", &ck2);
    CHECK(ck2.bk == sk.cloud.bk);
    bool threw = false;
    try {
        load_cloud_key(tmp + "/asan_secret.key.does-not-exist", &ck2);
    } catch (const CodecError&) {
        threw = true;
    }
    CHECK(threw);

    // encrypt / oracle gates / decrypt
    Rng rng(99);
    const size_t S = (size_t)p.n + 1;
    std::vector<Torus32> a(4 * S), b(4 * S), out(S);
    for (int i = 0; i < 4; i++) {
        lwe_encrypt_bit(p, sk.lwe_key.data(), i >> 1, rng, a.data() + i * S);
        lwe_encrypt_bit(p, sk.lwe_key.data(), i & 1, rng, b.data() + i * S);
    }
    orc_params op{p.n, p.N, p.k, p.l, p.Bgbit, p.ks_t, p.ks_basebit};
    orc_cloudkey* ock = orc_cloudkey_new(&op, sk.cloud.bk.data(), sk.cloud.ksk.data());
    CHECK(ock != nullptr);
    for (int mode = 0; mode < 3; mode++) {
        orc_cloudkey_set_polymul(ock, mode);
        for (int i = 0; i < 4; i++) {
            orc_gate_and(ock, out.data(), a.data() + i * S, b.data() + i * S);
            CHECK(lwe_decrypt_bit(p, sk.lwe_key.data(), out.data()) == ((i >> 1) & (i & 1)));
            orc_gate_xor(ock, out.data(), a.data() + i * S, b.data() + i * S);
            CHECK(lwe_decrypt_bit(p, sk.lwe_key.data(), out.data()) == ((i >> 1) ^ (i & 1)));
        }
    }
    orc_cloudkey_set_polymul(ock, ORC_POLYMUL_NTT);
    // a 3-bit ripple add through the oracle's circuit code
    std::vector<Torus32> x(3 * S), y(3 * S), c(S), sum(3 * S), co(S);
    for (int i = 0; i < 3; i++) {
        lwe_encrypt_bit(p, sk.lwe_key.data(), (5 >> i) & 1, rng, x.data() + i * S);
        lwe_encrypt_bit(p, sk.lwe_key.data(), (6 >> i) & 1, rng, y.data() + i * S);
    }
    lwe_encrypt_bit(p, sk.lwe_key.data(), 1, rng, c.data());
    orc_add(ock, sum.data(), co.data(), x.data(), y.data(), c.data(), 3);
    int v = 0;
    for (int i = 0; i < 3; i++) v |= lwe_decrypt_bit(p, sk.lwe_key.data(), sum.data() + i * S) << i;
    v |= lwe_decrypt_bit(p, sk.lwe_key.data(), co.data()) << 3;
    CHECK(v == 5 + 6 + 1);
    orc_cloudkey_free(ock);

    // circuit builder + plaintext simulation of every kind
    for (int kind : {CIRC_ADD, CIRC_SUB, CIRC_RSUB, CIRC_MUL, CIRC_MULADD}) {
        const int bits = kind == CIRC_MULADD ? 64 : 32;
        Circuit circ;
        CHECK(build_circuit(kind, bits, &circ));
        std::vector<uint8_t> in(circ.n_inputs, 0), res(circ.outputs.size());
        in[0] = 1;         // A = 1
        in[bits + 1] = 1;  // B = 2
        simulate_circuit(circ, in.data(), res.data());
        uint64_t got = 0;
        for (size_t i = 0; i < 64 && i < res.size(); i++) got |= (uint64_t)res[i] << i;
        const uint64_t m = bits == 64 ? ~0ull : ((1ull << bits) - 1);
        const uint64_t want = kind == CIRC_ADD ? 3 : kind == CIRC_SUB ? ((1 - 2) & m) : kind == CIRC_RSUB ? 1 : 2;
        CHECK((got & (kind == CIRC_MUL || kind == CIRC_MULADD ? ~0ull : m)) == want);
    }
    Circuit bad;
    CHECK(!build_circuit(CIRC_MUL, 256, &bad));
    printf("ASAN_SELFTEST_OK\n");
    return 0;
}
