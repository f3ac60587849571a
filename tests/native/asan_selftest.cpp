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
    // the same add recorded and replayed level-parallel (orc_defer_*): identical samples
    {
        std::vector<Torus32> sum2(3 * S), co2(S);
        orc_defer_begin(ock);
        orc_add(ock, sum2.data(), co2.data(), x.data(), y.data(), c.data(), 3);
        CHECK(orc_defer_run(ock, 2) >= 3);
        CHECK(sum2 == sum && co2 == co);
        // bootsMUX(c, x0, y0)
        std::vector<Torus32> m(S);
        orc_gate_mux(ock, m.data(), c.data(), x.data(), y.data());
        CHECK(lwe_decrypt_bit(p, sk.lwe_key.data(), m.data()) == (5 & 1));  // c = 1 selects x bit 0
        std::vector<Torus32> gb(4 * S);
        orc_gates_batch(ock, 1, 4, gb.data(), a.data(), b.data(), 2);
        for (int i = 0; i < 4; i++) CHECK(lwe_decrypt_bit(p, sk.lwe_key.data(), gb.data() + i * S) == ((i >> 1) ^ (i & 1)));
    }
    orc_cloudkey_free(ock);

    // the tolerant key reader on a header in another section order with an untagged, variance-free body
    {
        FILE* f = fopen((tmp + "/asan_alt.key").c_str(), "wb");
        CHECK(f != nullptr);
        fprintf(f, "-----BEGIN TGSWPARAMS-----\nl: %d\nBgbit: %d\n-----END TGSWPARAMS-----\n", p.l, p.Bgbit);
        fprintf(f, "-----BEGIN LWEPARAMS-----\nn: %d\nalpha_min: %.17g\nalpha_max: %.17g\n-----END LWEPARAMS-----\n", p.n, p.lwe_alpha_min, p.lwe_alpha_max);
        fprintf(f, "-----BEGIN GATEBOOTSPARAMS-----\nks_t: %d\nks_basebit: %d\n-----END GATEBOOTSPARAMS-----\n", p.ks_t, p.ks_basebit);
        fprintf(f, "-----BEGIN TLWEPARAMS-----\nN: %d\nk: %d\nalpha_min: %.17g\nalpha_max: %.17g\n-----END TLWEPARAMS-----\n", p.N, p.k, p.tlwe_alpha_min, p.tlwe_alpha_max);
        fwrite(sk.cloud.ksk.data(), 4, sk.cloud.ksk.size(), f);
        fwrite(sk.cloud.bk.data(), 4, sk.cloud.bk.size(), f);
        fclose(f);
        CloudKeyData alt;
        load_cloud_key(tmp + "/asan_alt.key", &alt);
        CHECK(alt.bk == sk.cloud.bk && alt.ksk == sk.cloud.ksk && alt.p.n == p.n);
        CHECK(last_key_layout().find("no variance") != std::string::npos);
        CHECK(load_params(tmp + "/asan_alt.key").Bgbit == p.Bgbit);
        f = fopen((tmp + "/asan_short.key").c_str(), "wb");
        fprintf(f, "-----BEGIN GATEBOOTSPARAMS-----\nks_t: 8\n");  // truncated inside a section
        fclose(f);
        bool refused = false;
        try {
            load_cloud_key(tmp + "/asan_short.key", &alt);
        } catch (const CodecError&) {
            refused = true;
        }
        CHECK(refused);
    }
    // kernel-keyed generator (ChaCha20): runs, and two instances differ
    {
        Rng r1 = Rng::secure(), r2 = Rng::secure();
        uint64_t acc1 = 0, acc2 = 0;
        for (int i = 0; i < 40; i++) {
            acc1 ^= r1.next();
            acc2 ^= r2.next();
        }
        CHECK(acc1 != acc2);
    }

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
    // folded and chained builders: same plaintext function, fewer gates
    {
        Circuit plain, folded, chain;
        CHECK(build_circuit(CIRC_MUL, 32, &plain) && build_circuit(CIRC_MUL, 32, &folded, true, true));
        CHECK(folded.n_bootstraps < plain.n_bootstraps && folded.n_reference_bootstraps == plain.n_bootstraps);
        CHECK(build_circuit(chain_kind(CIRC_ADD, CIRC_SUB, true), 32, &chain));
        std::vector<uint8_t> in(plain.n_inputs, 0), r1(plain.outputs.size()), r2(plain.outputs.size());
        for (size_t i = 0; i < 64; i++) in[i] = (uint8_t)((0xDEADBEEF12345678ull >> i) & 1);
        simulate_circuit(plain, in.data(), r1.data());
        simulate_circuit(folded, in.data(), r2.data());
        CHECK(r1 == r2);
        std::vector<uint8_t> cin(chain.n_inputs, 0), cres(chain.outputs.size());
        cin[3] = 1;        // A = 8
        cin[32] = 1;       // B = 1
        cin[96 + 1] = 1;   // C = 2
        simulate_circuit(chain, cin.data(), cres.data());
        uint32_t got = 0;
        for (int i = 0; i < 32; i++) got |= (uint32_t)cres[i] << i;
        CHECK(got == 8 + 1 - 2);
    }
    Circuit bad;
    CHECK(!build_circuit(CIRC_MUL, 256, &bad));
    CHECK(!build_circuit(chain_kind(CIRC_MUL, CIRC_MUL, true), 128, &bad));
    printf("ASAN_SELFTEST_OK\n");
    return 0;
}
