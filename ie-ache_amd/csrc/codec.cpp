// See codec.h.
#include "codec.h"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>

namespace ieache {

namespace {

void xread(FILE* f, void* dst, size_t bytes, const char* what) {
    if (bytes && fread(dst, 1, bytes, f) != bytes) throw CodecError(std::string("short read: ") + what);
}
void xwrite(FILE* f, const void* src, size_t bytes) {
    if (bytes && fwrite(src, 1, bytes, f) != bytes) throw CodecError("short write");
}
int32_t read_i32(FILE* f, const char* what) {
    int32_t v;
    xread(f, &v, 4, what);
    return v;
}
void expect_uid(FILE* f, int32_t uid, const char* what) {
    const int32_t got = read_i32(f, what);
    if (got != uid)
        throw CodecError(std::string("bad type tag for ") + what + ": got " + std::to_string(got) + ", want " +
                         std::to_string(uid));
}

// "-----BEGIN <TITLE>-----\nname: value\n...\n-----END <TITLE>-----\n"
using Props = std::map<std::string, std::string>;

void write_section(FILE* f, const char* title, const Props& props) {
    fprintf(f, "-----BEGIN %s-----\n", title);
    for (const auto& kv : props) fprintf(f, "%s: %s\n", kv.first.c_str(), kv.second.c_str());
    fprintf(f, "-----END %s-----\n", title);
}

std::string read_line(FILE* f) {
    std::string s;
    int c;
    while ((c = fgetc(f)) != EOF && c != '\n') s.push_back((char)c);
    if (c == EOF && s.empty()) throw CodecError("unexpected end of file in text section");
    if (!s.empty() && s.back() == '\r') s.pop_back();
    return s;
}

Props read_section(FILE* f, const char* title) {
    const std::string begin = std::string("-----BEGIN ") + title + "-----";
    const std::string end = std::string("-----END ") + title + "-----";
    std::string line = read_line(f);
    if (line != begin) throw CodecError("expected '" + begin + "', got '" + line.substr(0, 60) + "'");
    Props p;
    for (;;) {
        line = read_line(f);
        if (line == end) break;
        const size_t colon = line.find(": ");
        if (colon == std::string::npos) throw CodecError("malformed property line in " + std::string(title));
        p[line.substr(0, colon)] = line.substr(colon + 2);
    }
    return p;
}

std::string fmt_double(double v) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.17g", v);
    return buf;
}
int32_t prop_int(const Props& p, const char* name) {
    auto it = p.find(name);
    if (it == p.end()) throw CodecError(std::string("missing property ") + name);
    return (int32_t)strtol(it->second.c_str(), nullptr, 10);
}
double prop_double(const Props& p, const char* name) {
    auto it = p.find(name);
    if (it == p.end()) throw CodecError(std::string("missing property ") + name);
    return strtod(it->second.c_str(), nullptr);  // decimal, exponent or hex-float
}

}  // namespace

void read_lwe_samples(FILE* f, int32_t n, size_t count, Torus32* out, double* var) {
    for (size_t i = 0; i < count; i++) {
        expect_uid(f, kLweSampleUid, "LweSample");
        xread(f, out + i * (size_t)(n + 1), (size_t)4 * (n + 1), "LweSample coefficients");
        double v;
        xread(f, &v, 8, "LweSample variance");
        if (var) var[i] = v;
    }
}

void write_lwe_samples(FILE* f, int32_t n, size_t count, const Torus32* rows, size_t row_stride, const double* var) {
    for (size_t i = 0; i < count; i++) {
        xwrite(f, &kLweSampleUid, 4);
        xwrite(f, rows + i * row_stride, (size_t)4 * (n + 1));
        const double v = var ? var[i] : 0.0;
        xwrite(f, &v, 8);
    }
}

// libtfhe write_tfheGateBootstrappingParameters: GATEBOOTSPARAMS, LWEPARAMS,
// TGSWPARAMS, TLWEPARAMS
void write_params(FILE* f, const Params& p) {
    write_section(f, "GATEBOOTSPARAMS", {{"ks_basebit", std::to_string(p.ks_basebit)}, {"ks_t", std::to_string(p.ks_t)}});
    write_section(f, "LWEPARAMS", {{"alpha_max", fmt_double(p.lwe_alpha_max)},
                                    {"alpha_min", fmt_double(p.lwe_alpha_min)},
                                    {"n", std::to_string(p.n)}});
    write_section(f, "TGSWPARAMS", {{"Bgbit", std::to_string(p.Bgbit)}, {"l", std::to_string(p.l)}});
    write_section(f, "TLWEPARAMS", {{"N", std::to_string(p.N)},
                                     {"alpha_max", fmt_double(p.tlwe_alpha_max)},
                                     {"alpha_min", fmt_double(p.tlwe_alpha_min)},
                                     {"k", std::to_string(p.k)}});
}

Params read_params(FILE* f) {
    Params p;
    Props s = read_section(f, "GATEBOOTSPARAMS");
    p.ks_t = prop_int(s, "ks_t");
    p.ks_basebit = prop_int(s, "ks_basebit");
    s = read_section(f, "LWEPARAMS");
    p.n = prop_int(s, "n");
    p.lwe_alpha_min = prop_double(s, "alpha_min");
    p.lwe_alpha_max = prop_double(s, "alpha_max");
    s = read_section(f, "TGSWPARAMS");
    p.l = prop_int(s, "l");
    p.Bgbit = prop_int(s, "Bgbit");
    s = read_section(f, "TLWEPARAMS");
    p.N = prop_int(s, "N");
    p.k = prop_int(s, "k");
    p.tlwe_alpha_min = prop_double(s, "alpha_min");
    p.tlwe_alpha_max = prop_double(s, "alpha_max");
    if (!p.supported()) throw CodecError("parameter set in key header is not supported");
    return p;
}

// libtfhe write_lweBootstrappingKey_content: tag, key-switch key (tag, max
// variance, all coefficients), max variance, all TGSW coefficients.
static void write_cloud_body(FILE* f, const CloudKeyData& ck) {
    if (ck.bk.size() != ck.p.bk_count() || ck.ksk.size() != ck.p.ksk_count())
        throw CodecError("cloud key arrays do not match the parameter set");
    const double var = 0.0;
    xwrite(f, &kLweBootstrappingKeyUid, 4);
    xwrite(f, &kLweKeySwitchKeyUid, 4);
    xwrite(f, &var, 8);
    xwrite(f, ck.ksk.data(), ck.ksk.size() * 4);
    xwrite(f, &var, 8);
    xwrite(f, ck.bk.data(), ck.bk.size() * 4);
}

static void read_cloud_body(FILE* f, const Params& p, CloudKeyData* ck) {
    double var;
    expect_uid(f, kLweBootstrappingKeyUid, "bootstrapping key");
    expect_uid(f, kLweKeySwitchKeyUid, "key-switch key");
    xread(f, &var, 8, "key-switch variance");
    if (ck) {
        ck->p = p;
        ck->ksk.resize(p.ksk_count());
        xread(f, ck->ksk.data(), ck->ksk.size() * 4, "key-switch key body");
    } else if (fseek(f, (long)(p.ksk_count() * 4), SEEK_CUR) != 0) {
        throw CodecError("seek failed in key-switch key body");
    }
    xread(f, &var, 8, "bootstrapping key variance");
    if (ck) {
        ck->bk.resize(p.bk_count());
        xread(f, ck->bk.data(), ck->bk.size() * 4, "bootstrapping key body");
    } else if (fseek(f, (long)(p.bk_count() * 4), SEEK_CUR) != 0) {
        throw CodecError("seek failed in bootstrapping key body");
    }
}

void write_cloud_key(FILE* f, const CloudKeyData& ck) {
    write_params(f, ck.p);
    write_cloud_body(f, ck);
}

void read_cloud_key(FILE* f, CloudKeyData* ck) {
    const Params p = read_params(f);
    read_cloud_body(f, p, ck);
}

// libtfhe write_tfheGateBootstrappingSecretKeySet: cloud key set, LWE key, TGSW key
void write_secret_key(FILE* f, const SecretKeyData& sk) {
    if ((int32_t)sk.lwe_key.size() != sk.p.n || (int32_t)sk.tlwe_key.size() != sk.p.k * sk.p.N)
        throw CodecError("secret key arrays do not match the parameter set");
    write_cloud_key(f, sk.cloud);
    xwrite(f, &kLweKeyUid, 4);
    xwrite(f, sk.lwe_key.data(), sk.lwe_key.size() * 4);
    xwrite(f, &kTGswKeyUid, 4);
    xwrite(f, sk.tlwe_key.data(), sk.tlwe_key.size() * 4);
}

void read_secret_key(FILE* f, SecretKeyData* sk, bool with_cloud) {
    sk->p = read_params(f);
    read_cloud_body(f, sk->p, with_cloud ? &sk->cloud : nullptr);
    if (!with_cloud) {
        sk->cloud.p = sk->p;
        sk->cloud.bk.clear();
        sk->cloud.ksk.clear();
    }
    expect_uid(f, kLweKeyUid, "LWE key");
    sk->lwe_key.resize(sk->p.n);
    xread(f, sk->lwe_key.data(), sk->lwe_key.size() * 4, "LWE key bits");
    expect_uid(f, kTGswKeyUid, "TGSW key");
    sk->tlwe_key.resize((size_t)sk->p.k * sk->p.N);
    xread(f, sk->tlwe_key.data(), sk->tlwe_key.size() * 4, "TGSW key bits");
}

namespace {
struct File {
    FILE* f;
    File(const std::string& path, const char* mode) : f(fopen(path.c_str(), mode)) {
        if (!f) throw CodecError("cannot open " + path + ": " + strerror(errno));
    }
    ~File() {
        if (f) fclose(f);
    }
};
}  // namespace

void save_cloud_key(const std::string& path, const CloudKeyData& ck) {
    File f(path, "wb");
    write_cloud_key(f.f, ck);
}
void load_cloud_key(const std::string& path, CloudKeyData* ck) {
    File f(path, "rb");
    read_cloud_key(f.f, ck);
}
void save_secret_key(const std::string& path, const SecretKeyData& sk) {
    File f(path, "wb");
    write_secret_key(f.f, sk);
}
void load_secret_key(const std::string& path, SecretKeyData* sk, bool with_cloud) {
    File f(path, "rb");
    read_secret_key(f.f, sk, with_cloud);
}

}  // namespace ieache
