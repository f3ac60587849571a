// See codec.h.
#include "codec.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
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

// libtfhe write_tfheGateBootstrappingParameters: GATEBOOTSPARAMS, LWEPARAMS, then
// write_tGswParams = TLWEPARAMS followed by TGSWPARAMS
void write_params(FILE* f, const Params& p) {
    write_section(f, "GATEBOOTSPARAMS", {{"ks_basebit", std::to_string(p.ks_basebit)}, {"ks_t", std::to_string(p.ks_t)}});
    write_section(f, "LWEPARAMS", {{"alpha_max", fmt_double(p.lwe_alpha_max)},
                                    {"alpha_min", fmt_double(p.lwe_alpha_min)},
                                    {"n", std::to_string(p.n)}});
    write_section(f, "TLWEPARAMS", {{"N", std::to_string(p.N)},
                                     {"alpha_max", fmt_double(p.tlwe_alpha_max)},
                                     {"alpha_min", fmt_double(p.tlwe_alpha_min)},
                                     {"k", std::to_string(p.k)}});
    write_section(f, "TGSWPARAMS", {{"Bgbit", std::to_string(p.Bgbit)}, {"l", std::to_string(p.l)}});
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

void write_cloud_key(FILE* f, const CloudKeyData& ck) {
    write_params(f, ck.p);
    write_cloud_body(f, ck);
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

// ------------------------------------------------------------------------
// Tolerant reader
// ------------------------------------------------------------------------
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

// read-only mapping of a whole file
struct Mapped {
    const unsigned char* data = nullptr;
    size_t size = 0;
    explicit Mapped(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw CodecError("cannot open " + path + ": " + strerror(errno));
        struct stat st;
        if (fstat(fd, &st) != 0) {
            close(fd);
            throw CodecError("cannot stat " + path);
        }
        size = (size_t)st.st_size;
        if (size) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                close(fd);
                throw CodecError("cannot map " + path + ": " + strerror(errno));
            }
            data = static_cast<const unsigned char*>(m);
        }
        close(fd);
    }
    ~Mapped() {
        if (data) munmap(const_cast<unsigned char*>(data), size);
    }
    Mapped(const Mapped&) = delete;
    Mapped& operator=(const Mapped&) = delete;
};

struct TextSection {
    std::string title;
    Props props;
    size_t begin, end;  // byte range in the file, END line and its newline included
};

// Finds every "-----BEGIN T-----\n ... -----END T-----\n" block.  A 16-byte ASCII marker turning
// up inside key material by chance has probability ~2^-100 per position; a candidate must also
// parse completely (printable "name: value" lines up to its END line) to count.
std::vector<TextSection> scan_sections(const Mapped& m) {
    static const char kBegin[] = "-----BEGIN ";
    const size_t blen = sizeof(kBegin) - 1;
    std::vector<TextSection> out;
    size_t pos = 0;
    while (pos + blen < m.size) {
        const void* hit = memmem(m.data + pos, m.size - pos, kBegin, blen);
        if (!hit) break;
        const size_t at = (size_t)(static_cast<const unsigned char*>(hit) - m.data);
        pos = at + 1;
        auto line_at = [&](size_t from, std::string* line) -> size_t {  // returns position after '\n', 0 on failure
            line->clear();
            for (size_t i = from; i < m.size && i < from + 256; i++) {
                const unsigned char ch = m.data[i];
                if (ch == '\n') return i + 1;
                if (ch == '\r') continue;
                if (ch < 0x20 || ch > 0x7e) return 0;
                line->push_back((char)ch);
            }
            return 0;
        };
        std::string line;
        size_t next = line_at(at, &line);
        if (!next || line.size() < blen + 6 || line.compare(line.size() - 5, 5, "-----") != 0) continue;
        TextSection s;
        s.title = line.substr(blen, line.size() - blen - 5);
        s.begin = at;
        const std::string endline = "-----END " + s.title + "-----";
        bool ok = false;
        for (int guard = 0; guard < 64; guard++) {
            const size_t after = line_at(next, &line);
            if (!after) break;
            next = after;
            if (line == endline) {
                ok = true;
                break;
            }
            const size_t colon = line.find(": ");
            if (colon == std::string::npos) break;
            s.props[line.substr(0, colon)] = line.substr(colon + 2);
        }
        if (!ok) continue;
        s.end = next;
        out.push_back(std::move(s));
        pos = next;
    }
    return out;
}

const TextSection* find_section(const std::vector<TextSection>& secs, const char* title) {
    for (const auto& s : secs)
        if (s.title == title) return &s;  // the first one: the gate-bootstrapping header comes first in every layout
    return nullptr;
}

Params params_from_sections(const std::vector<TextSection>& secs) {
    Params p;
    const TextSection* g = find_section(secs, "GATEBOOTSPARAMS");
    const TextSection* lw = find_section(secs, "LWEPARAMS");
    const TextSection* tg = find_section(secs, "TGSWPARAMS");
    const TextSection* tl = find_section(secs, "TLWEPARAMS");
    for (const auto& need : {std::make_pair(g, "GATEBOOTSPARAMS"), std::make_pair(lw, "LWEPARAMS"),
                             std::make_pair(tg, "TGSWPARAMS"), std::make_pair(tl, "TLWEPARAMS")})
        if (!need.first) throw CodecError(std::string("key header has no ") + need.second + " section");
    p.ks_t = prop_int(g->props, "ks_t");
    p.ks_basebit = prop_int(g->props, "ks_basebit");
    p.n = prop_int(lw->props, "n");
    p.lwe_alpha_min = prop_double(lw->props, "alpha_min");
    p.lwe_alpha_max = prop_double(lw->props, "alpha_max");
    p.l = prop_int(tg->props, "l");
    p.Bgbit = prop_int(tg->props, "Bgbit");
    p.N = prop_int(tl->props, "N");
    p.k = prop_int(tl->props, "k");
    p.tlwe_alpha_min = prop_double(tl->props, "alpha_min");
    p.tlwe_alpha_max = prop_double(tl->props, "alpha_max");
    if (!p.supported()) throw CodecError("parameter set in key header is not supported");
    // an LWEKSPARAMS section (libtfhe writes one in front of a stand-alone key-switch key), if present, must agree
    if (const TextSection* ks = find_section(secs, "LWEKSPARAMS")) {
        auto has = [&](const char* k) { return ks->props.count(k) != 0; };
        if ((has("t") && prop_int(ks->props, "t") != p.ks_t) || (has("basebit") && prop_int(ks->props, "basebit") != p.ks_basebit))
            throw CodecError("LWEKSPARAMS disagrees with GATEBOOTSPARAMS");
    }
    return p;
}

// the file minus its text sections, addressed as one contiguous byte stream
struct BinaryStream {
    const Mapped& m;
    std::vector<std::pair<size_t, size_t>> spans;  // [begin, end) in the file
    size_t total = 0;
    BinaryStream(const Mapped& mm, const std::vector<TextSection>& secs) : m(mm) {
        size_t at = 0;
        for (const auto& s : secs) {
            if (s.begin > at) spans.emplace_back(at, s.begin);
            at = s.end;
        }
        if (m.size > at) spans.emplace_back(at, m.size);
        for (const auto& sp : spans) total += sp.second - sp.first;
    }
    void read(size_t pos, size_t len, void* dst) const {
        unsigned char* out = static_cast<unsigned char*>(dst);
        for (const auto& sp : spans) {
            const size_t slen = sp.second - sp.first;
            if (pos >= slen) {
                pos -= slen;
                continue;
            }
            const size_t take = std::min(len, slen - pos);
            memcpy(out, m.data + sp.first + pos, take);
            out += take;
            len -= take;
            pos = 0;
            if (!len) return;
        }
        if (len) throw CodecError("short read: key body");
    }
    int32_t i32(size_t pos) const {
        int32_t v;
        read(pos, 4, &v);
        return v;
    }
};

enum FieldKind { F_UID, F_VAR, F_KSK, F_BK, F_LWEKEY, F_TGSWKEY };
struct Field {
    FieldKind kind;
    int32_t uid = 0;          // F_UID: expected tag
    bool skip_d0 = false;     // F_KSK: the d = 0 entries are not in the file
    bool sample_var = false;  // F_KSK / F_BK: a variance double follows every sample / TLWE row
    std::string name;
};
struct Layout {
    std::vector<Field> fields;
    std::string desc;
    size_t bytes(const Params& p) const {
        size_t b = 0;
        for (const Field& f : fields) b += field_bytes(f, p);
        return b;
    }
    static size_t field_bytes(const Field& f, const Params& p) {
        switch (f.kind) {
            case F_UID: return 4;
            case F_VAR: return 8;
            case F_KSK: {
                const size_t samples = (size_t)p.k * p.N * p.ks_t * (p.ks_base() - (f.skip_d0 ? 1 : 0));
                return samples * ((size_t)(p.n + 1) * 4 + (f.sample_var ? 8 : 0));
            }
            case F_BK: {
                const size_t rows = (size_t)p.n * p.kpl();
                return rows * ((size_t)(p.k + 1) * p.N * 4 + (f.sample_var ? 8 : 0));
            }
            case F_LWEKEY: return (size_t)p.n * 4;
            case F_TGSWKEY: return (size_t)p.k * p.N * 4;
        }
        return 0;
    }
};

// every cloud-key body layout considered; this build's own writer is hypothesis 0
std::vector<Layout> cloud_body_layouts() {
    std::vector<Layout> out;
    for (int ks_first = 1; ks_first >= 0; ks_first--)
        for (int bk_uid = 1; bk_uid >= 0; bk_uid--)
            for (int ks_uid = 1; ks_uid >= 0; ks_uid--)
                for (int ks_var = 1; ks_var <= 3; ks_var++)          // 1: one double, 2: none, 3: one per sample
                    for (int skip_d0 = 0; skip_d0 <= 1; skip_d0++)
                        for (int bk_var = 1; bk_var <= 3; bk_var++) {  // 1: one double, 2: none, 3: one per TLWE row
                            Layout L;
                            auto uid = [&](int32_t v, const char* nm) {
                                Field f{F_UID};
                                f.uid = v;
                                f.name = nm;
                                L.fields.push_back(f);
                            };
                            auto ks = [&] {
                                if (ks_uid) uid(kLweKeySwitchKeyUid, "key-switch tag");
                                if (ks_var == 1) L.fields.push_back(Field{F_VAR});
                                Field f{F_KSK};
                                f.skip_d0 = skip_d0;
                                f.sample_var = ks_var == 3;
                                L.fields.push_back(f);
                            };
                            auto bk = [&] {
                                if (bk_var == 1) L.fields.push_back(Field{F_VAR});
                                Field f{F_BK};
                                f.sample_var = bk_var == 3;
                                L.fields.push_back(f);
                            };
                            if (bk_uid) uid(kLweBootstrappingKeyUid, "bootstrapping-key tag");
                            if (ks_first) {
                                ks();
                                bk();
                            } else {
                                bk();
                                ks();
                            }
                            static const char* vn[4] = {"", "one variance", "no variance", "variance per sample"};
                            L.desc = std::string(bk_uid ? "bk tag, " : "") + (ks_first ? "KS{" : "BK{") +
                                     (ks_first ? std::string(ks_uid ? "tag, " : "") + vn[ks_var] + (skip_d0 ? ", d=0 rows omitted" : ", all base rows")
                                               : std::string(vn[bk_var])) +
                                     "} then " + (ks_first ? "BK{" : "KS{") +
                                     (ks_first ? std::string(vn[bk_var])
                                               : std::string(ks_uid ? "tag, " : "") + vn[ks_var] + (skip_d0 ? ", d=0 rows omitted" : ", all base rows")) +
                                     "}";
                            out.push_back(std::move(L));
                        }
    return out;
}

std::vector<Layout> secret_layouts() {
    std::vector<Layout> out;
    const std::vector<Layout> bodies = cloud_body_layouts();
    for (int cloud_first = 1; cloud_first >= 0; cloud_first--)
        for (int lwe_first = 1; lwe_first >= 0; lwe_first--)
            for (int key_uids = 1; key_uids >= 0; key_uids--)
                for (const Layout& body : bodies) {
                    Layout L;
                    auto keys = [&] {
                        for (int q = 0; q < 2; q++) {
                            const bool lwe = (q == 0) == (lwe_first != 0);
                            if (key_uids) {
                                Field u{F_UID};
                                u.uid = lwe ? kLweKeyUid : kTGswKeyUid;
                                u.name = lwe ? "LWE key tag" : "TGSW key tag";
                                L.fields.push_back(u);
                            }
                            L.fields.push_back(Field{lwe ? F_LWEKEY : F_TGSWKEY});
                        }
                    };
                    if (!cloud_first) keys();
                    L.fields.insert(L.fields.end(), body.fields.begin(), body.fields.end());
                    if (cloud_first) keys();
                    L.desc = std::string(cloud_first ? "cloud body [" : "secret keys, then cloud body [") + body.desc + "]" +
                             (cloud_first ? ", then " : "; ") + (lwe_first ? "LWE key, TGSW key" : "TGSW key, LWE key") +
                             (key_uids ? " (tagged)" : " (untagged)");
                    out.push_back(std::move(L));
                }
    return out;
}

thread_local std::string g_layout;

// Chooses the layout whose byte count equals the binary part of the file and whose tags / key bits
// check out, then decodes it.  Any output pointer may be null (that array is only located).
void decode(const std::string& path, const std::vector<Layout>& candidates, const Params& p, const BinaryStream& bin,
            CloudKeyData* ck, SecretKeyData* sk) {
    const Layout* chosen = nullptr;
    size_t size_matches = 0;
    std::string why;
    std::vector<const Layout*> passing;
    std::vector<char> passing_d0_zero;  // per passing layout: every sampled d = 0 row of the key-switch key was all zero
    auto plausible_variance = [&](size_t at) {
        // a variance is -1 (libtfhe's "unset"), 0, or a small positive number.  Eight bytes of uniformly random key
        // material pass this with probability ~1/4 (sign bit clear and exponent below 1023), and eight ZERO bytes -- the
        // head of a key-switch key that carries its d = 0 rows -- always do, so a single field proves little: the
        // structural checks below (every per-sample variance, all-zero d = 0 rows) and the uniqueness test after the
        // loop are what separate layouts of equal size that differ only in where the doubles stand.
        double v;
        bin.read(at, 8, &v);
        return v == -1.0 || (v >= 0.0 && v < 1.0);
    };
    for (const Layout& L : candidates) {
        if (L.bytes(p) != bin.total) continue;
        size_matches++;
        bool ok = true, d0_zero = true;
        size_t pos = 0;
        auto reject = [&](const std::string& msg) {
            if (why.empty()) why = msg;
            ok = false;
        };
        for (const Field& f : L.fields) {
            if (f.kind == F_UID && bin.i32(pos) != f.uid) {
                reject("expected " + f.name + " " + std::to_string(f.uid) + ", found " + std::to_string(bin.i32(pos)));
                break;
            }
            if (f.kind == F_VAR && !plausible_variance(pos)) {
                reject("a variance field holds an implausible value");
                break;
            }
            if (f.kind == F_KSK) {
                const size_t S4 = (size_t)(p.n + 1) * 4, rec = S4 + (f.sample_var ? 8 : 0);
                const size_t base = (size_t)p.ks_base(), per_ij = f.skip_d0 ? base - 1 : base;
                const size_t n_ij = (size_t)p.k * p.N * p.ks_t;
                // 256 positions spread over the whole array (all of it when it is that small)
                const size_t stride_ij = n_ij > 256 ? n_ij / 256 : 1;
                std::vector<int32_t> row((size_t)p.n + 1);
                for (size_t ij = 0; ij < n_ij && ok; ij += stride_ij) {
                    const size_t at = pos + ij * per_ij * rec;
                    if (!f.skip_d0 && d0_zero) {
                        // libtfhe's KS[i][j][0] encrypts 0 without noise: an all-zero sample (lweNoiselessTrivial).  A SOFT
                        // property: lweKeySwitch never reads those rows (k_ksm_prepare zeroes them whatever the file held), and a
                        // generator that encrypts h = 0 with noise writes a perfectly usable key -- so it only breaks ties
                        // between hypotheses that pass everything else (below), it does not reject one
                        bin.read(at, S4, row.data());
                        for (int32_t v : row)
                            if (v != 0) {
                                d0_zero = false;
                                break;
                            }
                    }
                    if (f.sample_var && ok)
                        for (size_t d = 0; d < per_ij && ok; d++)
                            if (!plausible_variance(at + d * rec + S4)) reject("a per-sample variance of the key-switch key holds an implausible value");
                }
                if (!ok) break;
            }
            if (f.kind == F_BK && f.sample_var) {
                const size_t row4 = (size_t)(p.k + 1) * p.N * 4, rows = (size_t)p.n * p.kpl();
                const size_t stride_r = rows > 512 ? rows / 512 : 1;
                for (size_t r = 0; r < rows && ok; r += stride_r)
                    if (!plausible_variance(pos + r * (row4 + 8) + row4)) reject("a per-row variance of the bootstrapping key holds an implausible value");
                if (!ok) break;
            }
            if (f.kind == F_LWEKEY || f.kind == F_TGSWKEY) {  // secret keys are bits
                const size_t cnt = Layout::field_bytes(f, p) / 4;
                for (size_t i = 0; i < cnt && ok; i++) {
                    const int32_t v = bin.i32(pos + 4 * i);
                    if (v != 0 && v != 1) ok = false;
                }
                if (!ok) {
                    if (why.empty()) why = "secret-key words are not bits";
                    break;
                }
            }
            pos += Layout::field_bytes(f, p);
        }
        if (ok) {
            passing.push_back(&L);
            passing_d0_zero.push_back(d0_zero ? 1 : 0);
        }
    }
    if (passing.size() > 1) {
        // tie-break: hypotheses whose d = 0 rows are all zero (libtfhe's own writer) go before those where they are not
        bool any_zero = false;
        for (char z : passing_d0_zero) any_zero = any_zero || z;
        if (any_zero) {
            std::vector<const Layout*> kept;
            for (size_t q = 0; q < passing.size(); q++)
                if (passing_d0_zero[q]) kept.push_back(passing[q]);
            passing.swap(kept);
        }
    }
    if (passing.size() > 1) {
        // Two layouts of the same size both look sound: decoding the wrong one would shift the key material by a few
        // bytes without any error.  This build's own writer (the first candidate) is identified by its two type tags
        // at fixed places; anything else is refused rather than guessed.
        bool own = passing[0] == &candidates[0];
        if (!own) {
            // hypotheses that place the key-switch key and the bootstrapping key identically differ only in the order of
            // two untagged secret-key arrays, where libtfhe's order (LWE key, then TGSW key) is the documented preference
            auto placement = [&](const Layout& L) {
                std::vector<size_t> v;
                size_t pos = 0;
                for (const Field& f : L.fields) {
                    if (f.kind == F_KSK || f.kind == F_BK) {
                        v.push_back(pos);
                        v.push_back((size_t)f.kind * 4 + (f.skip_d0 ? 2 : 0) + (f.sample_var ? 1 : 0));
                    }
                    pos += Layout::field_bytes(f, p);
                }
                return v;
            };
            const std::vector<size_t> first = placement(*passing[0]);
            own = true;
            for (const Layout* L : passing) own = own && placement(*L) == first;
        }
        if (!own) {
            std::string msg = path + ": ambiguous key layout: " + std::to_string(passing.size()) + " hypotheses fit the size, the tags and the structural checks (";
            for (size_t q = 0; q < passing.size() && q < 3; q++) msg += (q ? " | " : "") + passing[q]->desc;
            throw CodecError(msg + (passing.size() > 3 ? " | ...)" : ")"));
        }
    }
    if (!passing.empty()) chosen = passing[0];  // candidates are ordered by preference (this build's writer first)
    if (!chosen) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: no key layout fits: %zu binary bytes for n=%d N=%d k=%d l=%d t=%d basebit=%d (%zu of %zu hypotheses match the size%s%s)",
                 path.c_str(), bin.total, p.n, p.N, p.k, p.l, p.ks_t, p.ks_basebit, size_matches, candidates.size(),
                 why.empty() ? "" : "; ", why.c_str());
        throw CodecError(buf);
    }
    g_layout = chosen->desc;
    size_t pos = 0;
    for (const Field& f : chosen->fields) {
        const size_t fb = Layout::field_bytes(f, p);
        switch (f.kind) {
            case F_KSK:
                if (ck) {
                    ck->ksk.assign(p.ksk_count(), 0);
                    const size_t S = (size_t)p.n + 1, base = (size_t)p.ks_base();
                    const size_t rec = S * 4 + (f.sample_var ? 8 : 0);
                    if (!f.skip_d0 && !f.sample_var) {
                        bin.read(pos, fb, ck->ksk.data());
                    } else {
                        size_t src = pos;
                        for (size_t ij = 0; ij < (size_t)p.k * p.N * p.ks_t; ij++)
                            for (size_t d = f.skip_d0 ? 1 : 0; d < base; d++, src += rec)
                                bin.read(src, S * 4, ck->ksk.data() + (ij * base + d) * S);
                    }
                }
                break;
            case F_BK:
                if (ck) {
                    ck->bk.resize(p.bk_count());
                    const size_t row = (size_t)(p.k + 1) * p.N;
                    if (!f.sample_var) {
                        bin.read(pos, fb, ck->bk.data());
                    } else {
                        for (size_t r = 0; r < (size_t)p.n * p.kpl(); r++)
                            bin.read(pos + r * (row * 4 + 8), row * 4, ck->bk.data() + r * row);
                    }
                }
                break;
            case F_LWEKEY:
                if (sk) {
                    sk->lwe_key.resize(p.n);
                    bin.read(pos, fb, sk->lwe_key.data());
                }
                break;
            case F_TGSWKEY:
                if (sk) {
                    sk->tlwe_key.resize((size_t)p.k * p.N);
                    bin.read(pos, fb, sk->tlwe_key.data());
                }
                break;
            default:
                break;
        }
        pos += fb;
    }
}

}  // namespace

const std::string& last_key_layout() { return g_layout; }

Params load_params(const std::string& path) {
    Mapped m(path);
    return params_from_sections(scan_sections(m));
}

void load_cloud_key(const std::string& path, CloudKeyData* ck) {
    Mapped m(path);
    const std::vector<TextSection> secs = scan_sections(m);
    const Params p = params_from_sections(secs);
    const BinaryStream bin(m, secs);
    ck->p = p;
    static const std::vector<Layout> layouts = cloud_body_layouts();
    decode(path, layouts, p, bin, ck, nullptr);
}

void load_secret_key(const std::string& path, SecretKeyData* sk, bool with_cloud) {
    Mapped m(path);
    const std::vector<TextSection> secs = scan_sections(m);
    const Params p = params_from_sections(secs);
    const BinaryStream bin(m, secs);
    sk->p = p;
    sk->cloud.p = p;
    sk->cloud.bk.clear();
    sk->cloud.ksk.clear();
    static const std::vector<Layout> layouts = secret_layouts();
    decode(path, layouts, p, bin, with_cloud ? &sk->cloud : nullptr, sk);
}

void save_cloud_key(const std::string& path, const CloudKeyData& ck) {
    File f(path, "wb");
    write_cloud_key(f.f, ck);
}
void save_secret_key(const std::string& path, const SecretKeyData& sk) {
    File f(path, "wb");
    write_secret_key(f.f, sk);
}

}  // namespace ieache
