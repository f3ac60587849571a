// C ABI (include/ieache.h) over the C++ evaluator.  No exception leaves this file.
#include "../../include/ieache.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "circuit.h"
#include "cloud_run.h"
#include "codec.h"
#include "daemon.h"
#include "evaluator.h"
#include "mix_plan.h"
#include "tfhe_host.h"

using namespace ieache;

struct ieache_ctx {
    std::unique_ptr<Evaluator> eval;
    std::map<std::tuple<int, int, bool, int>, Circuit> circuits;  // (kind, bits, folded, level cap)
    std::map<std::tuple<int, int, bool, int>, uint64_t> circuit_used;  // last use of each level-capped variant (LRU order)
    uint64_t circuit_clock = 0;
    std::string variant;
    bool fold = false;           // "fold_constants"
    bool level_quantum = true;   // "level_quantum": batch-aware level widths for the slack-balanced circuits
};

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
template <class F>
int guarded(F&& f) {
    try {
        g_err.clear();
        return f();
    } catch (const CodecError& e) {
        return fail(IEACHE_EIO, e.what());
    } catch (const std::bad_alloc&) {
        return fail(IEACHE_ENOMEM, "out of host memory");
    } catch (const std::invalid_argument& e) {
        return fail(IEACHE_EINVAL, e.what());
    } catch (const std::exception& e) {
        return fail(IEACHE_ENODEV, e.what());
    } catch (...) {
        return fail(IEACHE_ENODEV, "unknown failure");
    }
}
Params to_params(const ieache_params& a) {
    Params p;
    p.n = a.n;
    p.N = a.N;
    p.k = a.k;
    p.l = a.l;
    p.Bgbit = a.Bgbit;
    p.ks_t = a.ks_t;
    p.ks_basebit = a.ks_basebit;
    p.lwe_alpha_min = a.lwe_alpha_min;
    p.lwe_alpha_max = a.lwe_alpha_max;
    p.tlwe_alpha_min = a.tlwe_alpha_min;
    p.tlwe_alpha_max = a.tlwe_alpha_max;
    return p;
}
void from_params(const Params& p, ieache_params* a) {
    a->n = p.n;
    a->N = p.N;
    a->k = p.k;
    a->l = p.l;
    a->Bgbit = p.Bgbit;
    a->ks_t = p.ks_t;
    a->ks_basebit = p.ks_basebit;
    a->lwe_alpha_min = p.lwe_alpha_min;
    a->lwe_alpha_max = p.lwe_alpha_max;
    a->tlwe_alpha_min = p.tlwe_alpha_min;
    a->tlwe_alpha_max = p.tlwe_alpha_max;
}
void to_stats(const EvalStats& s, ieache_stats* o) {
    if (!o) return;
    o->total_ms = s.total_ms;
    o->blind_rotate_ms = s.blind_rotate_ms;
    o->keyswitch_ms = s.keyswitch_ms;
    o->blind_rotate_launches = s.blind_rotate_launches;
    o->keyswitch_launches = s.keyswitch_launches;
    o->bootstraps = s.bootstraps;
    o->levels = s.levels;
    o->chunks = s.chunks;
}
// A kernel dereferencing a host or stray address faults the GPU (and can take the node's other GPUs with it), so
// the device-pointer entry points refuse anything the runtime does not know as device-accessible memory.
void require_device_pointer(const void* ptr, const char* name) {
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged && attr.type != hipMemoryTypeUnified))
        throw std::invalid_argument(std::string(name) + " is not a device pointer");
}

const Circuit* get_circuit(ieache_ctx* ctx, int kind, int bits, size_t batch = 0) {
    auto fetch = [&](int cap) -> const Circuit* {
        auto key = std::make_tuple(kind, bits, ctx->fold, cap);
        auto it = ctx->circuits.find(key);
        if (it != ctx->circuits.end()) {
            if (cap != 0) ctx->circuit_used[key] = ++ctx->circuit_clock;
            return &it->second;
        }
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, ctx->fold, cap)) return nullptr;
        if (cap != 0) {
            // Bounded: the cap follows the batch size (and the kernel family: "exact_fft" changes the resident-gate count), and
            // a variant of the wide multipliers is several MB, so at most kCappedVariants per (kind, width, folding) are kept.
            // The least recently used one goes, and only once the new circuit exists -- a caller that alternates between two
            // batch sizes or toggles exact_fft per call (bench.py's exact leg, the audit flows) rebuilds nothing.
            constexpr size_t kCappedVariants = 3;
            std::vector<std::tuple<int, int, bool, int>> mine;
            for (const auto& kv : ctx->circuits) {
                const auto& k = kv.first;
                if (std::get<0>(k) == kind && std::get<1>(k) == bits && std::get<2>(k) == ctx->fold && std::get<3>(k) != 0) mine.push_back(k);
            }
            while (mine.size() >= kCappedVariants) {
                size_t oldest = 0;
                for (size_t i = 1; i < mine.size(); i++)
                    if (ctx->circuit_used[mine[i]] < ctx->circuit_used[mine[oldest]]) oldest = i;
                ctx->circuits.erase(mine[oldest]);
                ctx->circuit_used.erase(mine[oldest]);
                mine.erase(mine.begin() + oldest);
            }
            ctx->circuit_used[key] = ++ctx->circuit_clock;
        }
        return &ctx->circuits.emplace(key, std::move(c)).first->second;
    };
    const Circuit* base = fetch(0);
    if (const char* e = getenv("IEACHE_LEVEL_CAP")) {  // measurement aid: force a level width (and with it the balanced schedule)
        const int forced = atoi(e);
        if (base && forced > 0) return fetch(forced);
    }
    if (!base || !ctx->level_quantum) return base;
    // level width chosen so that a level x this batch is a whole number of the rounds of gates the GPU holds at once
    // (same DAG, same output bits, another level assignment): the slack-balanced 64/128-bit multipliers at any batch
    // below a round, the ASAP-scheduled 32-bit multiplier family at small batches (circuit_level_cap)
    const int cap = circuit_level_cap(*base, (int64_t)batch, ctx->eval->resident_gates(), ctx->eval->resident_gates_two_wave());
    const int mean = (int)((base->n_bootstraps + base->depth - 1) / base->depth);
    if (cap <= 0 || (base->balanced_schedule && cap == mean)) return base;
    const Circuit* capped = fetch(cap);
    return capped && capped->balanced_schedule ? capped : base;
}
// the context is owned by a unique_ptr until it is handed to the caller, so a throwing key load
// (or Evaluator constructor) releases it
template <class Load>
ieache_ctx* make_ctx(const Params& p, int device, Load&& load) {
    std::unique_ptr<ieache_ctx> ctx(new ieache_ctx);
    ctx->eval.reset(new Evaluator(p, device));
    load(*ctx->eval);
    return ctx.release();
}
void fill_info(const Circuit& c, bool fold, ieache_circuit_info* out) {
    out->n_inputs = c.n_inputs;
    out->n_outputs = (int32_t)c.outputs.size();
    out->n_slots = c.n_slots;
    out->depth = c.depth;
    out->max_width = c.max_width;
    out->bootstraps = c.n_bootstraps;
    out->n_and = c.n_and;
    out->n_xor = c.n_xor;
    out->sched_max_width = c.sched_max_width;
    out->folded = fold ? 1 : 0;
    out->reference_bootstraps = c.n_reference_bootstraps;
    out->sched_levels = c.n_levels();
    out->level_cap = 0;
}
}  // namespace

extern "C" {

const char* ieache_version(void) { return "ieache-amd 0.2 (gfx950)"; }
const char* ieache_last_error(void) { return g_err.c_str(); }
const char* ieache_last_key_layout(void) { return last_key_layout().c_str(); }

int ieache_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void ieache_default_params(ieache_params* out) {
    if (out) from_params(Params{}, out);
}

int ieache_cloud_run(const char* workdir) {
    // the reference binary takes no arguments, so the GPU is chosen through the environment
    const char* dev = getenv("IEACHE_DEVICE");
    return guarded([&] { return cloud_run(workdir ? workdir : ".", nullptr, nullptr, dev ? atoi(dev) : 0); });
}

ieache_ctx* ieache_ctx_create(const char* cloud_key_path, int device) {
    ieache_ctx* ctx = nullptr;
    const int rc = guarded([&] {
        if (!cloud_key_path) return fail(IEACHE_EINVAL, "null path");
        CloudKeyData ck;
        load_cloud_key(cloud_key_path, &ck);
        ctx = make_ctx(ck.p, device, [&](Evaluator& e) { e.load_keys_host(ck.bk.data(), ck.ksk.data()); });
        return 0;
    });
    return rc == 0 ? ctx : nullptr;
}

ieache_ctx* ieache_ctx_create_raw(const ieache_params* p, const int32_t* bk, const int32_t* ksk, int device) {
    ieache_ctx* ctx = nullptr;
    const int rc = guarded([&] {
        if (!p || !bk || !ksk) return fail(IEACHE_EINVAL, "null argument");
        ctx = make_ctx(to_params(*p), device, [&](Evaluator& e) { e.load_keys_host(bk, ksk); });
        return 0;
    });
    return rc == 0 ? ctx : nullptr;
}

ieache_ctx* ieache_ctx_create_device(const ieache_params* p, const int32_t* d_bk, const int32_t* d_ksk, int device) {
    ieache_ctx* ctx = nullptr;
    const int rc = guarded([&] {
        if (!p || !d_bk || !d_ksk) return fail(IEACHE_EINVAL, "null argument");
        ctx = make_ctx(to_params(*p), device, [&](Evaluator& e) {
            require_device_pointer(d_bk, "d_bk");
            require_device_pointer(d_ksk, "d_ksk");
            e.load_keys_device(d_bk, d_ksk);
        });
        return 0;
    });
    return rc == 0 ? ctx : nullptr;
}

void ieache_ctx_destroy(ieache_ctx* ctx) {
    try {
        delete ctx;
    } catch (...) {
    }
}

int ieache_ctx_params(const ieache_ctx* ctx, ieache_params* out) {
    if (!ctx || !out) return fail(IEACHE_EINVAL, "null argument");
    from_params(ctx->eval->params(), out);
    return 0;
}

int ieache_lwe_stride(const ieache_ctx* ctx) { return ctx ? ctx->eval->params().lwe_stride() : IEACHE_EINVAL; }

void* ieache_ctx_stream(const ieache_ctx* ctx) { return ctx ? (void*)ctx->eval->stream() : nullptr; }

int ieache_ctx_cloud_run(ieache_ctx* ctx, const char* workdir) {
    if (!ctx) return fail(IEACHE_EINVAL, "null context");
    return guarded([&] { return cloud_run(workdir ? workdir : ".", ctx->eval.get(), nullptr, ctx->eval->device()); });
}

int ieache_ctx_set_chunk(ieache_ctx* ctx, int64_t items) {
    if (!ctx || items < 1) return fail(IEACHE_EINVAL, "bad chunk");
    ctx->eval->set_chunk((size_t)items);
    return 0;
}

int ieache_ctx_force_generic(ieache_ctx* ctx, int on) {
    if (!ctx) return fail(IEACHE_EINVAL, "null context");
    ctx->eval->set_force_generic(on != 0);
    return 0;
}

int ieache_ctx_wait_stream(ieache_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(IEACHE_EINVAL, "null context");
    return guarded([&] {
        ctx->eval->wait_for_stream((hipStream_t)hip_stream);
        return 0;
    });
}

int ieache_ctx_set_option(ieache_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return fail(IEACHE_EINVAL, "null argument");
    if (std::string(name) == "level_quantum") {
        if (value != 0 && value != 1) return fail(IEACHE_EINVAL, "level_quantum takes 0 or 1");
        ctx->level_quantum = value != 0;
        return 0;
    }
    if (std::string(name) == "fold_constants") {
        if (value != 0 && value != 1) return fail(IEACHE_EINVAL, "fold_constants takes 0 or 1");
        ctx->fold = value != 0;
        return 0;
    }
    if (!ctx->eval->set_option(name, value)) return fail(IEACHE_EINVAL, std::string("unknown option or bad value: ") + name);
    return 0;
}

int ieache_ctx_fft_guard(const ieache_ctx* ctx, double* max_deviation, int64_t* reruns) {
    if (!ctx) return fail(IEACHE_EINVAL, "null context");
    if (max_deviation) *max_deviation = ctx->eval->fft_guard_max();
    if (reruns) *reruns = ctx->eval->fft_guard_reruns();
    return 0;
}

int ieache_ctx_get_option(const ieache_ctx* ctx, const char* name, int64_t* value) {
    if (!ctx || !name) return fail(IEACHE_EINVAL, "null argument");
    if (std::string(name) == "level_quantum") {
        if (value) *value = ctx->level_quantum ? 1 : 0;
        return 0;
    }
    if (std::string(name) == "fold_constants") {
        if (value) *value = ctx->fold ? 1 : 0;
        return 0;
    }
    if (!ctx->eval->get_option(name, value)) return fail(IEACHE_EINVAL, std::string("unknown option: ") + name);
    return 0;
}

int ieache_ctx_fft_audit(const ieache_ctx* ctx, int64_t* audits, int64_t* gates_compared, int64_t* mismatches) {
    if (!ctx) return fail(IEACHE_EINVAL, "null context");
    ctx->eval->fft_audit_counts(audits, gates_compared, mismatches);
    return 0;
}

const char* ieache_ctx_kernel_variant(const ieache_ctx* ctx) {
    if (!ctx) return "";
    const_cast<ieache_ctx*>(ctx)->variant = ctx->eval->kernel_variant();
    return ctx->variant.c_str();
}

const char* ieache_ctx_kernel_for_launch(const ieache_ctx* ctx, int64_t gates) {
    if (!ctx) return "";
    const_cast<ieache_ctx*>(ctx)->variant = ctx->eval->kernel_for_launch(gates);
    return ctx->variant.c_str();
}

int ieache_circuit_info_get_ex(int kind, int bits, int fold_constants, ieache_circuit_info* out) {
    return guarded([&] {
        if (!out) return fail(IEACHE_EINVAL, "null argument");
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, fold_constants != 0)) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        fill_info(c, fold_constants != 0, out);
        return 0;
    });
}
// The 0.1 entry point: its callers were compiled against the 56-byte struct (through `folded`), so it writes exactly that
// prefix.  The fields added since are reached through the _ex / _cap entry points, which take the current struct.
int ieache_circuit_info_get(int kind, int bits, ieache_circuit_info* out) {
    if (!out) return fail(IEACHE_EINVAL, "null argument");
    ieache_circuit_info full;
    memset(&full, 0, sizeof full);  // padding included: the prefix is copied bytewise
    const int rc = ieache_circuit_info_get_ex(kind, bits, 0, &full);
    if (rc == 0) memcpy(out, &full, IEACHE_CIRCUIT_INFO_V01_BYTES);
    return rc;
}

int ieache_ctx_circuit_level_cap(const ieache_ctx* ctx, int kind, int bits, int64_t batch) {
    return guarded([&] {
        if (!ctx) return fail(IEACHE_EINVAL, "null context");
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, ctx->fold)) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        return (int)circuit_level_cap(c, batch, ctx->eval->resident_gates(), ctx->eval->resident_gates_two_wave());
    });
}

int ieache_circuit_level_cap(int kind, int bits, int fold_constants, int64_t batch, int resident_workgroups) {
    return guarded([&] {
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, fold_constants != 0)) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        return (int)circuit_level_cap(c, batch, resident_workgroups);
    });
}

int ieache_circuit_info_get_cap(int kind, int bits, int fold_constants, int level_cap, ieache_circuit_info* out) {
    return guarded([&] {
        if (!out || level_cap < 0) return fail(IEACHE_EINVAL, "bad argument");
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, fold_constants != 0, level_cap)) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        fill_info(c, fold_constants != 0, out);
        out->level_cap = c.balanced_schedule ? level_cap : 0;
        return 0;
    });
}

int ieache_circuit_simulate_cap(int kind, int bits, int fold_constants, int level_cap, const uint8_t* in_bits, uint8_t* out_bits) {
    return guarded([&] {
        if (!in_bits || !out_bits || level_cap < 0) return fail(IEACHE_EINVAL, "bad argument");
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, fold_constants != 0, level_cap)) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        simulate_circuit(c, in_bits, out_bits);
        return 0;
    });
}

int ieache_circuit_simulate_ex(int kind, int bits, int fold_constants, const uint8_t* in_bits, uint8_t* out_bits) {
    return guarded([&] {
        if (!in_bits || !out_bits) return fail(IEACHE_EINVAL, "null argument");
        Circuit c;
        if (!build_circuit(kind, bits, &c, true, fold_constants != 0)) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        simulate_circuit(c, in_bits, out_bits);
        return 0;
    });
}
int ieache_circuit_simulate(int kind, int bits, const uint8_t* in_bits, uint8_t* out_bits) {
    return ieache_circuit_simulate_ex(kind, bits, 0, in_bits, out_bits);
}

int ieache_eval_batch(ieache_ctx* ctx, int kind, int bits, size_t batch, const int32_t* in_lwe, int32_t* out_lwe,
                      ieache_stats* stats) {
    return guarded([&] {
        if (!ctx || !in_lwe || !out_lwe) return fail(IEACHE_EINVAL, "null argument");
        const Circuit* c = get_circuit(ctx, kind, bits, batch);
        if (!c) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        EvalStats st;
        eval_circuit_host(*ctx->eval, *c, batch, in_lwe, out_lwe, stats ? &st : nullptr);
        to_stats(st, stats);
        return 0;
    });
}

int ieache_eval_batch_device(ieache_ctx* ctx, int kind, int bits, size_t batch, const int32_t* d_in, int32_t* d_out,
                             ieache_stats* stats) {
    return guarded([&] {
        if (!ctx || !d_in || !d_out) return fail(IEACHE_EINVAL, "null argument");
        const Circuit* c = get_circuit(ctx, kind, bits, batch);
        if (!c) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        if (batch) {
            require_device_pointer(d_in, "d_in");
            require_device_pointer(d_out, "d_out");
        }
        EvalStats st;
        ctx->eval->eval_circuit_device(*c, batch, d_in, d_out, stats ? &st : nullptr);
        to_stats(st, stats);
        return 0;
    });
}

int ieache_prepare_batch(ieache_ctx* ctx, int kind, int bits, size_t batch) {
    return guarded([&] {
        if (!ctx) return fail(IEACHE_EINVAL, "null argument");
        const Circuit* c = get_circuit(ctx, kind, bits, batch);
        if (!c) return fail(IEACHE_EINVAL, "unsupported circuit kind/bits");
        ctx->eval->prepare_circuit(*c, batch);
        return 0;
    });
}

int ieache_gates_device(ieache_ctx* ctx, int gate_type, size_t count, const int32_t* d_a, const int32_t* d_b,
                        int32_t* d_out, ieache_stats* stats) {
    return guarded([&] {
        if (!ctx || !d_a || !d_b || !d_out) return fail(IEACHE_EINVAL, "null argument");
        if (gate_type < 0 || gate_type > 3) return fail(IEACHE_EINVAL, "unknown gate type");
        if (count) {
            require_device_pointer(d_a, "d_a");
            require_device_pointer(d_b, "d_b");
            require_device_pointer(d_out, "d_out");
        }
        EvalStats st;
        ctx->eval->gates_device(gate_type, count, d_a, d_b, d_out, stats ? &st : nullptr);
        to_stats(st, stats);
        return 0;
    });
}

namespace {
// host rows (n+1) <-> device rows (stride)
struct DevRows {
    Torus32* p = nullptr;
    size_t rows, stride;
    DevRows(size_t r, size_t s) : rows(r), stride(s) {
        HIP_CHECK(hipMalloc(&p, rows * stride * 4 + 16));
        HIP_CHECK(hipMemset(p, 0, rows * stride * 4 + 16));
    }
    ~DevRows() { (void)hipFree(p); }
    void upload(const int32_t* h, size_t width) {
        HIP_CHECK(hipMemcpy2D(p, stride * 4, h, width * 4, width * 4, rows, hipMemcpyHostToDevice));
    }
    void download(int32_t* h, size_t width) {
        HIP_CHECK(hipMemcpy2D(h, width * 4, p, stride * 4, width * 4, rows, hipMemcpyDeviceToHost));
    }
};
// the same on the evaluator's kept staging rows (Evaluator::staging): what the gate / MUX host entry points use, so that
// a warm call allocates nothing
struct StagedRows {
    Torus32* p;
    size_t rows, stride;
    StagedRows(Evaluator& ev, int slot, size_t r, size_t s) : p(ev.staging(slot, r * s * 4)), rows(r), stride(s) {}
    void upload(const int32_t* h, size_t width) {
        if (rows) HIP_CHECK(hipMemcpy2D(p, stride * 4, h, width * 4, width * 4, rows, hipMemcpyHostToDevice));
    }
    void download(int32_t* h, size_t width) {
        if (rows) HIP_CHECK(hipMemcpy2D(h, width * 4, p, stride * 4, width * 4, rows, hipMemcpyDeviceToHost));
    }
};
}  // namespace

int ieache_gates(ieache_ctx* ctx, int gate_type, size_t count, const int32_t* a, const int32_t* b, int32_t* out,
                 ieache_stats* stats) {
    return guarded([&] {
        if (!ctx || !a || !b || !out) return fail(IEACHE_EINVAL, "null argument");
        if (gate_type < 0 || gate_type > 3) return fail(IEACHE_EINVAL, "unknown gate type");
        const Params& p = ctx->eval->params();
        HIP_CHECK(hipSetDevice(ctx->eval->device()));
        StagedRows da(*ctx->eval, 0, count, p.lwe_stride()), db(*ctx->eval, 1, count, p.lwe_stride()), dout(*ctx->eval, 3, count, p.lwe_stride());
        da.upload(a, p.n + 1);
        db.upload(b, p.n + 1);
        EvalStats st;
        ctx->eval->gates_device(gate_type, count, da.p, db.p, dout.p, stats ? &st : nullptr);
        dout.download(out, p.n + 1);
        to_stats(st, stats);
        return 0;
    });
}

int ieache_mux_device(ieache_ctx* ctx, size_t count, const int32_t* d_a, const int32_t* d_b, const int32_t* d_c,
                      int32_t* d_out, ieache_stats* stats) {
    return guarded([&] {
        if (!ctx || !d_a || !d_b || !d_c || !d_out) return fail(IEACHE_EINVAL, "null argument");
        if (count) {
            require_device_pointer(d_a, "d_a");
            require_device_pointer(d_b, "d_b");
            require_device_pointer(d_c, "d_c");
            require_device_pointer(d_out, "d_out");
        }
        EvalStats st;
        ctx->eval->mux_device(count, d_a, d_b, d_c, d_out, stats ? &st : nullptr);
        to_stats(st, stats);
        return 0;
    });
}

int ieache_mux(ieache_ctx* ctx, size_t count, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* out,
               ieache_stats* stats) {
    return guarded([&] {
        if (!ctx || !a || !b || !c || !out) return fail(IEACHE_EINVAL, "null argument");
        const Params& p = ctx->eval->params();
        HIP_CHECK(hipSetDevice(ctx->eval->device()));
        StagedRows da(*ctx->eval, 0, count, p.lwe_stride()), db(*ctx->eval, 1, count, p.lwe_stride()), dc(*ctx->eval, 2, count, p.lwe_stride()),
            dout(*ctx->eval, 3, count, p.lwe_stride());
        da.upload(a, p.n + 1);
        db.upload(b, p.n + 1);
        dc.upload(c, p.n + 1);
        EvalStats st;
        ctx->eval->mux_device(count, da.p, db.p, dc.p, dout.p, stats ? &st : nullptr);
        dout.download(out, p.n + 1);
        to_stats(st, stats);
        return 0;
    });
}

int ieache_debug_blind_rotate(ieache_ctx* ctx, size_t count, const int32_t* x, int32_t* acc, int32_t steps) {
    return guarded([&] {
        if (!ctx || !x || !acc) return fail(IEACHE_EINVAL, "null argument");
        const Params& p = ctx->eval->params();
        HIP_CHECK(hipSetDevice(ctx->eval->device()));
        DevRows dx(count, p.lwe_stride()), dacc(count, (size_t)2 * p.N);
        dx.upload(x, p.n + 1);
        ctx->eval->debug_blind_rotate(count, dx.p, dacc.p, steps);
        dacc.download(acc, (size_t)2 * p.N);
        return 0;
    });
}

int ieache_debug_keyswitch(ieache_ctx* ctx, size_t count, const int32_t* u, int32_t* out) {
    return guarded([&] {
        if (!ctx || !u || !out) return fail(IEACHE_EINVAL, "null argument");
        const Params& p = ctx->eval->params();
        HIP_CHECK(hipSetDevice(ctx->eval->device()));
        DevRows du(count, (size_t)p.N + 1), dout(count, p.lwe_stride());
        du.upload(u, (size_t)p.N + 1);
        ctx->eval->debug_keyswitch(count, du.p, dout.p);
        dout.download(out, p.n + 1);
        return 0;
    });
}

// ---------------- CPU tools ----------------
int ieache_keygen_raw(const ieache_params* p, const uint32_t* seed_words, int n_seed_words, int32_t* lwe_key,
                      int32_t* tlwe_key, int32_t* bk, int32_t* ksk) {
    return guarded([&] {
        if (!p) return fail(IEACHE_EINVAL, "null params");
        const Params pp = to_params(*p);
        if (!pp.supported()) return fail(IEACHE_EINVAL, "unsupported parameter set");
        SecretKeyData sk;
        keygen(pp, seed_words, n_seed_words, &sk, bk != nullptr || ksk != nullptr);
        if (lwe_key) memcpy(lwe_key, sk.lwe_key.data(), sk.lwe_key.size() * 4);
        if (tlwe_key) memcpy(tlwe_key, sk.tlwe_key.data(), sk.tlwe_key.size() * 4);
        if (bk) memcpy(bk, sk.cloud.bk.data(), sk.cloud.bk.size() * 4);
        if (ksk) memcpy(ksk, sk.cloud.ksk.data(), sk.cloud.ksk.size() * 4);
        return 0;
    });
}

int ieache_keygen_files(const char* dir, const ieache_params* p, const uint32_t* seed, int n_seed,
                        const uint32_t* nbit_seed, int n_nbit_seed) {
    return guarded([&] {
        const std::string d = dir ? dir : ".";
        Params pp;
        if (p) pp = to_params(*p);
        if (!pp.supported()) return fail(IEACHE_EINVAL, "unsupported parameter set");
        static const uint32_t kSeed[3] = {314, 1592, 657}, kBitSeed[3] = {314, 1592, 888};  // keygen.c:30,34
        if (!seed && n_seed >= 0) {
            seed = kSeed;
            n_seed = 3;
        }
        if (!nbit_seed && n_nbit_seed >= 0) {
            nbit_seed = kBitSeed;
            n_nbit_seed = 3;
        }
        SecretKeyData key, nbit;
        keygen(pp, seed, n_seed, &key, true);
        save_secret_key(d + "/secret.key", key);   // keygen.c:38-40
        save_cloud_key(d + "/cloud.key", key.cloud);  // :43-45
        keygen(pp, nbit_seed, n_nbit_seed, &nbit, true);
        save_secret_key(d + "/nbit.key", nbit);  // :48-50
        return 0;
    });
}

int ieache_encrypt_bits(const ieache_params* p, const int32_t* lwe_key, const uint8_t* bits, size_t count,
                        uint64_t seed, int32_t* out) {
    return guarded([&] {
        if (!p || !lwe_key || !bits || !out) return fail(IEACHE_EINVAL, "null argument");
        const Params pp = to_params(*p);
        Rng rng = seed ? Rng(seed) : Rng::secure();
        for (size_t i = 0; i < count; i++) lwe_encrypt_bit(pp, lwe_key, bits[i] & 1, rng, out + i * (size_t)(pp.n + 1));
        return 0;
    });
}

int ieache_decrypt_bits(const ieache_params* p, const int32_t* lwe_key, const int32_t* samples, size_t count,
                        uint8_t* bits) {
    return guarded([&] {
        if (!p || !lwe_key || !samples || !bits) return fail(IEACHE_EINVAL, "null argument");
        const Params pp = to_params(*p);
        for (size_t i = 0; i < count; i++) bits[i] = (uint8_t)lwe_decrypt_bit(pp, lwe_key, samples + i * (size_t)(pp.n + 1));
        return 0;
    });
}

int ieache_read_secret_key(const char* path, ieache_params* p, int32_t* lwe_key, int32_t* tlwe_key) {
    return guarded([&] {
        if (!path) return fail(IEACHE_EINVAL, "null path");
        SecretKeyData sk;
        load_secret_key(path, &sk, false);
        if (p) from_params(sk.p, p);
        if (lwe_key) memcpy(lwe_key, sk.lwe_key.data(), sk.lwe_key.size() * 4);
        if (tlwe_key) memcpy(tlwe_key, sk.tlwe_key.data(), sk.tlwe_key.size() * 4);
        return 0;
    });
}

int ieache_read_cloud_key(const char* path, ieache_params* p, int32_t* bk, int32_t* ksk) {
    return guarded([&] {
        if (!path) return fail(IEACHE_EINVAL, "null path");
        if (!bk && !ksk) {
            const Params pp = load_params(path);
            if (p) from_params(pp, p);
            return 0;
        }
        CloudKeyData ck;
        load_cloud_key(path, &ck);
        if (p) from_params(ck.p, p);
        if (bk) memcpy(bk, ck.bk.data(), ck.bk.size() * 4);
        if (ksk) memcpy(ksk, ck.ksk.data(), ck.ksk.size() * 4);
        return 0;
    });
}

int ieache_write_cloud_key(const char* path, const ieache_params* p, const int32_t* bk, const int32_t* ksk) {
    return guarded([&] {
        if (!path || !p || !bk || !ksk) return fail(IEACHE_EINVAL, "null argument");
        CloudKeyData ck;
        ck.p = to_params(*p);
        ck.bk.assign(bk, bk + ck.p.bk_count());
        ck.ksk.assign(ksk, ksk + ck.p.ksk_count());
        save_cloud_key(path, ck);
        return 0;
    });
}

int ieache_write_secret_key(const char* path, const ieache_params* p, const int32_t* lwe_key, const int32_t* tlwe_key,
                            const int32_t* bk, const int32_t* ksk) {
    return guarded([&] {
        if (!path || !p || !lwe_key || !tlwe_key || !bk || !ksk) return fail(IEACHE_EINVAL, "null argument");
        SecretKeyData sk;
        sk.p = to_params(*p);
        sk.lwe_key.assign(lwe_key, lwe_key + sk.p.n);
        sk.tlwe_key.assign(tlwe_key, tlwe_key + (size_t)sk.p.k * sk.p.N);
        sk.cloud.p = sk.p;
        sk.cloud.bk.assign(bk, bk + sk.p.bk_count());
        sk.cloud.ksk.assign(ksk, ksk + sk.p.ksk_count());
        save_secret_key(path, sk);
        return 0;
    });
}

int ieache_read_samples(const char* path, int32_t n, size_t first, size_t count, int32_t* out) {
    return guarded([&] {
        if (!path || !out || n < 1) return fail(IEACHE_EINVAL, "bad argument");
        FILE* f = fopen(path, "rb");
        if (!f) throw CodecError(std::string("cannot open ") + path);
        try {
            if (fseek(f, (long)(first * lwe_sample_bytes(n)), SEEK_SET) != 0) throw CodecError("seek failed");
            read_lwe_samples(f, n, count, out);
        } catch (...) {
            fclose(f);
            throw;
        }
        fclose(f);
        return 0;
    });
}

int ieache_write_samples(const char* path, int32_t n, size_t count, const int32_t* rows, int append) {
    return guarded([&] {
        if (!path || !rows || n < 1) return fail(IEACHE_EINVAL, "bad argument");
        FILE* f = fopen(path, append ? "ab" : "wb");
        if (!f) throw CodecError(std::string("cannot open ") + path);
        try {
            write_lwe_samples(f, n, count, rows, (size_t)n + 1);
        } catch (...) {
            fclose(f);
            throw;
        }
        fclose(f);
        return 0;
    });
}

int ieache_alice(const char* secret_key_path, const char* nbit_key_path, const char* cloud_data_path, int append,
                 uint32_t sign_code, uint32_t bit_size, const uint32_t* words, uint64_t seed) {
    return guarded([&] {
        if (!secret_key_path || !nbit_key_path || !cloud_data_path || !words) return fail(IEACHE_EINVAL, "null argument");
        SecretKeyData key, nbit;
        load_secret_key(secret_key_path, &key, false);
        load_secret_key(nbit_key_path, &nbit, false);
        if (key.p.n != nbit.p.n) throw CodecError("secret.key and nbit.key disagree on n");
        const size_t S = (size_t)key.p.n + 1;
        std::vector<Torus32> rows(352 * S);
        Rng rng = seed ? Rng(seed) : Rng::secure();
        auto enc_word = [&](const SecretKeyData& k, uint32_t v, size_t word_index) {
            for (int i = 0; i < 32; i++)  // alice.c:123-125: bit i of the word is sample i
                lwe_encrypt_bit(k.p, k.lwe_key.data(), (v >> i) & 1, rng, rows.data() + (word_index * 32 + i) * S);
        };
        enc_word(nbit, sign_code, 0);                            // alice.c:116-118
        enc_word(nbit, bit_size, 1);                             // :120-122
        for (int w = 0; w < 8; w++) enc_word(key, words[w], 2 + w);  // :123-146
        enc_word(key, 0, 10);                                    // :147-149 carry = 0
        FILE* f = fopen(cloud_data_path, append ? "ab" : "wb");
        if (!f) throw CodecError(std::string("cannot open ") + cloud_data_path);
        try {
            write_lwe_samples(f, key.p.n, 352, rows.data(), S);  // alice.c:167-191
        } catch (...) {
            fclose(f);
            throw;
        }
        fclose(f);
        return 0;
    });
}

int ieache_verif(const char* secret_key_path, const char* nbit_key_path, const char* answer_data_path,
                 uint32_t* sign_code, uint32_t* bit_size, uint32_t* words9) {
    return guarded([&] {
        if (!secret_key_path || !nbit_key_path || !answer_data_path) return fail(IEACHE_EINVAL, "null argument");
        SecretKeyData key, nbit;
        load_secret_key(secret_key_path, &key, false);
        load_secret_key(nbit_key_path, &nbit, false);
        const size_t S = (size_t)key.p.n + 1;
        std::vector<Torus32> rows(352 * S);
        FILE* f = fopen(answer_data_path, "rb");
        if (!f) throw CodecError(std::string("cannot open ") + answer_data_path);
        try {
            read_lwe_samples(f, key.p.n, 352, rows.data());
        } catch (...) {
            fclose(f);
            throw;
        }
        fclose(f);
        auto dec_word = [&](const SecretKeyData& k, size_t word_index) {
            uint32_t v = 0;
            for (int i = 0; i < 32; i++)  // verif.c:57-60, 92-95
                v |= (uint32_t)lwe_decrypt_bit(k.p, k.lwe_key.data(), rows.data() + (word_index * 32 + i) * S) << i;
            return v;
        };
        if (sign_code) *sign_code = dec_word(nbit, 0);
        if (bit_size) *bit_size = dec_word(nbit, 1);
        if (words9)
            for (int w = 0; w < 9; w++) words9[w] = dec_word(key, 2 + w);
        return 0;
    });
}

// ---- 5. resident-key daemon ----
int64_t ieache_serve(const char* socket_path, const char* cloud_key_path, const char* nbit_key_path, int device,
                     int64_t max_requests) {
    return ieache_serve_devices(socket_path, cloud_key_path, nbit_key_path, &device, 1, max_requests);
}

int ieache_debug_mix_plan(int cus, int n, int64_t gates, int s1, int ratio_x100, int out[9]) {
    if (!out) return fail(IEACHE_EINVAL, "null argument");
    MixGeometry g;
    MixSteps m;
    for (int i = 0; i < 9; i++) out[i] = 0;
    if (!mix_geometry_for(cus, gates, 0, 0, &g) || !mix_steps_for(n, g, s1, ratio_x100, &m)) return 0;  // no rotation at this size
    const int v[9] = {g.k, g.tw, m.s1, m.s2, m.cycles, m.tail_s1, m.tail_s2, m.covered, (int)mix_subset_size(gates, g.k)};
    for (int i = 0; i < 9; i++) out[i] = v[i];
    return 1;
}

int ieache_shard_slice(size_t total, size_t parts, size_t part, size_t* first, size_t* count) {
    if (!first || !count || parts == 0 || part >= parts) return fail(IEACHE_EINVAL, "bad shard arguments");
    daemon_shard(total, parts, part, first, count);
    return 0;
}

int64_t ieache_serve_devices(const char* socket_path, const char* cloud_key_path, const char* nbit_key_path, const int* devices,
                             int n_devices, int64_t max_requests) {
    int64_t served = 0;
    const int rc = guarded([&] {
        if (!socket_path || !cloud_key_path || !devices) return fail(IEACHE_EINVAL, "null argument");
        if (n_devices < 1 || n_devices > 64) return fail(IEACHE_EINVAL, "1 .. 64 devices");
        DaemonConfig cfg;
        cfg.socket_path = socket_path;
        cfg.cloud_key_path = cloud_key_path;
        if (nbit_key_path) cfg.nbit_key_path = nbit_key_path;
        cfg.device = devices[0];
        cfg.devices.assign(devices, devices + n_devices);
        cfg.max_requests = max_requests;
        if (const char* w = getenv("IEACHE_DAEMON_BATCH_WINDOW_MS")) cfg.batch_window_ms = atoi(w) > 0 ? atoi(w) : 0;
        if (const char* m = getenv("IEACHE_DAEMON_MAX_BATCH")) cfg.max_batch = atoi(m) > 0 ? atoi(m) : 1;
        served = daemon_serve(cfg);
        return 0;
    });
    return rc != 0 ? rc : served;
}

static int client_call(const char* socket_path, uint32_t op, const void* payload, size_t len, DaemonReply* reply) {
    return guarded([&] {
        if (!socket_path) return fail(IEACHE_EINVAL, "null socket path");
        *reply = daemon_request(socket_path, op, payload, len);
        if (reply->rc < 0) g_err = reply->log;  // the daemon's message for IEACHE_E*
        return (int)reply->rc;
    });
}

int ieache_client_ping(const char* socket_path) {
    DaemonReply r;
    return client_call(socket_path, DAEMON_PING, nullptr, 0, &r);
}

int ieache_client_run_dir(const char* socket_path, const char* workdir) {
    if (!workdir) return fail(IEACHE_EINVAL, "null workdir");
    DaemonReply r;
    const int rc = client_call(socket_path, DAEMON_RUN_DIR, workdir, strlen(workdir), &r);
    if (rc >= 0) fputs(r.log.c_str(), stdout);  // the chatter main() of cloud.c prints
    return rc;
}

int ieache_client_run_data(const char* socket_path, int operator_code, const void* cloud_data, size_t cloud_data_len,
                           void* answer, size_t answer_cap, size_t* answer_len) {
    if (!cloud_data && cloud_data_len) return fail(IEACHE_EINVAL, "null cloud.data");
    DaemonReply r;
    std::vector<unsigned char> payload;
    const int prc = guarded([&] {
        payload.resize(4 + cloud_data_len);
        const int32_t op = operator_code;
        memcpy(payload.data(), &op, 4);
        if (cloud_data_len) memcpy(payload.data() + 4, cloud_data, cloud_data_len);
        return 0;
    });
    if (prc != 0) return prc;
    const int rc = client_call(socket_path, DAEMON_RUN_DATA, payload.data(), payload.size(), &r);
    if (rc < 0) return rc;
    if (answer_len) *answer_len = r.data.size();
    if (r.data.size() > answer_cap || (!answer && !r.data.empty())) return fail(IEACHE_EINVAL, "answer buffer too small");
    if (!r.data.empty()) memcpy(answer, r.data.data(), r.data.size());
    return rc;
}

int ieache_client_shutdown(const char* socket_path) {
    DaemonReply r;
    return client_call(socket_path, DAEMON_SHUTDOWN, nullptr, 0, &r);
}

}  // extern "C"
