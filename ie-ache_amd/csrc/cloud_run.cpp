// The `./cloud` process contract of the reference, as a function.
//
// Mirrors main() of /root/reference/Cloud/cloud.c:650-2720 step by step: same
// files in the working directory, same metadata arithmetic, same dispatch on
// (operator, sign case, bit size), same answer.data layout, same exit codes.
// The only thing that changes is HOW the gates are evaluated: the selected
// circuit runs level-batched on the GPU through ieache::Evaluator instead of
// one libtfhe bootstrap at a time.
#include "cloud_run.h"

#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>

#include <sys/time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <random>
#include <vector>

#include "circuit.h"
#include "codec.h"
#include "evaluator.h"
#include "tfhe_host.h"

namespace ieache {

namespace {
struct FileCloser {
    FILE* f;
    ~FileCloser() {
        if (f) fclose(f);
    }
};
double now_s() {
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}
int32_t decrypt_word(const Params& p, const int32_t* key, const Torus32* samples) {
    int32_t v = 0;
    for (int i = 0; i < 32; i++) v |= lwe_decrypt_bit(p, key, samples + (size_t)i * (p.n + 1)) << i;  // cloud.c:710-713
    return v;
}
}  // namespace

CloudDirSession::CloudDirSession(const std::string& dir_in, const Params& cloud_params, FILE* log)
    : dir(dir_in.empty() ? std::string(".") : dir_in) {
    auto path = [&](const char* name) { return dir + "/" + name; };
    load_secret_key(path("nbit.key"), &nbit, /*with_cloud=*/false);  // cloud.c:661-663
    data = fopen(path("cloud.data").c_str(), "rb");                   // cloud.c:703-705
    if (!data) throw CodecError("cannot open cloud.data");
    int32_t int_op = 0;  // cloud.c:769-773
    {
        std::ifstream in(path("operator.txt"));
        in >> int_op;
    }
    io.params = cloud_params;
    io.nbit = &nbit;
    io.cloud_data = data;
    io.op = int_op;
    io.open_answer = [this]() -> FILE* {  // cloud.c:809, after the inputs have been read
        answer = fopen((dir + "/answer.data").c_str(), "wb");
        if (!answer) throw CodecError("cannot create answer.data");
        return answer;
    };
    io.stats_path = path("averagestandard.txt");
    io.log = log;
}

CloudDirSession::~CloudDirSession() {
    if (data) fclose(data);
    if (answer) fclose(answer);
}

int cloud_run(const std::string& dir, Evaluator* shared_eval, CloudRunReport* report, int device, FILE* log) {
    if (!log) log = stdout;
    const std::string d = dir.empty() ? std::string(".") : dir;
    // IEACHE_TIMING=1: where a cold ./cloud spends its time (stderr), phase by phase
    static const bool timing = getenv("IEACHE_TIMING") && atoi(getenv("IEACHE_TIMING")) != 0;
    double t_last = now_s();
    auto tick = [&](const char* what) {
        if (!timing) return;
        const double t = now_s();
        fprintf(stderr, "[cloud-timing] %-44s %8.1f ms\n", what, (t - t_last) * 1e3);
        t_last = t;
    };
    fprintf(log, "Reading the key...\n");
    // cloud.c:656-663
    CloudKeyData ck;
    std::unique_ptr<Evaluator> owned;
    // A cold process (the reference starts ./cloud once per operator) spends ~150 ms bringing up the HIP runtime and
    // ~45 ms reading and decoding the 114 MB key file: do the two at the same time.
    std::thread warm;
    if (!shared_eval) {
        warm = std::thread([device] {
            if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr);  // forces runtime + context creation; errors resurface in Evaluator()
        });
        try {
            load_cloud_key(d + "/cloud.key", &ck);
        } catch (...) {
            warm.join();
            throw;
        }
    }
    struct Joiner {
        std::thread& t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{warm};
    tick("cloud.key read and decoded");
    CloudDirSession session(d, shared_eval ? shared_eval->params() : ck.p, log);
    tick("nbit.key, cloud.data, operator.txt opened");
    const int rc = cloud_run_io(session.io, [&]() -> Evaluator* {
        if (shared_eval) return shared_eval;
        tick("inputs read, metadata words written");
        if (warm.joinable()) warm.join();
        owned.reset(new Evaluator(ck.p, device));
        tick("HIP runtime + evaluator created");
        owned->load_keys_host(ck.bk.data(), ck.ksk.data());
        tick("key uploaded, spectra / limb matrix built");
        return owned.get();
    }, report);
    tick("circuit evaluated, answer.data written");
    return rc;
}

// Everything main() does before its circuit: inputs, metadata arithmetic in the clear, the 64 metadata samples of
// answer.data, and the choice of branch.  job->has_circuit says whether a value circuit is to be evaluated;
// otherwise job->rc (0 or 126) is the whole answer (64-sample answer.data).
void cloud_prepare(const CloudRunIO& io, CloudJob* job, CloudRunReport* report) {
    FILE* const log = io.log ? io.log : stdout;
    const SecretKeyData& nbit = *io.nbit;
    const Params p = io.params;
    const Params& np = nbit.p;
    if (np.n != p.n) throw CodecError("nbit.key and cloud.key disagree on the LWE dimension");
    const int32_t n = p.n;
    const size_t S = (size_t)n + 1, WORD = 32 * S;

    // cloud.c:703-766: 22 arrays of 32 samples
    fprintf(log, "Reading input 1...\n");
    std::vector<Torus32> data(22 * WORD);
    read_lwe_samples(io.cloud_data, n, 22 * 32, data.data());
    fprintf(log, "Reading input 2...\n");
    const Torus32* neg1 = data.data();
    const Torus32* bit1 = data.data() + WORD;
    const Torus32* opnd1 = data.data() + 2 * WORD;    // ciphertext1..8
    const Torus32* carry1 = data.data() + 10 * WORD;  // ciphertextcarry1
    const Torus32* neg2 = data.data() + 11 * WORD;
    const Torus32* bit2 = data.data() + 12 * WORD;
    const Torus32* opnd2 = data.data() + 13 * WORD;   // ciphertext9..16
    const int32_t int_bit1 = decrypt_word(np, nbit.lwe_key.data(), bit1);
    const int32_t int_bit2 = decrypt_word(np, nbit.lwe_key.data(), bit2);

    fprintf(log, "Reading operation code...\n");
    const int32_t int_op = io.op;
    int32_t int_negative1 = decrypt_word(np, nbit.lwe_key.data(), neg1);  // :780-785
    fprintf(log, "%d => negative1\n", int_negative1);
    if (int_negative1 == 2) int_negative1 = 1;  // :787-789
    const int32_t int_negative2 = decrypt_word(np, nbit.lwe_key.data(), neg2);
    fprintf(log, "%d => negative2\n", int_negative2);
    const int32_t int_negative = int_negative1 + int_negative2;  // :804

    struct { FILE* f; } ans{io.open_answer()};  // :809
    if (!ans.f) throw CodecError("no answer sink");
    job->answer = ans.f;
    job->params = p;
    job->rc = 0;
    job->has_circuit = false;
    int32_t ciphernegative = 0;  // :812-821
    if (int_negative == 1) ciphernegative = 1;
    if (int_negative == 2) ciphernegative = 2;
    if (int_negative == 3) ciphernegative = 4;
    Rng rng = Rng::secure();  // these samples leave the process: ChaCha20 keyed from the kernel
    std::vector<Torus32> word(WORD);
    for (int i = 0; i < 32; i++)  // :822-824 fresh encryption under the nbit key
        lwe_encrypt_bit(np, nbit.lwe_key.data(), (ciphernegative >> i) & 1, rng, word.data() + i * S);
    write_lwe_samples(ans.f, n, 32, word.data(), S);
    fprintf(log, "%d => total negatives\n", ciphernegative);

    int32_t int_bit = 0;  // :829-856
    if (int_op == 4) {
        int_bit = (int_bit1 >= int_bit2 ? int_bit1 : int_bit2) * 2;
        for (int i = 0; i < 32; i++)
            lwe_encrypt_bit(np, nbit.lwe_key.data(), (int_bit >> i) & 1, rng, word.data() + i * S);
        write_lwe_samples(ans.f, n, 32, word.data(), S);
        fprintf(log, "%d written to answer.data\n", int_bit);
        int_bit = int_bit1 >= int_bit2 ? int_bit1 : int_bit2;
    } else if (int_bit1 >= int_bit2) {
        int_bit = int_bit1;
        write_lwe_samples(ans.f, n, 32, bit1, S);
        fprintf(log, "%d written to answer.data\n", int_bit);
    } else {
        int_bit = int_bit2;
        write_lwe_samples(ans.f, n, 32, bit2, S);
        fprintf(log, "%d written to answer.data\n", int_bit);
    }
    if (report) {
        report->op = int_op;
        report->neg = int_negative;
        report->int_bit = int_bit;
        report->circuit_kind = 0;
        report->bootstraps = 0;
        report->gpu_ms = 0;
    }
    if (int_op == 4 && int_bit >= 256) {  // :860-864
        fprintf(log, "Cannot multiply 256 bit number!\n");
        job->rc = 126;
        return;
    }

    // dispatch (cloud.c:870, 1194-1196, 1809, 2368)
    int32_t kind = 0;
    const char* label = "";
    if ((int_op == 1 && (int_negative != 1 && int_negative != 2)) || (int_op == 2 && (int_negative == 1 || int_negative == 2))) {
        kind = CIRC_ADD;
        label = int_op == 1 ? "Addition" : "Subtraction";
    } else if (int_op == 2 || (int_op == 1 && (int_negative == 1 || int_negative == 2))) {
        if ((int_op == 2 && int_negative == 0) || (int_op == 1 && int_negative == 2)) {
            kind = CIRC_SUB;
            label = int_op == 2 ? "Subtraction" : "Addition computation with 2nd value negative";
        } else {
            kind = CIRC_RSUB;
            label = int_op == 2 ? "Subtraction" : "Addition computation with 1st value negative";
        }
    } else if (int_op == 4) {
        kind = CIRC_MUL;
        label = "Multiplication";
    }
    if (kind == 0) return;  // unknown operator: main() falls through, answer.data keeps 64 samples
    fprintf(log, "%d bit %s computation\n", int_bit, label);
    const bool size_ok = kind == CIRC_MUL ? (int_bit == 32 || int_bit == 64 || int_bit == 128)
                                          : (int_bit == 32 || int_bit == 64 || int_bit == 128 || int_bit == 256);
    if (!size_ok) return;  // no branch of main() matches: 64-sample answer.data = failure marker

    // Opt-in parallel-prefix adders: same decrypted answer, 7x fewer levels for a
    // single expression; NOT the reference's gate sequence, so off unless asked for.
    if (const char* adder = getenv("IEACHE_ADDER")) {
        if (std::string(adder) == "kogge-stone" && kind >= CIRC_ADD && kind <= CIRC_RSUB) kind += CIRC_ADD_KS - CIRC_ADD;
    }
    if (const char* mult = getenv("IEACHE_MULTIPLIER")) {  // opt-in carry-save multiplier: 32 levels instead of 255 at 32 bits
        if (std::string(mult) == "wallace" && kind == CIRC_MUL) kind = CIRC_MUL_WALLACE;
    }
    // Opt-in constant folding (SURVEY App. C note): fewer bootstraps, same decrypted answer, not the
    // reference's ciphertext bits
    const char* fold_env = getenv("IEACHE_FOLD");
    const bool fold = fold_env && fold_env[0] && fold_env[0] != '0';
    const int32_t n_inputs = circuit_n_inputs(kind, int_bit);
    if (n_inputs < 0) return;
    const int W = int_bit / 32;
    // circuit inputs: operand-1 words, operand-2 words, ciphertextcarry1
    job->in.assign((size_t)n_inputs * S, 0);
    memcpy(job->in.data(), opnd1, (size_t)W * WORD * 4);
    memcpy(job->in.data() + (size_t)W * WORD, opnd2, (size_t)W * WORD * 4);
    memcpy(job->in.data() + (size_t)2 * W * WORD, carry1, WORD * 4);
    job->carry1.assign(carry1, carry1 + WORD);
    job->kind = kind;
    job->int_bit = int_bit;
    job->fold = fold;
    job->has_circuit = true;
}

// What main() does after its circuit: the value words, LSW first, then ciphertextcarry1 as filler up to 9 words
// (e.g. cloud.c:899-917), the MUL timing log (:2467-2471) and the chatter.
void cloud_finish(const CloudRunIO& io, const CloudJob& job, const Torus32* out, size_t n_out_samples, double seconds) {
    FILE* const log = io.log ? io.log : stdout;
    const int32_t n = job.params.n;
    const size_t S = (size_t)n + 1;
    fprintf(log, "Computation Time: %lf[sec]\n", seconds);
    if (job.kind == CIRC_MUL || job.kind == CIRC_MUL_WALLACE) {  // cloud.c:2467-2471
        FILE* t_file = io.stats_path.empty() ? nullptr : fopen(io.stats_path.c_str(), "a");
        if (t_file) {
            fprintf(t_file, "%lf\n", seconds);
            fclose(t_file);
        }
    }
    fprintf(log, "writing the answer to file...\n");
    write_lwe_samples(job.answer, n, n_out_samples, out, S);
    for (size_t w = n_out_samples / 32; w < 9; w++) write_lwe_samples(job.answer, n, 32, job.carry1.data(), S);
}

int cloud_run_io(const CloudRunIO& io, const std::function<Evaluator*()>& get_eval, CloudRunReport* report) {
    FILE* const log = io.log ? io.log : stdout;
    CloudJob job;
    cloud_prepare(io, &job, report);
    if (!job.has_circuit) return job.rc;
    Circuit circ;
    if (!build_circuit(job.kind, job.int_bit, &circ, true, job.fold)) return 0;
    const size_t S = (size_t)job.params.n + 1;
    std::vector<Torus32> out(circ.outputs.size() * S);
    Evaluator* eval = get_eval();
    if (!eval) throw std::runtime_error("no evaluator");
    fprintf(log, "Doing the homomorphic computation...\n");
    const double t0 = now_s();
    EvalStats st;
    eval_circuit_host(*eval, circ, 1, job.in.data(), out.data(), &st);
    cloud_finish(io, job, out.data(), circ.outputs.size(), now_s() - t0);
    if (report) {
        report->circuit_kind = job.kind;
        report->bootstraps = st.bootstraps;
        report->gpu_ms = st.total_ms;
    }
    return 0;
}

// Several prepared jobs of one (kind, width, folding) as ONE level-batched evaluation: what makes a resident
// daemon with many clients use the GPU the way bench.py does.  outs[i] receives job i's value samples.
void cloud_eval_jobs(Evaluator& eval, const std::vector<CloudJob*>& jobs, std::vector<std::vector<Torus32>>* outs, EvalStats* stats) {
    if (jobs.empty()) return;
    const CloudJob& first = *jobs[0];
    // built circuits are kept for the life of the process (a daemon evaluates the same handful over and over; building the
    // 128-bit multiplier's DAG and levelising it takes longer than evaluating a small batch of it), keyed like capi.cpp's
    // per-context cache: (kind, width, folding, level cap)
    // Bounded: per (kind, width, folding) the default schedule plus ONE level-capped variant, the most recent (the cap follows
    // the batch size, and a long-running daemon sees many batch sizes; each entry of the wide multipliers is several MB).
    // Entries are shared_ptr: an evaluation in flight keeps its circuit alive when another thread's call replaces it.
    struct Entry {
        std::shared_ptr<const Circuit> base, capped;
        int32_t cap = 0;
    };
    static std::mutex cache_mutex;
    static std::map<std::tuple<int32_t, int32_t, bool>, Entry> cache;
    auto fetch = [&](int32_t cap) -> std::shared_ptr<const Circuit> {
        std::lock_guard<std::mutex> lock(cache_mutex);
        Entry& e = cache[std::make_tuple(first.kind, first.int_bit, first.fold)];
        if (cap == 0 && e.base) return e.base;
        if (cap != 0 && e.capped && e.cap == cap) return e.capped;
        std::shared_ptr<Circuit> c(new Circuit);
        if (!build_circuit(first.kind, first.int_bit, c.get(), true, first.fold, cap)) return nullptr;
        if (cap == 0) {
            e.base = c;
        } else {
            e.capped = c;  // replaces the variant of another batch size
            e.cap = cap;
        }
        return c;
    };
    const std::shared_ptr<const Circuit> base = fetch(0);
    if (!base) throw std::invalid_argument("unsupported circuit");
    std::shared_ptr<const Circuit> circ = base;
    const int32_t cap = circuit_level_cap(*base, (int64_t)jobs.size(), eval.resident_gates(), eval.resident_gates_two_wave());
    if (cap > 0) {
        const std::shared_ptr<const Circuit> capped = fetch(cap);
        if (capped && capped->balanced_schedule) circ = capped;
    }
    const size_t S = (size_t)first.params.n + 1, n_in = (size_t)circ->n_inputs * S, n_out = circ->outputs.size() * S;
    std::vector<Torus32> in(jobs.size() * n_in), out(jobs.size() * n_out);
    for (size_t i = 0; i < jobs.size(); i++) {
        if (jobs[i]->kind != first.kind || jobs[i]->int_bit != first.int_bit || jobs[i]->fold != first.fold || jobs[i]->in.size() != n_in)
            throw std::invalid_argument("jobs of different circuits in one batch");
        memcpy(in.data() + i * n_in, jobs[i]->in.data(), n_in * 4);
    }
    eval_circuit_host(eval, *circ, jobs.size(), in.data(), out.data(), stats);
    outs->resize(jobs.size());
    for (size_t i = 0; i < jobs.size(); i++) (*outs)[i].assign(out.begin() + i * n_out, out.begin() + (i + 1) * n_out);
}

void eval_circuit_host(Evaluator& eval, const Circuit& c, size_t batch, const Torus32* in, Torus32* out,
                       EvalStats* stats) {
    const Params& p = eval.params();
    const size_t S = (size_t)p.n + 1, stride = (size_t)p.lwe_stride();
    const size_t n_in = (size_t)c.n_inputs * batch, n_out = c.outputs.size() * batch;
    if (batch == 0) return;
    HIP_CHECK(hipSetDevice(eval.device()));
    // the evaluator's own staging rows: a warm call (same or smaller batch) allocates nothing
    Torus32* d_in = eval.staging(0, n_in * stride * 4);
    Torus32* d_out = eval.staging(3, n_out * stride * 4);
    HIP_CHECK(hipMemcpy2D(d_in, stride * 4, in, S * 4, S * 4, n_in, hipMemcpyHostToDevice));
    eval.eval_circuit_device(c, batch, d_in, d_out, stats);
    HIP_CHECK(hipMemcpy2D(out, S * 4, d_out, stride * 4, S * 4, n_out, hipMemcpyDeviceToHost));
}

}  // namespace ieache
