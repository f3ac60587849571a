// `./cloud` process contract of the reference as a function (cloud_run.cpp).
#pragma once
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "circuit.h"
#include "evaluator.h"
#include "params.h"

namespace ieache {

struct CloudRunReport {
    int32_t op = 0, neg = 0, int_bit = 0;
    int32_t circuit_kind = 0;  // 0 = nothing evaluated
    int64_t bootstraps = 0;
    double gpu_ms = 0;
};

// Same file contract and return codes as main() of Cloud/cloud.c: reads
// cloud.key, nbit.key, cloud.data, operator.txt from `dir`; writes answer.data
// (64 metadata samples + 288 value samples, or exactly 64 samples when nothing
// was computed) and appends to averagestandard.txt for MUL.  Returns 0 or 126.
// `shared_eval` (optional) supplies an evaluator whose keys are already
// resident, so cloud.key is not re-read (the reference reloads it per call,
// cloud.c:656-658).  Throws CodecError / std::runtime_error on I/O or GPU failure.
int cloud_run(const std::string& dir, Evaluator* shared_eval, CloudRunReport* report, int device = 0,
              FILE* log = nullptr /* stdout */);

// The same contract on open streams instead of files in a directory: what the resident-key daemon
// (daemon.cpp) runs when cloud.data arrives over a socket.
struct CloudRunIO {
    Params params;                       // of the cloud key
    const SecretKeyData* nbit = nullptr; // the metadata key (cloud.c:661-663)
    FILE* cloud_data = nullptr;          // 704 LweSamples (cloud.c:703-766)
    int32_t op = 0;                      // content of operator.txt (cloud.c:769-773)
    std::function<FILE*()> open_answer;  // called once the inputs are read (cloud.c:809)
    std::string stats_path;              // averagestandard.txt; empty = do not append
    FILE* log = nullptr;                 // stdout chatter; null = stdout
};
// The files of one `./cloud` call (nbit.key, cloud.data, operator.txt -> answer.data, averagestandard.txt in `dir`)
// opened and wired into a CloudRunIO; closes them when it goes.  Not copyable: io points into it.
struct CloudDirSession {
    std::string dir;
    SecretKeyData nbit;
    FILE* data = nullptr;
    FILE* answer = nullptr;
    CloudRunIO io;
    CloudDirSession(const std::string& dir, const Params& cloud_params, FILE* log);
    ~CloudDirSession();
    CloudDirSession(const CloudDirSession&) = delete;
    CloudDirSession& operator=(const CloudDirSession&) = delete;
};

// The contract in three steps, so that a daemon holding several requests can evaluate the circuits of all of
// them as one batch: cloud_prepare (inputs, metadata, the 64 metadata samples, branch choice), evaluation of
// job.in through circuit (job.kind, job.int_bit, job.fold), cloud_finish (value samples + filler words, logs).
struct CloudJob {
    Params params;
    int rc = 0;                 // what main() returns when has_circuit is false: 0 or 126
    bool has_circuit = false;
    int32_t kind = 0, int_bit = 0;
    bool fold = false;
    std::vector<Torus32> in;      // circuit inputs, rows of n+1
    std::vector<Torus32> carry1;  // 32 rows: the filler word of answer.data (cloud.c:901-916)
    FILE* answer = nullptr;       // io.open_answer()'s sink, metadata already written
};
void cloud_prepare(const CloudRunIO& io, CloudJob* job, CloudRunReport* report);
void cloud_finish(const CloudRunIO& io, const CloudJob& job, const Torus32* out, size_t n_out_samples, double seconds);
void cloud_eval_jobs(Evaluator& eval, const std::vector<CloudJob*>& jobs, std::vector<std::vector<Torus32>>* outs, EvalStats* stats);

// get_eval is called only if a circuit is actually evaluated.
int cloud_run_io(const CloudRunIO& io, const std::function<Evaluator*()>& get_eval, CloudRunReport* report);

// Host-buffer convenience: rows of (n+1) int32 in and out.
void eval_circuit_host(Evaluator& eval, const Circuit& c, size_t batch, const Torus32* in, Torus32* out,
                       EvalStats* stats);

}  // namespace ieache
