// `./cloud` process contract of the reference as a function (cloud_run.cpp).
#pragma once
#include <memory>
#include <string>

#include "circuit.h"
#include "evaluator.h"

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
int cloud_run(const std::string& dir, Evaluator* shared_eval, CloudRunReport* report, int device = 0);

// Host-buffer convenience: rows of (n+1) int32 in and out.
void eval_circuit_host(Evaluator& eval, const Circuit& c, size_t batch, const Torus32* in, Torus32* out,
                       EvalStats* stats);

}  // namespace ieache
