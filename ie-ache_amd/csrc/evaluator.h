// Device-side gate-bootstrapping evaluator (HIP, gfx950).
//
// Replaces, for whole levels of independent gates at once, what the reference
// does one gate at a time through libtfhe's bootsAND / bootsXOR ->
// tfhe_bootstrap_FFT (Cloud/cloud.c:30-43,159).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "circuit.h"
#include "params.h"

namespace ieache {

struct EvalStats {
    double total_ms = 0;         // wall time of the call on the GPU timeline (events on the stream)
    double blind_rotate_ms = 0;  // time with a blind-rotation launch in flight: the sum over launches on one stream; with
                                 // overlapped levels the union of the two streams' intervals on the device timeline
    double keyswitch_ms = 0;     // the same for key-switch launches (under overlap mostly hidden behind the other stream's rotation)
    int64_t blind_rotate_launches = 0;  // k_blind_rotate_* kernel launches (one per slice of CMux steps per chunk)
    int64_t keyswitch_launches = 0;
    int64_t chunks = 0;  // (level, chunk) work units = key-switch launches
    int64_t bootstraps = 0;  // gate instances bootstrapped
    int64_t levels = 0;
};

class Evaluator {
public:
    Evaluator(const Params& p, int device);
    ~Evaluator();
    Evaluator(const Evaluator&) = delete;
    Evaluator& operator=(const Evaluator&) = delete;

    const Params& params() const { return p_; }
    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }
    // gate instances the blind-rotation kernel keeps resident at once (one workgroup each, 4 per CU): a launch
    // takes ceil(gate instances / this) rounds
    int resident_gates() const { return resident_gates_; }
    // ... and by the two-waves-per-gate one-limb kernel that takes launches up to that size (0 when "exact_fft" leaves one size)
    int resident_gates_two_wave() const { return resident_two_wave_; }
    // Orders everything launched later on the evaluator's stream after the work queued so far on
    // `producer` (the stream that wrote the key / input buffers handed over as device pointers).
    void wait_for_stream(hipStream_t producer);

    // Upload the cloud key.  Raw libtfhe order: bk [n][(k+1)l][k+1][N],
    // ksk [kN][t][base][n+1].  The *_device form takes pointers already in this
    // GPU's memory (e.g. the receive buffer of an RCCL broadcast).
    void load_keys_host(const Torus32* bk, const Torus32* ksk);
    void load_keys_device(const Torus32* d_bk, const Torus32* d_ksk);
    bool keys_loaded() const { return keys_loaded_; }

    // `count` independent gates of one type on device rows of lwe_stride()
    // int32: out[i] = gate(a[i], b[i]).
    void gates_device(int32_t type, size_t count, const Torus32* d_a, const Torus32* d_b,
                      Torus32* d_out, EvalStats* stats);

    // bootsMUX: out[i] = a[i] ? b[i] : c[i] (two blind rotations + one key switch per gate)
    void mux_device(size_t count, const Torus32* d_a, const Torus32* d_b, const Torus32* d_c, Torus32* d_out,
                    EvalStats* stats);

    // allocates what an evaluation of `c` over `batch` expressions needs (idempotent; eval_circuit_device calls it itself)
    void prepare_circuit(const Circuit& c, size_t batch);
    // One circuit on `batch` independent expressions.
    //   d_in  [batch][circuit.n_inputs][lwe_stride]
    //   d_out [batch][circuit.outputs.size()][lwe_stride]
    void eval_circuit_device(const Circuit& c, size_t batch, const Torus32* d_in, Torus32* d_out,
                             EvalStats* stats);

    // Device rows the host-buffer entry points stage operands (slot 0 .. 2) and results (slot 3) in: owned by the evaluator,
    // kept between calls and grown on demand, so that a warm call allocates nothing.  Operand slots are zero outside what
    // the caller uploads (rows of lwe_stride() words, n + 1 of them uploaded).  get_option("staging_allocations") counts
    // the (re)allocations made so far.
    Torus32* staging(int slot, size_t bytes);

    // ---- single-stage hooks (parity tests compare each against its oracle stage) ----
    // x [count][lwe_stride] -> acc [count][2][N] after `steps` CMux steps (steps<0: all n),
    // starting from the test-vector initialisation.
    void debug_blind_rotate(size_t count, const Torus32* d_x, Torus32* d_acc, int32_t steps);
    // u [count][N+1] -> out [count][lwe_stride]
    void debug_keyswitch(size_t count, const Torus32* d_u, Torus32* d_out);

    // Force the generic (any-parameter) kernels even where a specialised one exists.
    void set_force_generic(bool v) { force_generic_ = v; }
    // Maximum gate instances per launch (bounds the scratch buffers).
    void set_chunk(size_t items);
    // Named tuning knobs: "chunk", "force_generic", "ks_mfma_min" (gate instances per launch from which the key switch runs
    // as an int8 product on the MFMA pipe, keyswitch_mfma.hip; default 64), "ks_mfma_split" (workgroups its walk is cut into
    // per tile: 1, 2, 4, 8, or 0 = by launch size), "ks_sliced_min" (gate instances per launch from which, below that or
    // with it disabled, the hand-scheduled walk is used; default 576), "ks_gates" (its gate instances per workgroup:
    // 4, 8, 16, 32, or 0 = by launch size), "ks_slice" (coefficients per launch of it, 0 = whole walk),
    // "ks_batch_min" (same threshold for the compiler-scheduled gate-batched kernel, the cross-check),
    // "ks_split_max" (workgroups the per-gate key switch may cut one gate's walk into when a launch holds only a
    // handful of gates; default 16, 1 = never),
    // "br_slice" (CMux steps per blind-rotation launch; 0 = by kernel and launch size: 16 for wide launches that take
    // several rounds of resident gates, 64 while every gate of the launch is resident at once, the whole rotation for the
    // four-waves-per-gate and latency kernels),
    // "br_wide_max" (launches of at most this many gate instances use the latency-oriented
    // 2L-waves-per-gate kernel; default = the device's CU count, 0 = never), "br_variant",
    // "exact_fft" (1 = two-limb blind rotation always), "one_limb_min" (launches of at least this many gate
    // instances use the one-limb kernels; default: one per CU + 1), "two_wave_max" (of those, launches up to this many
    // gate instances take two waves per gate, k_blind_rotate_w2r; default 5 per CU), "four_wave_max" (launches up to this many
    // take four waves per gate, k_blind_rotate_w4r; default 2 per CU), "fft_guard_inject" (test hook: 1 makes
    // the next call find the rounding guard tripped, so that it repeats itself on the two-limb kernels), "fft_audit" (see
    // fft_audit_counts), "fft_audit_inject" (test hook: 1 makes the next audit report a differing row).
    // Returns false for an unknown name or a value out of range.
    bool set_option(const std::string& name, int64_t value);
    // "overlap" (default 1; IEACHE_OVERLAP): launches go to TWO streams of the context (own scratch each, the one key copy),
    // so that the ragged end of one stream's launch, its key switch and its prologue run under the other stream's rotation:
    //   * a circuit over a batch whose mean level holds at least "pipe_min" gate instances (default 8 per CU;
    //     IEACHE_PIPE_MIN): the batch is cut into two halves of EXPRESSIONS and each half runs through every level on its own
    //     stream -- expressions are independent, so there is one fork after the input copy and one join before the outputs
    //     are gathered, nothing in between (add16 x 4096: +3.5 %, mul32 x 1024: +1.1 %, profiles/r5_overlap_ab.txt);
    //     kernels are chosen by the gate instances in flight on both streams ("pipe_lanes" = 3 or 4 cuts the batch into
    //     that many pipelines instead: measured no better than two, profiles/r5_overlap_ab.txt);
    //   * otherwise (flat gate calls, narrower circuits) a level of at least "overlap_min" gate instances (default 16 per
    //     CU; IEACHE_OVERLAP_MIN) is cut into pieces of at most half the level that alternate between the two streams, and
    //     the next level starts when both have finished.
    //   * "pipe_auto" (default 1): with a mean level between pipe_min / 8 and 2 x pipe_min neither mode wins everywhere, so the
    //     first four evaluations of a (circuit, batch) alternate without / with pipelines and later ones take the faster
    //     (each is a complete evaluation; "tuned_evals" counts the trials);
    //   * "br_mix" (default 1; IEACHE_BR_MIX): a launch of 4 .. 7 gates per CU -- or of 8 .. 10.5: a full round of the
    //     one-wave kernel plus a small remainder -- that has the chip to itself runs as a ROTATION OF ROLES: the gates in
    //     three subsets on three streams, two subsets at a time on the two-waves-per-gate kernel for "mix_s1" x "mix_ratio"
    //     / 100 steps while the third takes "mix_s1" steps on the one-wave-per-gate kernel ("mix_wg" gates per workgroup),
    //     roles rotating, so that every wave slot of a CU works whatever the launch size: a size-independent 180-185 k
    //     gates/s where the single kernels give 130-180 k (profiles/r5_mix_sweep.txt; csrc/mix_plan.h).
    // The same gate instances go through the same kernels' arithmetic either way: output bits do not depend on it.
    // 0 = every launch on one stream -- the mode per-kernel timings (rocprofv3 averages, bench.py's roofline) are taken
    // in, since overlapped kernels share the chip.
    // Current value of an option, or of a read-only figure: "cus", "resident_gates", "overlapped_levels" (levels issued as
    // halves on two streams so far), "pipelined_evals" (circuit evaluations run as two expression-half pipelines so far),
    // "staging_allocations".  false for an unknown name.
    bool get_option(const std::string& name, int64_t* value) const;
    std::string kernel_variant() const;
    // name of the blind-rotation kernel a launch of `gates` gate instances takes under the current options
    std::string kernel_for_launch(int64_t gates) const;
    // The one-limb blind rotation (k_blind_rotate_w1) rounds sums an FP64 transform carries with ~2^-9 of error instead of
    // provably none; it records how far from an integer its coefficients came.  fft_guard_max(): the largest such distance
    // over the context's life (0.5 would be a wrong bit; the kernel's limit is 1/16); fft_guard_reruns(): calls that crossed
    // the limit and were therefore repeated on the two-limb kernel.  Option "exact_fft" = 1 uses the two-limb kernel always.
    double fft_guard_max() const;
    int64_t fft_guard_reruns() const;
    bool fft_guard_tripped();  // internal: reads and re-arms the device-side record
    // The sampled audit behind the guard (option "fft_audit" = K, default 64, 0 = off; IEACHE_FFT_AUDIT): every K-th launch
    // that took a one-limb kernel has 64 of its gate instances run again on the two-limb kernel and compared word for word;
    // a differing row makes the call repeat itself on the two-limb kernels (counted in fft_guard_reruns()).
    // -> audits run, gate instances compared, rows that differed, over the context's life.
    void fft_audit_counts(int64_t* audits, int64_t* gates, int64_t* mismatches) const;

    struct Impl;  // device buffers; defined in evaluator.hip

private:
    void gates_device_once(int32_t type, size_t count, const Torus32* d_a, const Torus32* d_b, Torus32* d_out, EvalStats* stats);
    void mux_device_once(size_t count, const Torus32* d_a, const Torus32* d_b, const Torus32* d_c, Torus32* d_out, EvalStats* stats);
    void eval_circuit_device_once(const Circuit& c, size_t batch, const Torus32* d_in, Torus32* d_out, EvalStats* stats);
    void debug_blind_rotate_once(size_t count, const Torus32* d_x, Torus32* d_acc, int32_t steps);
    void init();
    void destroy();
    Params p_;
    int device_;
    hipStream_t stream_ = nullptr;
    bool keys_loaded_ = false;
    bool force_generic_ = false;
    int resident_gates_ = 1024;
    int resident_two_wave_ = 0;
    Impl* d_ = nullptr;
};

// throws std::runtime_error carrying the HIP error string
void hip_check(hipError_t e, const char* what, const char* file, int line);
#define HIP_CHECK(x) ::ieache::hip_check((x), #x, __FILE__, __LINE__)

}  // namespace ieache
