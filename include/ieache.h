/*
 * ieache.h -- C ABI of the MI355X-native evaluator for the IE-ACHE Cloud path.
 *
 * Drop-in boundary for the hot path of kennethsoh/IE-ACHE: the homomorphic ALU
 * in Cloud/cloud.c and the libtfhe gate bootstrapping beneath it.  Plain C
 * types only, caller-allocated buffers, no exceptions cross this boundary:
 * every function returns 0 on success or a negative IEACHE_E* code (the
 * process-contract function returns the reference's exit codes 0 / 126), and
 * ieache_last_error() holds the message.  A context is not thread-safe; use one
 * per process per GPU.
 *
 * Streams: a context launches on its own non-blocking HIP stream
 * (ieache_ctx_stream).  Every entry point that takes DEVICE pointers
 * (ieache_ctx_create_device, ieache_eval_batch_device, ieache_gates_device,
 * ieache_mux_device) reads them on that stream without ordering against the
 * stream that produced them: the caller must either have synchronised the
 * producing stream (torch.cuda.synchronize(), hipStreamSynchronize) or call
 * ieache_ctx_wait_stream(ctx, producer) first.  Outputs are complete when the
 * call returns (each call synchronises the context's stream before returning).
 * Device-pointer arguments are checked with hipPointerGetAttributes: a host or
 * stray address is refused with IEACHE_EINVAL instead of reaching a kernel.
 *
 * Sample layout: one LWE sample = int32[n+1] = a[0..n-1], b (Torus32).  Host
 * buffers are packed rows of n+1; DEVICE buffers are rows of
 * ieache_lwe_stride() int32 (n+1 rounded up to a multiple of 4).
 *
 * Each entry point names the reference interface it replaces.
 */
#ifndef IEACHE_H
#define IEACHE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IEACHE_EINVAL (-22)   /* bad argument / unsupported parameter set */
#define IEACHE_EIO (-5)       /* file missing, short or malformed */
#define IEACHE_ENODEV (-19)   /* no usable GPU / HIP failure */
#define IEACHE_ENOMEM (-12)

/* TFHE parameter set; replaces TFheGateBootstrappingParameterSet as read from
 * the key header (Cloud/cloud.c:666-669, Keygen/keygen.c:22-23). */
typedef struct ieache_params {
    int32_t n, N, k, l, Bgbit, ks_t, ks_basebit;
    double lwe_alpha_min, lwe_alpha_max, tlwe_alpha_min, tlwe_alpha_max;
} ieache_params;

/* libtfhe >= 1.1 new_default_gate_bootstrapping_parameters(110) (keygen.c:22-23):
 * n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2, sigma 2^-15 / 2^-25. */
void ieache_default_params(ieache_params* out);

typedef struct ieache_stats {
    double total_ms;          /* GPU timeline of the call (HIP events on the evaluator's stream) */
    double blind_rotate_ms;   /* sum over blind-rotation launches */
    double keyswitch_ms;      /* sum over key-switch launches */
    int64_t blind_rotate_launches; /* blind-rotation kernel launches (one per slice of CMux steps per chunk) */
    int64_t keyswitch_launches;
    int64_t bootstraps;       /* bootsAND/bootsXOR-equivalent gate instances evaluated */
    int64_t levels;
    int64_t chunks;           /* (level, chunk) work units */
} ieache_stats;

/* circuit kinds = the branches of main() in Cloud/cloud.c */
#define IEACHE_CIRC_ADD 1     /* A+B                cloud.c:870-1190  */
#define IEACHE_CIRC_SUB 2     /* A+(~B+1)           cloud.c:1196-1807 */
#define IEACHE_CIRC_RSUB 3    /* B+(~A+1)           cloud.c:1809-2365 */
#define IEACHE_CIRC_MUL 4     /* A*B, double width  cloud.c:2366-2718 */
#define IEACHE_CIRC_ADD_KS 6   /* Kogge-Stone variants of ADD/SUB/RSUB: same decrypted */
#define IEACHE_CIRC_SUB_KS 7   /* result, depth 2*log2(bits)+2 instead of 3*bits, not the same       */
#define IEACHE_CIRC_RSUB_KS 8  /* ciphertext bits as the reference's ripple adders                   */
#define IEACHE_CIRC_MUL_WALLACE 9 /* A*B by carry-save (Wallace) reduction + one Kogge-Stone add: same decrypted
                                     product as MUL, ~8x fewer levels (32 instead of 255 at 32 bits) and fewer
                                     bootstraps; not the reference's ciphertext; IEACHE_MULTIPLIER=wallace selects
                                     it for the process contract */
#define IEACHE_CIRC_MULADD 5  /* (A*B)+C, the compute_final() chaining of
                                 Cloud/dragonfly_cipher_cloud.py:1300-1327 fused; 32-, 64- or 128-bit A,B
                                 (= IEACHE_CIRC_CHAIN(MUL, ADD, 1)) */
/* Any two operators chained as compute() + compute_final() do
 * (dragonfly_cipher_cloud.py:1219-1327), fused into one DAG without the answer.data
 * round trip: stage 1 = k1(A, B); stage 2 = k2(answer, C) when flip (cloud.data =
 * answer | C, :1306-1314) or k2(C, answer) otherwise (:1318-1326).  k1, k2 in
 * {ADD, SUB, RSUB, MUL}.  Inputs per expression: A bits, B bits, A's 32-sample carry
 * word, C at stage 2's width (bits, or 2*bits after a MUL) [, C's carry word when
 * !flip].  E.g. the paper's A+B-C = IEACHE_CIRC_CHAIN(ADD, SUB, 1), A*B*C =
 * IEACHE_CIRC_CHAIN(MUL, MUL, 1) (AC058.pdf Fig. 7). */
#define IEACHE_CIRC_CHAIN(k1, k2, flip) (32 + ((k1)-1) + 4 * ((k2)-1) + ((flip) ? 0 : 16))

/* gate types = libtfhe boot-gates.cpp entry points used by cloud.c:30-43,159 */
#define IEACHE_GATE_AND 0
#define IEACHE_GATE_XOR 1
#define IEACHE_GATE_OR 2
#define IEACHE_GATE_NAND 3
/* three-input gate, own entry points (ieache_mux*): bootsMUX(a,b,c) = a ? b : c.
 * Not called by Cloud/cloud.c; BASELINE.json's north_star names it. */
#define IEACHE_GATE_MUX 4

typedef struct ieache_circuit_info {
    int32_t n_inputs;    /* samples per expression: A bits, B bits, 32-sample carry word [, C bits] */
    int32_t n_outputs;   /* samples per expression, LSB first */
    int32_t n_slots;     /* wire-store rows per expression on the device */
    int32_t depth;       /* ASAP levels */
    int32_t max_width;   /* widest ASAP level */
    int64_t bootstraps, n_and, n_xor;
    int32_t sched_max_width; /* widest level of the slack-balanced schedule the executor runs */
    int32_t folded;          /* 1 when this is the constant-folded variant (see "fold_constants") */
    int64_t reference_bootstraps; /* gates Cloud/cloud.c performs for this circuit; == bootstraps unless folded */
    int32_t sched_levels;    /* levels the executor runs (== depth unless a level cap stretched the schedule) */
    int32_t level_cap;       /* the cap this schedule was built with (0 = mean ASAP width) */
} ieache_circuit_info;
/* the struct of header version 0.1 ended with `folded`: ieache_circuit_info_get() -- the entry point that existed then --
 * writes these bytes and no more, so a caller built against that header is not written past its struct; the later
 * fields come from ieache_circuit_info_get_ex / _get_cap */
#define IEACHE_CIRCUIT_INFO_V01_BYTES 56

const char* ieache_version(void);
const char* ieache_last_error(void);
/* Key files are libtfhe's tfhe_io.cpp serialisations; libtfhe is not in the reference tree, so the
 * reader locates text sections by title and picks the binary layout among enumerated hypotheses
 * by the exact byte count (csrc/codec.h).  This names the hypothesis the calling thread's last
 * key load matched. */
const char* ieache_last_key_layout(void);
int ieache_device_count(void);

/* ------------------------------------------------------------------ *
 * 1. Process contract.  Replaces subprocess.call("./cloud")           *
 *    (Cloud/dragonfly_cipher_cloud.py:1233,1248,1263,1278) = main()   *
 *    of Cloud/cloud.c:650-2720.  Reads cloud.key, nbit.key,           *
 *    cloud.data, operator.txt in `workdir`; writes answer.data        *
 *    (352 samples, or exactly 64 = failure marker checked at          *
 *    dragonfly_cipher_cloud.py:1295); appends averagestandard.txt on  *
 *    MUL.  Returns 0 or 126 like the reference, or IEACHE_E*.         *
 * ------------------------------------------------------------------ */
int ieache_cloud_run(const char* workdir);

/* ------------------------------------------------------------------ *
 * 2. Context: cloud key resident on one GPU.  Replaces                *
 *    new_tfheGateBootstrappingCloudKeySet_fromFile (cloud.c:656-658), *
 *    without the per-call reload.                                     *
 * ------------------------------------------------------------------ */
typedef struct ieache_ctx ieache_ctx;

/* from a cloud.key file */
ieache_ctx* ieache_ctx_create(const char* cloud_key_path, int device);
/* from raw arrays on the host: bk [n][(k+1)l][k+1][N], ksk [kN][t][base][n+1] */
ieache_ctx* ieache_ctx_create_raw(const ieache_params* p, const int32_t* bk, const int32_t* ksk, int device);
/* from raw arrays already in this GPU's memory (e.g. an RCCL broadcast buffer) */
ieache_ctx* ieache_ctx_create_device(const ieache_params* p, const int32_t* d_bk, const int32_t* d_ksk, int device);
void ieache_ctx_destroy(ieache_ctx* ctx);
int ieache_ctx_params(const ieache_ctx* ctx, ieache_params* out);
int ieache_lwe_stride(const ieache_ctx* ctx);
/* the HIP stream the evaluator launches on (hipStream_t as void*) */
void* ieache_ctx_stream(const ieache_ctx* ctx);
/* make the context's stream wait for the work queued so far on `hip_stream`
 * (hipStream_t as void*; NULL = the default stream) -- see "Streams" above */
int ieache_ctx_wait_stream(ieache_ctx* ctx, void* hip_stream);
/* The wide-launch blind rotation multiplies with ONE FP64 transform of the 32-bit key coefficients, as libtfhe does
 * (lwe-bootstrapping-functions-fft.cpp -> tGswFFTExternMulToTLwe), instead of the provably exact two-limb transform:
 * its rounded sums carry ~2^-9 of error, and the kernel records how far from an integer they came.
 * max_deviation: the largest distance seen by this context (0.5 would flip a bit; a launch above 1/16 makes the call
 * repeat itself on the two-limb kernel, counted in reruns).  Option "exact_fft" = 1 (or IEACHE_EXACT_FFT=1) uses the
 * two-limb kernel always.  Either pointer may be NULL. */
int ieache_ctx_fft_guard(const ieache_ctx* ctx, double* max_deviation, int64_t* reruns);
/* The guard watches the error LEVEL of every launch; the audit compares BITS of a sample: every K-th launch that took the
 * one-FFT kernel (option "fft_audit" = K, default 64, 0 = off; IEACHE_FFT_AUDIT=K) has 64 of its gate instances run again on
 * the two-limb kernel, and the extracted samples are compared word for word on the device.  A differing row makes the call
 * repeat itself on the two-limb kernels (counted in `reruns` above).  audits: audits run by this context; gates_compared:
 * gate instances they covered; mismatches: rows that differed (0 in every run so far).  Any pointer may be NULL.
 * A call whose output buffer overlaps an input cannot be repeated, so it runs on the two-limb kernels from the start. */
int ieache_ctx_fft_audit(const ieache_ctx* ctx, int64_t* audits, int64_t* gates_compared, int64_t* mismatches);
/* same contract as ieache_cloud_run but with this context's resident key */
int ieache_ctx_cloud_run(ieache_ctx* ctx, const char* workdir);
/* tuning / test knobs */
int ieache_ctx_set_chunk(ieache_ctx* ctx, int64_t gate_instances_per_launch);
int ieache_ctx_force_generic(ieache_ctx* ctx, int on);
/* named knobs: "chunk", "force_generic", "ks_mfma_min" (launches of at least this many gate instances key-switch as an
 * int8 product on the MFMA pipe; default 64), "ks_mfma_split", "ks_sliced_min", "ks_gates", "ks_slice", "ks_batch_min",
 * "br_slice", "br_wide_max", "br_variant" (0 or a number of csrc/blind_rotate_w64.h's table), "exact_fft", "exact_one_wave_min", "fft_audit", "one_limb_min", "four_wave_max", "two_wave_max", "ks_split_max"
 * (see csrc/evaluator.h), and
 * "level_quantum" (0/1, default 1: the slack-balanced circuits -- 64/128-bit multipliers -- get a level
 * width that makes level x batch a whole number of resident-workgroup rounds; same DAG and output bits, more
 * levels of exactly-full launches when the batch is small), and
 * "fold_constants" (0/1, default 0, also IEACHE_FOLD=1 for the process contract): build circuits
 * with constant operands folded and repeated gates shared.  cloud.c bootstraps every gate, even
 * `x AND 0` on the zero rows of its shift-add multipliers (SURVEY App. C note); the folded circuit
 * decrypts to the same bits with fewer bootstraps but is NOT the reference's ciphertext. */
int ieache_ctx_set_option(ieache_ctx* ctx, const char* name, int64_t value);
/* "overlap" (0/1, default 1; IEACHE_OVERLAP): the context issues its launches on two streams.  A circuit over a batch whose
 * mean level holds at least "pipe_min" gate instances (default 8 per CU) runs as two pipelines, each taking half of the
 * EXPRESSIONS through every level with no join in between; otherwise a level of at least "overlap_min" gate instances
 * (default 16 per CU) is issued as two halves with a join before the next level.  Same output bits either way; 0 = one
 * stream, the mode per-kernel timings are taken in (csrc/evaluator.h).  "pipe_auto" (0/1, default 1): between pipe_min / 8
 * and 2 x pipe_min the mode is chosen per (circuit, batch) by timing its first four evaluations, two in each mode.
 * "br_mix" (0/1, default 1; IEACHE_BR_MIX): launches of 4 .. 7 and of 8 .. 10.5 gates per CU run as a rotation of roles
 * between the two-waves- and the one-wave-per-gate kernel on three streams ("mix_s1", "mix_ratio", "mix_wg": turn length,
 * step ratio x 100, gates per workgroup).  "wg_gates" (0 = by launch size, 1 .. 4): gate instances per workgroup of the
 * one-wave-per-gate kernels.
 * ieache_ctx_get_option: the current value of any option above, or of the read-only figures "cus" (compute units of the
 * context's device), "resident_gates", "overlapped_levels", "pipelined_evals", "tuned_evals", "mixed_launches",
 * "staging_allocations". */
int ieache_ctx_get_option(const ieache_ctx* ctx, const char* name, int64_t* value);
const char* ieache_ctx_kernel_variant(const ieache_ctx* ctx);
/* Name of the blind-rotation kernel a launch of `gates` gate instances takes under the context's
 * current options (launch sizes select different kernels: docs in csrc/blind_rotate_w64.h).  Like
 * ieache_ctx_kernel_variant the string lives in the context until the next such call. */
const char* ieache_ctx_kernel_for_launch(const ieache_ctx* ctx, int64_t gates);

/* ------------------------------------------------------------------ *
 * 3. Batch evaluation: `batch` independent expressions through one    *
 *    circuit, level by level.  Replaces the add()/mul32()/mul64()/    *
 *    mul128()/split() call trees of cloud.c:18-647 and their use in   *
 *    main().  bits: 16 (generalised add(...,16,...)), 32, 64, 128,    *
 *    256 for ADD/SUB/RSUB; 32/64/128 for MUL and MULADD; chains: see  *
 *    IEACHE_CIRC_CHAIN.                                               *
 * ------------------------------------------------------------------ */
int ieache_circuit_info_get(int kind, int bits, ieache_circuit_info* out);
int ieache_circuit_info_get_ex(int kind, int bits, int fold_constants, ieache_circuit_info* out);
/* the schedule a context would pick for `batch` expressions on a GPU holding `resident_workgroups` blind rotations
 * at once (4 per CU; "level_quantum"): level_cap = 0 reproduces ieache_circuit_info_get_ex */
int ieache_circuit_level_cap(int kind, int bits, int fold_constants, int64_t batch, int resident_workgroups);
/* ... the same for the GPU and the kernels THIS context runs (one-wave and two-wave residency of its device, its
 * fold_constants setting): the level width ieache_eval_batch* will use for `batch` expressions; 0 = the default schedule */
int ieache_ctx_circuit_level_cap(const ieache_ctx* ctx, int kind, int bits, int64_t batch);
int ieache_circuit_info_get_cap(int kind, int bits, int fold_constants, int level_cap, ieache_circuit_info* out);
int ieache_circuit_simulate_cap(int kind, int bits, int fold_constants, int level_cap, const uint8_t* in_bits, uint8_t* out_bits);
/* host buffers: in [batch][n_inputs][n+1], out [batch][n_outputs][n+1] */
int ieache_eval_batch(ieache_ctx* ctx, int kind, int bits, size_t batch, const int32_t* in_lwe, int32_t* out_lwe,
                      ieache_stats* stats);
/* device buffers: rows of ieache_lwe_stride() int32 */
int ieache_eval_batch_device(ieache_ctx* ctx, int kind, int bits, size_t batch, const int32_t* d_in, int32_t* d_out,
                             ieache_stats* stats);
/* Builds (and caches) the circuit and allocates everything ieache_eval_batch*(ctx, kind, bits, batch, ...) needs on the
 * device -- wire store, gate tables, scratch of the widest level -- so that the evaluation itself allocates nothing.
 * Optional: the evaluation does the same on first use (the reference has no counterpart: ./cloud allocates per run,
 * cloud.c:651-700). */
int ieache_prepare_batch(ieache_ctx* ctx, int kind, int bits, size_t batch);
/* `count` independent gates: out[i] = gate(a[i], b[i]); replaces bootsAND /
 * bootsXOR / bootsOR / bootsNAND (cloud.c:30-43,159).  Device rows. */
int ieache_gates_device(ieache_ctx* ctx, int gate_type, size_t count, const int32_t* d_a, const int32_t* d_b,
                        int32_t* d_out, ieache_stats* stats);
/* host rows of n+1 */
int ieache_gates(ieache_ctx* ctx, int gate_type, size_t count, const int32_t* a, const int32_t* b, int32_t* out,
                 ieache_stats* stats);
/* out[i] = a[i] ? b[i] : c[i]; replaces bootsMUX (libtfhe boot-gates.cpp): per gate two blind
 * rotations without key switch, their extracted samples added to (0, 1/8), one key switch.
 * stats->bootstraps counts the blind rotations (2 per gate). */
int ieache_mux_device(ieache_ctx* ctx, size_t count, const int32_t* d_a, const int32_t* d_b, const int32_t* d_c,
                      int32_t* d_out, ieache_stats* stats);
int ieache_mux(ieache_ctx* ctx, size_t count, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* out,
               ieache_stats* stats);
/* plaintext simulation of the levelised circuit (host only, no GPU): bits in/out 0/1 */
int ieache_circuit_simulate(int kind, int bits, const uint8_t* in_bits, uint8_t* out_bits);
int ieache_circuit_simulate_ex(int kind, int bits, int fold_constants, const uint8_t* in_bits, uint8_t* out_bits);

/* stage hooks for parity tests (host rows): blind rotation from the
 * test-vector after `steps` CMux steps (<0: all n) -> acc [count][2][N];
 * key switch u [count][N+1] -> out [count][n+1] */
int ieache_debug_blind_rotate(ieache_ctx* ctx, size_t count, const int32_t* x, int32_t* acc, int32_t steps);
int ieache_debug_keyswitch(ieache_ctx* ctx, size_t count, const int32_t* u, int32_t* out);
/* The plan of a rotation of roles (option "br_mix") for a launch of `gates` gate instances on a device of `cus` compute units,
 * rotations of n steps, one-wave turns of s1 steps, two-wave turns of s1 x ratio_x100 / 100: returns 1 and out = {subsets k,
 * subsets on two waves at a time, s1, s2, whole rounds, shortened round's s1, s2, steps covered by the rounds (< n), gate
 * instances per subset}, or 0 (out zeroed) when that launch size takes a single kernel.  No device needed (csrc/mix_plan.h). */
int ieache_debug_mix_plan(int cus, int n, int64_t gates, int s1, int ratio_x100, int out[9]);

/* ------------------------------------------------------------------ *
 * 4. CPU tools around the path (no GPU): what Keygen/keygen.c:22-51,  *
 *    Client1/alice.c:116-191 and Output/verif.c:41-76 get from        *
 *    libtfhe.  Needed to produce and check ciphertexts without it.    *
 * ------------------------------------------------------------------ */
/* Randomness: a seed / seed-word list gives a REPRODUCIBLE xoshiro256** stream -- for test
 * vectors and for mirroring keygen.c's fixed seeds, not cryptographic (the generator's raw
 * outputs are published as the `a` coefficients).  seed == 0 (encrypt_bits, alice) or
 * n_seed_words < 0 (keygen) draws everything from a ChaCha20 stream keyed by getrandom(2).
 * The fresh metadata encryptions of the process contract always use the latter. */
/* raw key material; any output pointer may be NULL to skip it */
int ieache_keygen_raw(const ieache_params* p, const uint32_t* seed_words, int n_seed_words, int32_t* lwe_key /*[n]*/,
                      int32_t* tlwe_key /*[kN]*/, int32_t* bk, int32_t* ksk);
/* keygen.c equivalent: writes secret.key, cloud.key, nbit.key into `dir`
 * (seeds {314,1592,657} / {314,1592,888} as keygen.c:30,34 when seeds are NULL and the
 * counts are >= 0; a count < 0 = kernel entropy for that key set) */
int ieache_keygen_files(const char* dir, const ieache_params* p, const uint32_t* seed, int n_seed,
                        const uint32_t* nbit_seed, int n_nbit_seed);
/* bootsSymEncrypt / bootsSymDecrypt over arrays of bits; rows of n+1 */
int ieache_encrypt_bits(const ieache_params* p, const int32_t* lwe_key, const uint8_t* bits, size_t count,
                        uint64_t seed, int32_t* out);
int ieache_decrypt_bits(const ieache_params* p, const int32_t* lwe_key, const int32_t* samples, size_t count,
                        uint8_t* bits);
/* key files -> raw arrays (sizes from ieache_params; pass NULL to query params only) */
int ieache_read_secret_key(const char* path, ieache_params* p, int32_t* lwe_key, int32_t* tlwe_key);
int ieache_read_cloud_key(const char* path, ieache_params* p, int32_t* bk, int32_t* ksk);
int ieache_write_cloud_key(const char* path, const ieache_params* p, const int32_t* bk, const int32_t* ksk);
int ieache_write_secret_key(const char* path, const ieache_params* p, const int32_t* lwe_key, const int32_t* tlwe_key,
                            const int32_t* bk, const int32_t* ksk);
/* LweSample streams (cloud.data / answer.data): rows of n+1 */
int ieache_read_samples(const char* path, int32_t n, size_t first, size_t count, int32_t* out);
int ieache_write_samples(const char* path, int32_t n, size_t count, const int32_t* rows, int append);
/* alice.c equivalent: one operand -> 352 samples appended to `cloud_data_path`:
 * sign code + bit size under the nbit key, 8 value words + zero carry word
 * under the secret key (alice.c:116-191).  words: 8 x uint32, LSW first. */
int ieache_alice(const char* secret_key_path, const char* nbit_key_path, const char* cloud_data_path, int append,
                 uint32_t sign_code, uint32_t bit_size, const uint32_t* words, uint64_t seed);
/* verif.c decrypt step: answer.data -> sign code, bit size, 8 value words + carry word */
int ieache_verif(const char* secret_key_path, const char* nbit_key_path, const char* answer_data_path,
                 uint32_t* sign_code, uint32_t* bit_size, uint32_t* words9);

/* ------------------------------------------------------------------ *
 * 5. Resident-key daemon.  The reference reloads and     *
 *    re-transforms the cloud key in every ./cloud launch               *
 *    (cloud.c:656-663), i.e. once per operator; its caller             *
 *    (dragonfly_cipher_cloud.py:1233) only needs "run the files in     *
 *    this directory".  ieache_serve keeps the key on the GPU and       *
 *    serves that request over an AF_UNIX stream socket (wire format:   *
 *    csrc/daemon.h); the `cloud` shim forwards to it when              *
 *    IEACHE_DAEMON=<socket path> is set.                               *
 * ------------------------------------------------------------------ */
/* IEACHE_DAEMON_BATCH_WINDOW_MS=T (cloudd --batch-window-ms T; default 0 = one request at a time like the
 * reference): requests arriving within T ms are answered together, and those asking for the same circuit are
 * evaluated as ONE level-batched run -- a lone expression keeps a few of the GPU's 1 024 workgroup slots busy,
 * a hundred concurrent ones fill it.  IEACHE_DAEMON_MAX_BATCH caps a round (default 256). */
/* Blocks.  nbit_key_path may be NULL (= nbit.key next to cloud.key; only
 * RUN_DATA needs it).  max_requests < 0: until a shutdown request or
 * SIGINT/SIGTERM.  Returns the number of requests served or IEACHE_E*. */
int64_t ieache_serve(const char* socket_path, const char* cloud_key_path, const char* nbit_key_path, int device,
                     int64_t max_requests);
/* The same daemon on several GPUs of the node (cloudd --devices 0,1,... or IEACHE_DEVICES=0,1,...): one evaluator per listed
 * device, the cloud key read once and uploaded to each.  A round's same-circuit requests (batch window above) are cut into
 * contiguous slices, one per device -- ieache_shard_slice's rule, the one bench.py and ie-ache_amd/parallel.py apply across
 * ranks (independent expressions, no exchange between GPUs) --, evaluated concurrently and answered in request
 * order; a lone request runs on devices[0].  A device may be listed twice (two contexts on one card).  Answers do not
 * depend on the device list: every expression goes through the same circuit and kernels.
 * Reference caller served: dragonfly_cipher_cloud.py:1233 (one ./cloud per operator; batches come from concurrent clients). */
int64_t ieache_serve_devices(const char* socket_path, const char* cloud_key_path, const char* nbit_key_path, const int* devices,
                             int n_devices, int64_t max_requests);
/* slice [*first, *first + *count) of `total` independent expressions that part `part` of `parts` takes: contiguous, sizes
 * differing by at most one, the first total % parts parts one longer */
int ieache_shard_slice(size_t total, size_t parts, size_t part, size_t* first, size_t* count);
/* clients: return what main() of cloud.c would (0 / 126) or IEACHE_E*;
 * IEACHE_ENODEV when no daemon listens on socket_path */
int ieache_client_ping(const char* socket_path);
int ieache_client_run_dir(const char* socket_path, const char* workdir);
/* cloud.data bytes + operator code in, answer.data bytes out (caller buffer;
 * *answer_len = bytes needed even when answer_cap is too small -> IEACHE_EINVAL) */
int ieache_client_run_data(const char* socket_path, int operator_code, const void* cloud_data, size_t cloud_data_len,
                           void* answer, size_t answer_cap, size_t* answer_len);
int ieache_client_shutdown(const char* socket_path);

#ifdef __cplusplus
}
#endif
#endif
