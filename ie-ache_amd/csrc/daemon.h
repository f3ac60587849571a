// Resident-key Cloud daemon: the reference reloads and re-transforms ~100 MB of
// keys on every `./cloud` launch (Cloud/cloud.c:656-663, once per operator of every expression).
// Here one process keeps the cloud key on the GPU and serves the same contract over a local
// (AF_UNIX) stream socket: either "run the files in this directory" or "here is cloud.data and
// the operator, send answer.data back".  One request at a time, like the reference -- or, with a batching
// window (DaemonConfig::batch_window_ms, IEACHE_DAEMON_BATCH_WINDOW_MS), the requests of several clients at once:
// those asking for the same circuit are evaluated as one level-batched launch sequence, which is what fills a GPU.
//
// Wire format (little-endian, no padding):
//   request : u32 magic 'IEAC' | u32 version (1) | u32 op | u32 flags (0) | u64 payload_len | payload
//   response: u32 magic 'IEAC' | i32 rc | u64 log_len | u64 data_len | log bytes | data bytes
//   op PING     : no payload
//   op RUN_DIR  : payload = directory path; the daemon reads nbit.key, cloud.data, operator.txt there
//                 and writes answer.data (+ averagestandard.txt), exactly like `./cloud`; if that
//                 directory's cloud.key is not the resident one, it is loaded first
//   op RUN_DATA : payload = i32 operator | cloud.data bytes; response data = answer.data bytes
//                 (metadata key = the daemon's nbit key)
//   op SHUTDOWN : no payload; the daemon answers and exits its loop
//   op STATS    : no payload; log = "evaluations=E batched_requests=R largest_batch=B devices=D sharded_evaluations=S
//                 device_jobs=j0,j1,..." (S: evaluations whose jobs were split over more than one device; j_i: jobs device i ran)
//   rc: 0 or 126 as main() of cloud.c, negative IEACHE_E* on failure (message in log)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ieache {

constexpr uint32_t kDaemonMagic = 0x43414549u;  // "IEAC"
constexpr uint32_t kDaemonVersion = 1;
enum DaemonOp : uint32_t { DAEMON_PING = 1, DAEMON_RUN_DIR = 2, DAEMON_RUN_DATA = 3, DAEMON_SHUTDOWN = 4, DAEMON_STATS = 5 };
constexpr uint64_t kDaemonMaxPayload = 64ull << 20;  // cloud.data is 1.8 MB at n=630
constexpr uint64_t kDaemonMaxRoundBytes = 1ull << 30;  // request payloads buffered in one batching round (256 x 1.8 MB = 0.45 GB)
// receive deadlines: the connection that opens a round / one that joins a batching round (at least this, else the window)
constexpr int kDaemonFirstRecvMs = 5000, kDaemonJoinRecvMinMs = 250;

struct DaemonConfig {
    std::string socket_path;
    std::string cloud_key_path;  // loaded before the first accept
    std::string nbit_key_path;   // for RUN_DATA; empty = nbit.key next to cloud.key
    int device = 0;
    // Several GPUs (cloudd --devices 0,1,... / IEACHE_DEVICES): one evaluator per listed device, the cloud key read from
    // disk once and uploaded to each; the same-circuit jobs of a round are cut into contiguous slices, one per device
    // (daemon_shard, the rule ie-ache_amd/parallel.py's shard_slice applies across ranks), evaluated concurrently -- no
    // exchange between devices -- and answered in request order.  A device may be listed twice (two contexts on one card:
    // how the one-GPU test box exercises this path).  Empty = {device}.
    std::vector<int> devices;
    int64_t max_requests = -1;   // < 0: until SHUTDOWN
    bool announce = true;        // print "cloudd: ready on <path>" once listening
    // Requests arriving within this many milliseconds of the first one of a round are answered together, those that
    // ask for the same circuit as ONE level-batched evaluation.  0 = one request at a time.
    int batch_window_ms = 0;
    int max_batch = 256;
};

// Contiguous slice [first, first + count) of `total` jobs that part `part` of `parts` takes: sizes differ by at most one, the
// first total % parts parts take the extra job (= parallel.shard_slice).
void daemon_shard(size_t total, size_t parts, size_t part, size_t* first, size_t* count);

// Blocks serving requests; returns the number served.  Throws on setup failure
// (key load, GPU, bind).  A failing request is answered with a negative rc and
// the daemon keeps running.
int64_t daemon_serve(const DaemonConfig& cfg);

struct DaemonReply {
    int32_t rc = 0;
    std::string log;
    std::vector<unsigned char> data;
};
// Client side.  Throws std::runtime_error when the daemon cannot be reached or
// answers garbage.
DaemonReply daemon_request(const std::string& socket_path, uint32_t op, const void* payload, size_t len);

}  // namespace ieache
