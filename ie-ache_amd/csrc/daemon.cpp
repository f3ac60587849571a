// Resident-key Cloud daemon and its client (see daemon.h for the wire format).
#include "daemon.h"

#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <tuple>

#include "../../include/ieache.h"
#include "cloud_run.h"
#include "codec.h"
#include "evaluator.h"
#include "tfhe_host.h"

namespace ieache {

namespace {

struct Fd {
    int fd = -1;
    explicit Fd(int f = -1) : fd(f) {}
    Fd(const Fd&) = delete;
    Fd& operator=(const Fd&) = delete;
    ~Fd() { reset(); }
    void reset(int f = -1) {
        if (fd >= 0) close(fd);
        fd = f;
    }
};

bool read_full(int fd, void* buf, size_t n) {
    auto* p = static_cast<unsigned char*>(buf);
    while (n > 0) {
        const ssize_t r = recv(fd, p, n, 0);
        if (r == 0) return false;
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}
bool write_full(int fd, const void* buf, size_t n) {
    auto* p = static_cast<const unsigned char*>(buf);
    while (n > 0) {
        const ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}

#pragma pack(push, 1)
struct ReqHeader {
    uint32_t magic, version, op, flags;
    uint64_t payload_len;
};
struct RespHeader {
    uint32_t magic;
    int32_t rc;
    uint64_t log_len, data_len;
};
#pragma pack(pop)
static_assert(sizeof(ReqHeader) == 24 && sizeof(RespHeader) == 24, "wire headers are packed");

sockaddr_un make_addr(const std::string& path) {
    sockaddr_un a{};
    a.sun_family = AF_UNIX;
    if (path.empty() || path.size() >= sizeof(a.sun_path)) throw std::invalid_argument("socket path empty or longer than 107 bytes");
    memcpy(a.sun_path, path.c_str(), path.size() + 1);
    return a;
}

// identity of a key file: reloading is skipped while it does not change
struct FileId {
    dev_t dev = 0;
    ino_t ino = 0;
    off_t size = -1;
    timespec mtime{};
    bool valid = false;
    static FileId of(const std::string& path) {
        FileId id;
        struct stat st;
        if (stat(path.c_str(), &st) == 0) {
            id.dev = st.st_dev;
            id.ino = st.st_ino;
            id.size = st.st_size;
            id.mtime = st.st_mtim;
            id.valid = true;
        }
        return id;
    }
    bool same(const FileId& o) const {
        return valid && o.valid && dev == o.dev && ino == o.ino && size == o.size && mtime.tv_sec == o.mtime.tv_sec &&
               mtime.tv_nsec == o.mtime.tv_nsec;
    }
};

std::string dirname_of(const std::string& path) {
    const size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string(".") : (s == 0 ? std::string("/") : path.substr(0, s));
}

struct MemStream {  // open_memstream wrapper
    char* buf = nullptr;
    size_t len = 0;
    FILE* f = nullptr;
    MemStream() {
        f = open_memstream(&buf, &len);
        if (!f) throw std::bad_alloc();
    }
    MemStream(const MemStream&) = delete;
    MemStream& operator=(const MemStream&) = delete;
    void finish() {
        if (f) fclose(f);
        f = nullptr;
    }
    ~MemStream() {
        finish();
        free(buf);
    }
};

volatile sig_atomic_t g_stop = 0;
int g_listen_fd = -1;
void on_signal(int) {
    g_stop = 1;
    if (g_listen_fd >= 0) shutdown(g_listen_fd, SHUT_RD);  // wakes accept()
}

class Server {
public:
    explicit Server(const DaemonConfig& cfg) : cfg_(cfg) {
        load_cloud_key_file(cfg.cloud_key_path);
        const std::string nb = cfg.nbit_key_path.empty() ? dirname_of(cfg.cloud_key_path) + "/nbit.key" : cfg.nbit_key_path;
        if (FileId::of(nb).valid) {
            load_secret_key(nb, &nbit_, /*with_cloud=*/false);
            have_nbit_ = true;
        } else if (!cfg.nbit_key_path.empty()) {
            throw CodecError("cannot open " + nb);
        }
    }

    struct Pending {
        uint32_t op = 0;
        std::vector<unsigned char> payload;
        int32_t rc = 0;
        std::string log;
        std::vector<unsigned char> data;
    };

    // Answers every request of `reqs` (in arrival order for everything with side effects).  The value circuits
    // of the RUN_DIR / RUN_DATA requests among them are evaluated together: requests that ask for the same
    // circuit become ONE level-batched evaluation, which is where a GPU's throughput is (a lone 32-bit addition
    // keeps 2-5 of its 1 024 workgroup slots busy).
    void handle_batch(std::vector<Pending>& reqs) {
        std::vector<std::unique_ptr<Run>> runs;
        for (Pending& r : reqs) {
            if (r.op == DAEMON_RUN_DIR || r.op == DAEMON_RUN_DATA) {
                std::unique_ptr<Run> run(new Run);
                run->req = &r;
                try {
                    if (r.op == DAEMON_RUN_DIR) {
                        const std::string dir(r.payload.begin(), r.payload.end());
                        if (dir.empty() || dir.find('\0') != std::string::npos) throw std::invalid_argument("bad directory");
                        // a new session key (dragonfly_public_cloud.py receives cloud.key once per session) is picked up
                        // here; what was prepared under the old key is evaluated first
                        const std::string key = dir + "/cloud.key";
                        const FileId id = FileId::of(key);
                        if (id.valid && !id.same(key_id_)) {
                            evaluate(runs);
                            load_cloud_key_file(key);
                        }
                        fprintf(run->log.f, "Reading the key...\n");
                        run->dir.reset(new CloudDirSession(dir, eval_->params(), run->log.f));
                        cloud_prepare(run->dir->io, &run->job, nullptr);
                    } else {
                        if (!have_nbit_) throw std::invalid_argument("RUN_DATA needs the daemon to hold nbit.key (--nbit)");
                        if (r.payload.size() < 4) throw std::invalid_argument("RUN_DATA payload too short");
                        int32_t op = 0;
                        memcpy(&op, r.payload.data(), 4);
                        if (r.payload.size() == 4) throw CodecError("cloud.data is empty");
                        run->in = fmemopen(r.payload.data() + 4, r.payload.size() - 4, "rb");
                        if (!run->in) throw std::bad_alloc();
                        run->answer.reset(new MemStream);
                        run->io.params = eval_->params();
                        run->io.nbit = &nbit_;
                        run->io.cloud_data = run->in;
                        run->io.op = op;
                        MemStream* ans = run->answer.get();
                        run->io.open_answer = [ans]() -> FILE* { return ans->f; };
                        run->io.log = run->log.f;
                        cloud_prepare(run->io, &run->job, nullptr);
                    }
                    run->prepared = true;
                } catch (...) {
                    run->fail(std::current_exception());
                }
                runs.push_back(std::move(run));
            } else {
                evaluate(runs);  // keep the order of everything that is not a pure computation
                r.rc = handle(r.op, r.payload, &r.log, &r.data);
            }
        }
        evaluate(runs);
    }

    int64_t batches = 0, batched_requests = 0, largest_batch = 0, sharded = 0;

    // returns rc; fills log and data
    int32_t handle(uint32_t op, const std::vector<unsigned char>& payload, std::string* log, std::vector<unsigned char>* data) {
        try {
            switch (op) {
                case DAEMON_PING: *log = std::string(ieache_version()) + ", key " + key_path_; return 0;
                case DAEMON_SHUTDOWN: *log = "bye"; return 0;
                case DAEMON_STATS: {
                    char buf[200];
                    snprintf(buf, sizeof buf, "evaluations=%lld batched_requests=%lld largest_batch=%lld devices=%zu sharded_evaluations=%lld device_jobs=",
                             (long long)batches, (long long)batched_requests, (long long)largest_batch, evals_.size(), (long long)sharded);
                    *log = buf;
                    for (size_t d = 0; d < device_jobs_.size(); d++) *log += (d ? "," : "") + std::to_string(device_jobs_[d]);
                    return 0;
                }
                default: *log = "unknown request"; return IEACHE_EINVAL;
            }
        } catch (const CodecError& e) {
            *log += e.what();
            return IEACHE_EIO;
        } catch (const std::bad_alloc&) {
            *log += "out of host memory";
            return IEACHE_ENOMEM;
        } catch (const std::invalid_argument& e) {
            *log += e.what();
            return IEACHE_EINVAL;
        } catch (const std::exception& e) {
            *log += e.what();
            return IEACHE_ENODEV;
        }
    }

private:
    void load_cloud_key_file(const std::string& path) {
        const FileId id = FileId::of(path);
        if (!id.valid) throw CodecError("cannot open " + path);
        CloudKeyData ck;
        load_cloud_key(path, &ck);  // read and parsed once, whatever the number of devices
        evals_.clear();  // frees the old key's 290 MB per device before the new one is uploaded
        eval_ = nullptr;
        std::vector<int> devs = cfg_.devices.empty() ? std::vector<int>{cfg_.device} : cfg_.devices;
        for (int dev : devs) {
            evals_.emplace_back(new Evaluator(ck.p, dev));
            evals_.back()->load_keys_host(ck.bk.data(), ck.ksk.data());
        }
        eval_ = evals_[0].get();
        device_jobs_.assign(evals_.size(), 0);
        key_id_ = id;
        key_path_ = path;
    }

    // The jobs of one circuit over the daemon's devices: contiguous slices (daemon_shard), one host thread per device that
    // has a slice, each driving its own evaluator (own stream, own key copy); nothing is exchanged between devices.  A
    // failure on any device fails the evaluation (the first exception is rethrown once every thread has finished).
    void eval_sharded(const std::vector<CloudJob*>& jobs, std::vector<std::vector<Torus32>>* outs) {
        const size_t parts = std::min(evals_.size(), jobs.size());
        if (parts <= 1) {
            cloud_eval_jobs(*evals_[0], jobs, outs, nullptr);
            device_jobs_[0] += (int64_t)jobs.size();
            return;
        }
        outs->assign(jobs.size(), {});
        std::vector<std::vector<std::vector<Torus32>>> part_outs(parts);
        std::vector<std::exception_ptr> errors(parts);
        auto run_part = [&](size_t d) {
            try {
                size_t first = 0, count = 0;
                daemon_shard(jobs.size(), parts, d, &first, &count);
                const std::vector<CloudJob*> mine(jobs.begin() + first, jobs.begin() + first + count);
                cloud_eval_jobs(*evals_[d], mine, &part_outs[d], nullptr);
            } catch (...) {
                errors[d] = std::current_exception();
            }
        };
        std::vector<std::thread> threads;
        for (size_t d = 1; d < parts; d++) {
            try {
                threads.emplace_back(run_part, d);
            } catch (const std::system_error&) {
                run_part(d);  // no thread to be had: this slice runs here, after the ones already started
            }
        }
        run_part(0);  // the first slice on the serving thread itself
        for (std::thread& t : threads) t.join();
        for (const std::exception_ptr& e : errors)
            if (e) std::rethrow_exception(e);
        for (size_t d = 0; d < parts; d++) {
            size_t first = 0, count = 0;
            daemon_shard(jobs.size(), parts, d, &first, &count);
            for (size_t i = 0; i < count; i++) (*outs)[first + i] = std::move(part_outs[d][i]);
            device_jobs_[d] += (int64_t)count;
        }
        sharded++;
    }

    // one RUN_DIR / RUN_DATA request on its way through cloud_prepare -> (batched) evaluation -> cloud_finish
    struct Run {
        Pending* req = nullptr;
        MemStream log;
        std::unique_ptr<CloudDirSession> dir;  // RUN_DIR
        FILE* in = nullptr;                    // RUN_DATA: the payload as a stream
        std::unique_ptr<MemStream> answer;     // RUN_DATA: answer.data in memory
        CloudRunIO io;
        CloudJob job;
        bool prepared = false, failed = false;
        ~Run() {
            if (in) fclose(in);
        }
        const CloudRunIO& the_io() const { return dir ? dir->io : io; }
        void fail(std::exception_ptr e) {
            failed = true;
            log.finish();
            req->log.assign(log.buf ? log.buf : "", log.len);
            try {
                std::rethrow_exception(e);
            } catch (const CodecError& x) {
                req->log += x.what();
                req->rc = IEACHE_EIO;
            } catch (const std::bad_alloc&) {
                req->log += "out of host memory";
                req->rc = IEACHE_ENOMEM;
            } catch (const std::invalid_argument& x) {
                req->log += x.what();
                req->rc = IEACHE_EINVAL;
            } catch (const std::exception& x) {
                req->log += x.what();
                req->rc = IEACHE_ENODEV;
            } catch (...) {
                req->log += "unknown failure";
                req->rc = IEACHE_ENODEV;
            }
        }
        void reply_ok(int rc) {
            log.finish();
            req->log.assign(log.buf ? log.buf : "", log.len);
            req->rc = rc;
            if (answer) {
                answer->finish();
                req->data.assign(reinterpret_cast<unsigned char*>(answer->buf), reinterpret_cast<unsigned char*>(answer->buf) + answer->len);
            } else if (dir) {
                dir.reset();  // closes answer.data before the client is told it is there
            }
        }
    };

    static double now_s() {
        struct timeval tv;
        gettimeofday(&tv, nullptr);
        return tv.tv_sec + tv.tv_usec * 1e-6;
    }

    // evaluates everything prepared so far, grouped by circuit, and fills the replies
    void evaluate(std::vector<std::unique_ptr<Run>>& runs) {
        std::map<std::tuple<int32_t, int32_t, bool>, std::vector<Run*>> groups;
        for (auto& r : runs) {
            if (r->failed || !r->prepared) continue;
            if (!r->job.has_circuit) {
                r->reply_ok(r->job.rc);  // 126, or 0 with the 64-sample failure marker: nothing to evaluate
                continue;
            }
            groups[std::make_tuple(r->job.kind, r->job.int_bit, r->job.fold)].push_back(r.get());
        }
        for (auto& g : groups) {
            std::vector<Run*>& members = g.second;
            std::vector<CloudJob*> jobs;
            for (Run* r : members) {
                jobs.push_back(&r->job);
                fprintf(r->log.f, "Doing the homomorphic computation...\n");
            }
            try {
                std::vector<std::vector<Torus32>> outs;
                const double t0 = now_s();
                eval_sharded(jobs, &outs);
                const double dt = now_s() - t0;
                batches++;
                batched_requests += (int64_t)members.size();
                largest_batch = std::max<int64_t>(largest_batch, (int64_t)members.size());
                for (size_t i = 0; i < members.size(); i++) {
                    Run* r = members[i];
                    try {
                        if (members.size() > 1)
                        fprintf(r->log.f, "cloudd: evaluated together with %zu other request(s)%s\n", members.size() - 1,
                                evals_.size() > 1 ? (" on " + std::to_string(std::min(evals_.size(), members.size())) + " devices").c_str() : "");
                        cloud_finish(r->the_io(), r->job, outs[i].data(), outs[i].size() / ((size_t)r->job.params.n + 1), dt);
                        r->reply_ok(0);
                    } catch (...) {
                        r->fail(std::current_exception());
                    }
                }
            } catch (...) {
                for (Run* r : members)
                    if (!r->failed && r->req->log.empty()) r->fail(std::current_exception());
            }
        }
        runs.clear();
    }

    DaemonConfig cfg_;
    std::vector<std::unique_ptr<Evaluator>> evals_;  // one per configured device
    Evaluator* eval_ = nullptr;                      // evals_[0]: parameters, single-request path
    std::vector<int64_t> device_jobs_;
    FileId key_id_;
    std::string key_path_;
    SecretKeyData nbit_;
    bool have_nbit_ = false;
};

}  // namespace

void daemon_shard(size_t total, size_t parts, size_t part, size_t* first, size_t* count) {
    if (parts == 0 || part >= parts) {
        *first = total;
        *count = 0;
        return;
    }
    const size_t base = total / parts, extra = total % parts;
    *first = part * base + std::min(part, extra);
    *count = base + (part < extra ? 1 : 0);
}

int64_t daemon_serve(const DaemonConfig& cfg) {
    sockaddr_un addr = make_addr(cfg.socket_path);
    // refuse to steal a live daemon's socket; clear a stale one
    if (FileId::of(cfg.socket_path).valid) {
        Fd probe(socket(AF_UNIX, SOCK_STREAM, 0));
        if (probe.fd >= 0 && connect(probe.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) == 0)
            throw std::runtime_error("a daemon is already serving " + cfg.socket_path);
        unlink(cfg.socket_path.c_str());
    }
    Server server(cfg);  // key load + spectrum transform happen once, here

    Fd lfd(socket(AF_UNIX, SOCK_STREAM, 0));
    if (lfd.fd < 0) throw std::runtime_error(std::string("socket: ") + strerror(errno));
    const mode_t old = umask(0077);  // the socket hands out work done with a secret key: owner only
    const int brc = bind(lfd.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr);
    umask(old);
    if (brc != 0) throw std::runtime_error("bind " + cfg.socket_path + ": " + strerror(errno));
    if (listen(lfd.fd, 8) != 0) throw std::runtime_error(std::string("listen: ") + strerror(errno));
    g_stop = 0;
    g_listen_fd = lfd.fd;
    struct sigaction sa{}, old_int{}, old_term{};
    sa.sa_handler = on_signal;
    sigaction(SIGINT, &sa, &old_int);
    sigaction(SIGTERM, &sa, &old_term);
    if (cfg.announce) {
        printf("cloudd: ready on %s\n", cfg.socket_path.c_str());
        fflush(stdout);
    }
    int64_t served = 0;
    bool running = true;
    // Requests that arrive within `batch_window_ms` of the first one of a round are answered together (see
    // Server::handle_batch).  0 = one request at a time, like the reference.
    const int window_ms = cfg.batch_window_ms;
    const size_t max_batch = cfg.max_batch > 0 ? (size_t)cfg.max_batch : 1;
    while (running && !g_stop && (cfg.max_requests < 0 || served < cfg.max_requests)) {
        std::vector<std::unique_ptr<Fd>> conns;
        std::vector<Server::Pending> reqs;
        // A request is a header and a payload written back to back by the client library; a peer that connects and then
        // stalls must not hold up the requests already gathered in this round, so every connection reads under a receive
        // deadline (generous for the connection that opens a round -- nobody is waiting yet --, the batching window for
        // the ones that join it) and is dropped when it misses it.  The payloads buffered in one round are capped too.
        size_t round_bytes = 0;
        auto take = [&](int fd, int deadline_ms) {  // reads one request; a client that went away, stalled or sent garbage is answered or dropped here
            std::unique_ptr<Fd> c(new Fd(fd));
            struct timeval tv{deadline_ms / 1000, (deadline_ms % 1000) * 1000};
            (void)setsockopt(c->fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            ReqHeader h{};
            if (!read_full(c->fd, &h, sizeof h)) return;
            Server::Pending p;
            p.op = h.op;
            if (h.magic != kDaemonMagic || h.version != kDaemonVersion) {
                p.op = 0;
                p.rc = IEACHE_EINVAL;
                p.log = "bad magic or protocol version";
            } else if (h.payload_len > kDaemonMaxPayload) {
                p.op = 0;
                p.rc = IEACHE_EINVAL;
                p.log = "payload too large";
            } else if (round_bytes + h.payload_len > kDaemonMaxRoundBytes) {
                p.op = 0;
                p.rc = IEACHE_EINVAL;
                p.log = "cloudd: this round's request buffer is full; send the request again";
            } else {
                try {
                    p.payload.resize((size_t)h.payload_len);
                } catch (const std::bad_alloc&) {
                    return;
                }
                if (h.payload_len && !read_full(c->fd, p.payload.data(), p.payload.size())) return;  // includes a missed deadline
                round_bytes += p.payload.size();
            }
            tv = {0, 0};  // the reply is written without a deadline
            (void)setsockopt(c->fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            conns.push_back(std::move(c));
            reqs.push_back(std::move(p));
        };
        const int first = accept(lfd.fd, nullptr, nullptr);
        if (first < 0) {
            if (errno == EINTR && !g_stop) continue;
            break;
        }
        take(first, kDaemonFirstRecvMs);
        if (window_ms > 0) {
            struct timeval t0;
            gettimeofday(&t0, nullptr);
            while (reqs.size() < max_batch && (cfg.max_requests < 0 || served + (int64_t)reqs.size() < cfg.max_requests)) {
                struct timeval now;
                gettimeofday(&now, nullptr);
                const long spent = (now.tv_sec - t0.tv_sec) * 1000L + (now.tv_usec - t0.tv_usec) / 1000L;
                if (spent >= window_ms) break;
                struct pollfd pfd{lfd.fd, POLLIN, 0};
                const int pr = poll(&pfd, 1, (int)(window_ms - spent));
                if (pr <= 0 || !(pfd.revents & POLLIN)) break;
                const int fd = accept(lfd.fd, nullptr, nullptr);
                if (fd < 0) break;
                take(fd, std::max(kDaemonJoinRecvMinMs, window_ms));
            }
        }
        // requests refused while reading (op 0) keep their canned answer; the rest goes through the server
        std::vector<Server::Pending> work;
        std::vector<size_t> where;
        for (size_t i = 0; i < reqs.size(); i++)
            if (reqs[i].op != 0) {
                where.push_back(i);
                work.push_back(std::move(reqs[i]));
            }
        server.handle_batch(work);
        for (size_t j = 0; j < work.size(); j++) {
            if (work[j].op == DAEMON_SHUTDOWN) running = false;
            reqs[where[j]] = std::move(work[j]);
        }
        for (size_t i = 0; i < reqs.size(); i++) {
            const Server::Pending& p = reqs[i];
            RespHeader r{kDaemonMagic, p.rc, (uint64_t)p.log.size(), (uint64_t)p.data.size()};
            if (write_full(conns[i]->fd, &r, sizeof r) && write_full(conns[i]->fd, p.log.data(), p.log.size()))
                (void)write_full(conns[i]->fd, p.data.data(), p.data.size());
            served++;
        }
    }
    g_listen_fd = -1;
    sigaction(SIGINT, &old_int, nullptr);
    sigaction(SIGTERM, &old_term, nullptr);
    unlink(cfg.socket_path.c_str());
    return served;
}

DaemonReply daemon_request(const std::string& socket_path, uint32_t op, const void* payload, size_t len) {
    sockaddr_un addr = make_addr(socket_path);
    Fd s(socket(AF_UNIX, SOCK_STREAM, 0));
    if (s.fd < 0) throw std::runtime_error(std::string("socket: ") + strerror(errno));
    if (connect(s.fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) != 0)
        throw std::runtime_error("cannot reach the daemon at " + socket_path + ": " + strerror(errno));
    const ReqHeader h{kDaemonMagic, kDaemonVersion, op, 0, (uint64_t)len};
    if (!write_full(s.fd, &h, sizeof h) || (len && !write_full(s.fd, payload, len)))
        throw std::runtime_error("daemon closed the connection while the request was being sent");
    RespHeader r{};
    if (!read_full(s.fd, &r, sizeof r) || r.magic != kDaemonMagic) throw std::runtime_error("no valid reply from the daemon");
    if (r.log_len > kDaemonMaxPayload || r.data_len > kDaemonMaxPayload) throw std::runtime_error("daemon reply too large");
    DaemonReply out;
    out.rc = r.rc;
    out.log.resize((size_t)r.log_len);
    out.data.resize((size_t)r.data_len);
    if ((r.log_len && !read_full(s.fd, &out.log[0], out.log.size())) || (r.data_len && !read_full(s.fd, out.data.data(), out.data.size())))
        throw std::runtime_error("daemon reply truncated");
    return out;
}

}  // namespace ieache
