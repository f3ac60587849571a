// Resident-key Cloud daemon and its client (see daemon.h for the wire format).
#include "daemon.h"

#include <signal.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>

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

    // returns rc; fills log and data
    int32_t handle(uint32_t op, const std::vector<unsigned char>& payload, std::string* log, std::vector<unsigned char>* data) {
        try {
            switch (op) {
                case DAEMON_PING: *log = std::string(ieache_version()) + ", key " + key_path_; return 0;
                case DAEMON_SHUTDOWN: *log = "bye"; return 0;
                case DAEMON_RUN_DIR: return run_dir(std::string(payload.begin(), payload.end()), log);
                case DAEMON_RUN_DATA: return run_data(payload, log, data);
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
        load_cloud_key(path, &ck);
        eval_.reset();  // frees the old key's 290 MB before the new one is uploaded
        eval_.reset(new Evaluator(ck.p, cfg_.device));
        eval_->load_keys_host(ck.bk.data(), ck.ksk.data());
        key_id_ = id;
        key_path_ = path;
    }

    int32_t run_dir(const std::string& dir, std::string* log) {
        if (dir.empty() || dir.find('\0') != std::string::npos) throw std::invalid_argument("bad directory");
        // a new session key (dragonfly_public_cloud.py receives cloud.key once per session) is picked up here
        const std::string key = dir + "/cloud.key";
        const FileId id = FileId::of(key);
        if (id.valid && !id.same(key_id_)) load_cloud_key_file(key);
        MemStream out;
        const int rc = cloud_run(dir, eval_.get(), nullptr, cfg_.device, out.f);
        out.finish();
        log->assign(out.buf, out.len);
        return rc;
    }

    int32_t run_data(const std::vector<unsigned char>& payload, std::string* log, std::vector<unsigned char>* data) {
        if (!have_nbit_) throw std::invalid_argument("RUN_DATA needs the daemon to hold nbit.key (--nbit)");
        if (payload.size() < 4) throw std::invalid_argument("RUN_DATA payload too short");
        int32_t op = 0;
        memcpy(&op, payload.data(), 4);
        FILE* in = fmemopen(const_cast<unsigned char*>(payload.data()) + 4, payload.size() - 4, "rb");
        if (!in && payload.size() > 4) throw std::bad_alloc();
        struct InCloser {
            FILE* f;
            ~InCloser() {
                if (f) fclose(f);
            }
        } in_closer{in};
        if (!in) throw CodecError("cloud.data is empty");
        MemStream out, answer;
        CloudRunIO io;
        io.params = eval_->params();
        io.nbit = &nbit_;
        io.cloud_data = in;
        io.op = op;
        io.open_answer = [&]() -> FILE* { return answer.f; };
        io.log = out.f;
        Evaluator* e = eval_.get();
        int rc;
        try {
            rc = cloud_run_io(io, [e]() { return e; }, nullptr);
        } catch (...) {
            out.finish();
            log->assign(out.buf, out.len);
            throw;
        }
        out.finish();
        answer.finish();
        log->assign(out.buf, out.len);
        data->assign(reinterpret_cast<unsigned char*>(answer.buf), reinterpret_cast<unsigned char*>(answer.buf) + answer.len);
        return rc;
    }

    DaemonConfig cfg_;
    std::unique_ptr<Evaluator> eval_;
    FileId key_id_;
    std::string key_path_;
    SecretKeyData nbit_;
    bool have_nbit_ = false;
};

}  // namespace

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
    while (running && !g_stop && (cfg.max_requests < 0 || served < cfg.max_requests)) {
        Fd c(accept(lfd.fd, nullptr, nullptr));
        if (c.fd < 0) {
            if (errno == EINTR && !g_stop) continue;
            break;
        }
        ReqHeader h{};
        std::string log;
        std::vector<unsigned char> payload, data;
        int32_t rc;
        if (!read_full(c.fd, &h, sizeof h)) continue;  // client went away
        if (h.magic != kDaemonMagic || h.version != kDaemonVersion) {
            rc = IEACHE_EINVAL;
            log = "bad magic or protocol version";
        } else if (h.payload_len > kDaemonMaxPayload) {
            rc = IEACHE_EINVAL;
            log = "payload too large";
        } else {
            bool ok = true;
            try {
                payload.resize((size_t)h.payload_len);
            } catch (const std::bad_alloc&) {
                ok = false;
            }
            if (!ok || (h.payload_len && !read_full(c.fd, payload.data(), payload.size()))) continue;
            rc = server.handle(h.op, payload, &log, &data);
            if (h.op == DAEMON_SHUTDOWN) running = false;
        }
        RespHeader r{kDaemonMagic, rc, (uint64_t)log.size(), (uint64_t)data.size()};
        if (write_full(c.fd, &r, sizeof r) && write_full(c.fd, log.data(), log.size())) (void)write_full(c.fd, data.data(), data.size());
        served++;
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
