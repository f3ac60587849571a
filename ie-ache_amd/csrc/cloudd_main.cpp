// `cloudd`: the Cloud evaluator as a resident-key daemon.
//   cloudd [--socket PATH] [--key cloud.key] [--nbit nbit.key] [--device N | --devices N,M,...] [--max-requests K]
//          [--batch-window-ms T] [--max-batch B]
// --devices 0,1,...,7 (or IEACHE_DEVICES): one evaluator per listed GPU, the key loaded once and uploaded to each; the
// same-circuit requests of a batching round are cut into contiguous slices, one per GPU, and answered in request order.
// --batch-window-ms T: requests arriving within T ms of each other are answered together; those asking for the same
// circuit are evaluated as one level-batched GPU run (default 0: one request at a time, like the reference).
// Loads the cloud key once (the reference does it per operator, Cloud/cloud.c:656-663), then
// serves `cloud` shim / ieache_client_* requests on an AF_UNIX socket until SIGTERM or a
// shutdown request.  Defaults: ./cloudd.sock, ./cloud.key, nbit.key next to the cloud key.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ieache.h"

// "0,1,3" -> {0, 1, 3}; false on anything else
static bool parse_devices(const char* text, std::vector<int>* out) {
    out->clear();
    const char* p = text;
    while (*p) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v > 1023) return false;
        out->push_back((int)v);
        p = end;
        if (*p == ',') p++;
        else if (*p) return false;
    }
    return !out->empty();
}

int main(int argc, char** argv) {
    std::string sock = "cloudd.sock", key = "cloud.key", nbit;
    std::vector<int> devices;
    long long max_requests = -1;
    if (const char* e = getenv("IEACHE_DEVICES")) {
        if (*e && !parse_devices(e, &devices)) {
            fprintf(stderr, "cloudd: IEACHE_DEVICES wants a comma-separated list of device numbers\n");
            return 2;
        }
    }
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "cloudd: %s needs a value\n", what);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--socket") sock = need("--socket");
        else if (a == "--key") key = need("--key");
        else if (a == "--nbit") nbit = need("--nbit");
        else if (a == "--device") devices.assign(1, atoi(need("--device")));
        else if (a == "--devices") {
            if (!parse_devices(need("--devices"), &devices)) {
                fprintf(stderr, "cloudd: --devices wants a comma-separated list of device numbers\n");
                return 2;
            }
        }
        else if (a == "--max-requests") max_requests = atoll(need("--max-requests"));
        else if (a == "--batch-window-ms") setenv("IEACHE_DAEMON_BATCH_WINDOW_MS", need("--batch-window-ms"), 1);
        else if (a == "--max-batch") setenv("IEACHE_DAEMON_MAX_BATCH", need("--max-batch"), 1);
        else {
            fprintf(stderr, "usage: cloudd [--socket PATH] [--key cloud.key] [--nbit nbit.key] [--device N | --devices N,M,...] [--max-requests K] "
                            "[--batch-window-ms T] [--max-batch B]\n");
            return a == "--help" || a == "-h" ? 0 : 2;
        }
    }
    if (devices.empty()) devices.push_back(0);
    const long long served = ieache_serve_devices(sock.c_str(), key.c_str(), nbit.empty() ? nullptr : nbit.c_str(), devices.data(),
                                                  (int)devices.size(), max_requests);
    if (served < 0) {
        fprintf(stderr, "cloudd: %s\n", ieache_last_error());
        return 1;
    }
    printf("cloudd: served %lld requests\n", served);
    return 0;
}
