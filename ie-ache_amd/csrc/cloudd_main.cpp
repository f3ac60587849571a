// `cloudd`: the Cloud evaluator as a resident-key daemon.
//   cloudd [--socket PATH] [--key cloud.key] [--nbit nbit.key] [--device N] [--max-requests K]
//          [--batch-window-ms T] [--max-batch B]
// --batch-window-ms T: requests arriving within T ms of each other are answered together; those asking for the same
// circuit are evaluated as one level-batched GPU run (default 0: one request at a time, like the reference).
// Loads the cloud key once (the reference does it per operator, Cloud/cloud.c:656-663), then
// serves `cloud` shim / ieache_client_* requests on an AF_UNIX socket until SIGTERM or a
// shutdown request.  Defaults: ./cloudd.sock, ./cloud.key, nbit.key next to the cloud key.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/ieache.h"

int main(int argc, char** argv) {
    std::string sock = "cloudd.sock", key = "cloud.key", nbit;
    int device = 0;
    long long max_requests = -1;
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
        else if (a == "--device") device = atoi(need("--device"));
        else if (a == "--max-requests") max_requests = atoll(need("--max-requests"));
        else if (a == "--batch-window-ms") setenv("IEACHE_DAEMON_BATCH_WINDOW_MS", need("--batch-window-ms"), 1);
        else if (a == "--max-batch") setenv("IEACHE_DAEMON_MAX_BATCH", need("--max-batch"), 1);
        else {
            fprintf(stderr, "usage: cloudd [--socket PATH] [--key cloud.key] [--nbit nbit.key] [--device N] [--max-requests K] "
                            "[--batch-window-ms T] [--max-batch B]\n");
            return a == "--help" || a == "-h" ? 0 : 2;
        }
    }
    const long long served = ieache_serve(sock.c_str(), key.c_str(), nbit.empty() ? nullptr : nbit.c_str(), device, max_requests);
    if (served < 0) {
        fprintf(stderr, "cloudd: %s\n", ieache_last_error());
        return 1;
    }
    printf("cloudd: served %lld requests\n", served);
    return 0;
}
