// `cloud` executable shim: same invocation as the reference binary
// (subprocess.call("./cloud"), Cloud/dragonfly_cipher_cloud.py:1233): no
// arguments, files in the current directory, exit code 0 or 126.
#include <cstdio>

#include "../../include/ieache.h"

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    const int rc = ieache_cloud_run(dir);
    if (rc < 0) {
        fprintf(stderr, "cloud: %s\n", ieache_last_error());
        return 1;
    }
    return rc;
}
