"""Host-side mirror of the reference's boundary caller.

`compute()` and `compute_final()` follow Cloud/dragonfly_cipher_cloud.py:
1219-1297 and 1300-1327: write operator.txt, run the evaluator on the files in
the working directory, apply the 64-sample failure rule.  The reference does
subprocess.call("./cloud"); here the same contract is one ctypes call into
libieache.so (or the `cloud` executable built next to it, use_subprocess=True; or a
running `cloudd` that already holds the key on the GPU, daemon_socket=...).
"""
import os
import shutil
import subprocess
import time

from .evaluator import check, lib

# answer.data holding only the 64 metadata samples marks a failed computation
# (dragonfly_cipher_cloud.py:1295 hard-codes 162304 = 64 x 2536 for n=630)
FAILURE_SIZE = 162304

_PKG = os.path.dirname(os.path.abspath(__file__))


def compute(operator, workdir=".", ctx=None, use_subprocess=False, failure_size=FAILURE_SIZE, daemon_socket=None):
    """operator: 1 add, 2 subtract, 3 or 4 multiply (both write "4", :1256-1274).

    Returns (exit_code, answer_size, ok).  ok is False when answer.data holds
    64 samples or fewer (the reference then ships the short file and exits).
    failure_size is the reference's hard-coded 64 x 2536 (n=630); tests on other
    parameter sets pass 64 x (4n+16)."""
    code = {1: "1", 2: "2", 3: "4", 4: "4"}.get(int(operator))
    if code is None:
        return None, None, False  # :1286-1287 "else: None"
    with open(os.path.join(workdir, "operator.txt"), "w") as o:
        o.write(code)
    t0 = time.perf_counter()
    if daemon_socket is not None:
        from . import daemon
        rc, _log = daemon.run_dir(daemon_socket, workdir)
        if rc < 0:
            raise RuntimeError("cloudd: %s" % _log)
    elif use_subprocess:
        rc = subprocess.call([os.path.join(_PKG, "cloud")], cwd=workdir)
    elif ctx is not None:
        rc = ctx.cloud_run(workdir)
    else:
        rc = check(lib().ieache_cloud_run(os.fsencode(workdir)))
    with open(os.path.join(workdir, "timings.txt"), "a") as f:  # :1237-1240
        f.write("\nComputation time: ")
        f.write(str(round(time.perf_counter() - t0, 3)))
    ans_size = os.path.getsize(os.path.join(workdir, "answer.data"))
    return rc, ans_size, ans_size > failure_size


def compute_final(operator, workdir=".", flip=True, ctx=None, use_subprocess=False, failure_size=FAILURE_SIZE,
                  daemon_socket=None):
    """Second stage of a 3-operand expression (:1300-1327): cloud.data holds the
    third operand; combine it with the previous answer.data and run again."""
    cloud = os.path.join(workdir, "cloud.data")
    answer = os.path.join(workdir, "answer.data")
    if flip:  # [answer | operand C]
        with open(cloud, "rb") as c, open(answer, "ab") as a:
            shutil.copyfileobj(c, a, 8192)
        shutil.copyfile(answer, cloud)
    else:  # [operand C | answer]
        with open(answer, "rb") as a, open(cloud, "ab") as c:
            shutil.copyfileobj(a, c, 8192)
    os.remove(answer)
    return compute(operator, workdir, ctx=ctx, use_subprocess=use_subprocess, failure_size=failure_size,
                   daemon_socket=daemon_socket)
