"""Client and launcher for the resident-key Cloud daemon (`cloudd`, csrc/daemon.h).

The reference's `compute()` starts `./cloud` once per operator and every launch
reloads ~100 MB of keys (Cloud/cloud.c:656-663).  `cloudd` holds the key on the
GPU; a request is either "run the files in this directory" (what
subprocess.call("./cloud") means, dragonfly_cipher_cloud.py:1233) or "here is
cloud.data + the operator, send answer.data back".  The client below speaks the
wire format directly, so a caller needs no native code at all.
"""
import os
import socket
import struct
import subprocess
import time

MAGIC = 0x43414549  # "IEAC"
VERSION = 1
OP_PING, OP_RUN_DIR, OP_RUN_DATA, OP_SHUTDOWN, OP_STATS = 1, 2, 3, 4, 5
MAX_PAYLOAD = 64 << 20
_REQ = struct.Struct("<IIIIQ")
_RESP = struct.Struct("<IiQQ")

_PKG = os.path.dirname(os.path.abspath(__file__))


class DaemonError(RuntimeError):
    pass


def pack_request(op, payload=b""):
    return _REQ.pack(MAGIC, VERSION, op, 0, len(payload)) + payload


def unpack_response_header(raw):
    magic, rc, log_len, data_len = _RESP.unpack(raw)
    if magic != MAGIC:
        raise DaemonError("no valid reply from the daemon")
    if log_len > MAX_PAYLOAD or data_len > MAX_PAYLOAD:
        raise DaemonError("daemon reply too large")
    return rc, log_len, data_len


def _recv_exact(s, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = s.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise DaemonError("daemon reply truncated")
        buf += chunk
    return bytes(buf)


def request(socket_path, op, payload=b"", timeout=None):
    """One request/response exchange.  Returns (rc, log text, data bytes)."""
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.settimeout(timeout)
    try:
        try:
            s.connect(os.fspath(socket_path))
        except OSError as e:
            raise DaemonError("cannot reach the daemon at %s: %s" % (socket_path, e))
        s.sendall(pack_request(op, payload))
        rc, log_len, data_len = unpack_response_header(_recv_exact(s, _RESP.size))
        log = _recv_exact(s, log_len).decode("utf-8", "replace")
        data = _recv_exact(s, data_len)
        return rc, log, data
    finally:
        s.close()


def ping(socket_path, timeout=5.0):
    return request(socket_path, OP_PING, timeout=timeout)


def run_dir(socket_path, workdir):
    """Same effect as running ./cloud in `workdir`; returns (exit code, stdout chatter)."""
    rc, log, _ = request(socket_path, OP_RUN_DIR, os.fsencode(os.path.abspath(workdir)))
    return rc, log


def run_data(socket_path, operator_code, cloud_data):
    """cloud.data bytes + operator code -> (exit code, chatter, answer.data bytes)."""
    return request(socket_path, OP_RUN_DATA, struct.pack("<i", int(operator_code)) + bytes(cloud_data))


def stats(socket_path):
    """-> dict(evaluations, batched_requests, largest_batch, devices, sharded_evaluations, device_jobs): how the daemon has
    grouped its RUN_* requests so far and, with several devices, how many jobs each one ran (a list)."""
    rc, log, _ = request(socket_path, OP_STATS, timeout=30.0)
    if rc != 0:
        raise DaemonError("stats: %s" % log)
    return parse_stats(log)


def parse_stats(log):
    out = {}
    for kv in log.split():
        k, v = kv.split("=")
        out[k] = [int(x) for x in v.split(",") if x != ""] if k == "device_jobs" else int(v)
    return out


def shutdown(socket_path):
    return request(socket_path, OP_SHUTDOWN, timeout=30.0)[0]


def spawn(socket_path, cloud_key, nbit_key=None, device=0, wait=120.0, env=None, batch_window_ms=0, max_batch=None, devices=None):
    """Start `cloudd` and wait until it answers a ping (the key load + transform take ~0.4 s at n=630).
    batch_window_ms > 0: requests arriving within that window are answered together, same-circuit ones as one
    level-batched GPU run (csrc/daemon.h)."""
    exe = os.path.join(_PKG, "cloudd")
    if not os.path.exists(exe):
        raise DaemonError("%s is missing: run __graft_entry__.build()" % exe)
    cmd = [exe, "--socket", os.fspath(socket_path), "--key", os.fspath(cloud_key)]
    # devices=[0, 1, ...]: one evaluator per listed GPU, a round's same-circuit requests sliced across them (csrc/daemon.h)
    cmd += ["--devices", ",".join(str(int(d)) for d in devices)] if devices else ["--device", str(device)]
    if nbit_key:
        cmd += ["--nbit", os.fspath(nbit_key)]
    if batch_window_ms:
        cmd += ["--batch-window-ms", str(int(batch_window_ms))]
    if max_batch:
        cmd += ["--max-batch", str(int(max_batch))]
    proc = subprocess.Popen(cmd, env=env)
    t0 = time.monotonic()
    while time.monotonic() - t0 < wait:
        if proc.poll() is not None:
            raise DaemonError("cloudd exited with code %d before serving" % proc.returncode)
        if os.path.exists(socket_path):
            try:
                ping(socket_path)
                return proc
            except (DaemonError, OSError):
                pass
        time.sleep(0.05)
    proc.terminate()
    raise DaemonError("cloudd did not come up within %.0f s" % wait)
