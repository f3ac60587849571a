"""Resident-key daemon: wire format and both clients, on CPU.

A pure-Python stand-in server speaks csrc/daemon.h's protocol, so the native client
(ieache_client_*), the `cloud` shim's IEACHE_DAEMON switch and the Python client are
exercised without a GPU; the real `cloudd` is covered by the -m gpu tests.
"""
import ctypes as C
import os
import socket
import struct
import subprocess
import threading

import pytest


class FakeDaemon:
    """Answers every request with a scripted (rc, log, data) and records what it received."""

    def __init__(self, path, replies):
        self.path, self.replies, self.seen = str(path), list(replies), []
        self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.sock.bind(self.path)
        self.sock.listen(4)
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _recv(self, c, n):
        buf = b""
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                return None
            buf += chunk
        return buf

    def _run(self):
        for reply in self.replies:
            c, _ = self.sock.accept()
            with c:
                hdr = self._recv(c, 24)
                magic, version, op, flags, plen = struct.unpack("<IIIIQ", hdr)
                payload = self._recv(c, plen) if plen else b""
                self.seen.append((magic, version, op, flags, payload))
                if reply == "garbage":
                    c.sendall(b"\0" * 24)
                    continue
                if reply == "hangup":
                    continue
                rc, log, data = reply
                c.sendall(struct.pack("<IiQQ", 0x43414549, rc, len(log), len(data)) + log + data)
        self.sock.close()

    def join(self):
        self.thread.join(timeout=10)
        assert not self.thread.is_alive()


def test_wire_format_matches_the_header(ia):
    from ieache_amd import daemon
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "ie-ache_amd", "csrc", "daemon.h")).read()
    assert "0x43414549u" in hdr and daemon.MAGIC == 0x43414549 and struct.pack("<I", daemon.MAGIC) == b"IEAC"
    assert "kDaemonVersion = 1" in hdr and daemon.VERSION == 1
    for name, val in (("DAEMON_PING", 1), ("DAEMON_RUN_DIR", 2), ("DAEMON_RUN_DATA", 3), ("DAEMON_SHUTDOWN", 4)):
        assert "%s = %d" % (name, val) in hdr
    assert (daemon.OP_PING, daemon.OP_RUN_DIR, daemon.OP_RUN_DATA, daemon.OP_SHUTDOWN) == (1, 2, 3, 4)
    req = daemon.pack_request(daemon.OP_RUN_DIR, b"/x")
    assert req == struct.pack("<IIIIQ", daemon.MAGIC, 1, 2, 0, 2) + b"/x" and len(req) == 26
    assert daemon.unpack_response_header(struct.pack("<IiQQ", daemon.MAGIC, 126, 3, 4)) == (126, 3, 4)
    with pytest.raises(daemon.DaemonError):
        daemon.unpack_response_header(struct.pack("<IiQQ", 1, 0, 0, 0))
    with pytest.raises(daemon.DaemonError):
        daemon.unpack_response_header(struct.pack("<IiQQ", daemon.MAGIC, 0, daemon.MAX_PAYLOAD + 1, 0))


def test_python_client_roundtrip(ia, tmp_path):
    from ieache_amd import daemon
    sock = tmp_path / "d.sock"
    srv = FakeDaemon(sock, [(0, b"pong", b""), (126, b"Cannot multiply 256 bit number!\n", b""), (0, b"log", b"ANSWER"),
                            "garbage", "hangup", (0, b"bye", b"")])
    assert daemon.ping(sock)[:2] == (0, "pong")
    assert daemon.run_dir(sock, tmp_path) == (126, "Cannot multiply 256 bit number!\n")
    assert daemon.run_data(sock, 4, b"\x01\x02\x03") == (0, "log", b"ANSWER")
    with pytest.raises(daemon.DaemonError):
        daemon.ping(sock)
    with pytest.raises(daemon.DaemonError):
        daemon.ping(sock)
    assert daemon.shutdown(sock) == 0
    srv.join()
    assert [s[2] for s in srv.seen] == [1, 2, 3, 1, 1, 4]
    assert srv.seen[1][4] == os.fsencode(os.path.abspath(tmp_path))
    assert srv.seen[2][4] == struct.pack("<i", 4) + b"\x01\x02\x03"
    with pytest.raises(daemon.DaemonError):  # nobody listens any more
        daemon.ping(tmp_path / "nobody.sock")


def test_native_client_roundtrip(ia, tmp_path, capfd):
    L = ia.lib()
    sock = tmp_path / "d.sock"
    answer = bytes(range(200))
    srv = FakeDaemon(sock, [(0, b"pong", b""), (126, b"chatter\n", b""), (0, b"", answer), (0, b"", answer),
                            (-5, b"cannot open cloud.data", b""), "garbage", (0, b"bye", b"")])
    sp = os.fsencode(sock)
    assert L.ieache_client_ping(sp) == 0
    assert L.ieache_client_run_dir(sp, b"/some/dir") == 126
    buf, need = (C.c_ubyte * 256)(), C.c_size_t(0)
    data = b"cloud-data-bytes"
    assert L.ieache_client_run_data(sp, 2, data, len(data), buf, 256, C.byref(need)) == 0
    assert need.value == 200 and bytes(buf[:200]) == answer
    small = (C.c_ubyte * 10)()
    assert L.ieache_client_run_data(sp, 2, data, len(data), small, 10, C.byref(need)) == -22 and need.value == 200
    assert L.ieache_client_run_dir(sp, b"/some/dir") == -5 and b"cloud.data" in L.ieache_last_error()
    assert L.ieache_client_ping(sp) == -19 and b"reply" in L.ieache_last_error()
    assert L.ieache_client_shutdown(sp) == 0
    srv.join()
    assert srv.seen[1][2:] == (2, 0, b"/some/dir")
    assert srv.seen[2][2:] == (3, 0, struct.pack("<i", 2) + data)
    # error paths that never reach a socket
    assert L.ieache_client_ping(None) == -22
    assert L.ieache_client_run_dir(sp, None) == -22
    assert L.ieache_client_ping(os.fsencode(tmp_path / "nobody.sock")) == -19
    assert b"cannot reach" in L.ieache_last_error()
    assert L.ieache_client_ping(b"/" + b"x" * 200) == -22  # longer than sun_path
    assert L.ieache_serve(None, b"k", None, 0, 0) == -22
    assert L.ieache_serve(os.fsencode(tmp_path / "s.sock"), os.fsencode(tmp_path / "missing.key"), None, 0, 0) == -5


def test_cloud_shim_forwards_to_the_daemon(ia, tmp_path):
    exe = os.path.join(os.path.dirname(ia.library_path()), "cloud")
    sock = tmp_path / "d.sock"
    work = tmp_path / "work"
    work.mkdir()
    srv = FakeDaemon(sock, [(126, b"Cannot multiply 256 bit number!\n", b"")])
    r = subprocess.run([exe], cwd=work, env=dict(os.environ, IEACHE_DAEMON=str(sock)), capture_output=True, timeout=60)
    srv.join()
    assert r.returncode == 126 and b"Cannot multiply" in r.stdout
    assert srv.seen[0][2] == 2 and srv.seen[0][4] == os.fsencode(os.path.realpath(work))
    # no daemon behind the socket path: the shim says so and does the run itself (which fails here: no files)
    r = subprocess.run([exe], cwd=work, env=dict(os.environ, IEACHE_DAEMON=str(tmp_path / "nobody.sock")),
                       capture_output=True, timeout=60)
    assert r.returncode == 1 and b"daemon" in r.stderr and b"cloud.key" in r.stderr


def test_cloudd_fails_loudly_without_a_gpu(ia, tmp_path):
    from ieache_amd import tools
    exe = os.path.join(os.path.dirname(ia.library_path()), "cloudd")
    r = subprocess.run([exe, "--bogus"], capture_output=True, timeout=60)
    assert r.returncode == 2 and b"usage" in r.stderr
    r = subprocess.run([exe, "--socket", str(tmp_path / "s"), "--key", str(tmp_path / "missing.key"), "--max-requests", "0"],
                       capture_output=True, timeout=60)
    assert r.returncode == 1 and b"missing.key" in r.stderr
    tools.keygen_files(tmp_path, ia.default_params().copy(n=6, N=64))
    r = subprocess.run([exe, "--socket", str(tmp_path / "s"), "--key", str(tmp_path / "cloud.key"), "--max-requests", "0"],
                       capture_output=True, timeout=120)
    if ia.device_count() == 0:
        assert r.returncode == 1 and r.stderr.startswith(b"cloudd: ")  # no CPU fallback
    else:
        assert r.returncode == 0 and b"served 0 requests" in r.stdout
    assert not (tmp_path / "s").exists()


def test_device_list_parsing_and_stats_format(ia, tmp_path, monkeypatch):
    """cloudd --devices / IEACHE_DEVICES take a comma-separated list (anything else: exit 2 before any key is read); the client
    parses the STATS line with its per-device job counts; spawn() passes the list on."""
    from ieache_amd import daemon
    exe = os.path.join(os.path.dirname(ia.library_path()), "cloudd")
    for bad in ("0,x", "", "0;1", "-1", "0,,1"):
        r = subprocess.run([exe, "--devices", bad, "--key", str(tmp_path / "missing.key")], capture_output=True, timeout=60)
        assert r.returncode == 2 and b"--devices" in r.stderr, bad
    r = subprocess.run([exe, "--key", str(tmp_path / "missing.key")], env=dict(os.environ, IEACHE_DEVICES="zero"), capture_output=True, timeout=60)
    assert r.returncode == 2 and b"IEACHE_DEVICES" in r.stderr
    # a well-formed list gets as far as the key file (missing here): exit 1, the key's name in the message
    r = subprocess.run([exe, "--devices", "0,0,1", "--socket", str(tmp_path / "s"), "--key", str(tmp_path / "missing.key")], capture_output=True, timeout=60)
    assert r.returncode == 1 and b"missing.key" in r.stderr
    st = daemon.parse_stats("evaluations=3 batched_requests=8 largest_batch=6 devices=2 sharded_evaluations=1 device_jobs=5,3")
    assert st == {"evaluations": 3, "batched_requests": 8, "largest_batch": 6, "devices": 2, "sharded_evaluations": 1, "device_jobs": [5, 3]}
    seen = {}

    class FakeProc:
        returncode = 3

        def __init__(self, cmd, env=None):
            seen["cmd"] = cmd

        def poll(self):
            return 3

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    with pytest.raises(daemon.DaemonError):
        daemon.spawn(tmp_path / "x.sock", tmp_path / "cloud.key", devices=(0, 1, 2, 3, 4, 5, 6, 7), batch_window_ms=50)
    cmd = seen["cmd"]
    assert cmd[cmd.index("--devices") + 1] == "0,1,2,3,4,5,6,7" and "--device" not in cmd and "--batch-window-ms" in cmd
    with pytest.raises(daemon.DaemonError):
        daemon.spawn(tmp_path / "x.sock", tmp_path / "cloud.key", device=2)
    assert seen["cmd"][seen["cmd"].index("--device") + 1] == "2" and "--devices" not in seen["cmd"]
