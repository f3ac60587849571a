"""Single-expression latency through the file contract at full parameters (development aid):
what one ./cloud call of the reference costs here."""
import sys, os, time, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ieache_amd as ia
from ieache_amd import tools
d = tempfile.mkdtemp(prefix="ieache_lat_")
t = time.perf_counter(); tools.keygen_files(d); print("keygen (3 files) %.2f s" % (time.perf_counter() - t))
ctx = ia.Context.from_file(os.path.join(d, "cloud.key"))
def run(opname, operator, bits, a, b, env=None):
    tools.alice(d, 0, bits, a, seed=1); tools.alice(d, 0, bits, b, seed=2, append=True)
    for k, v in (env or {}).items(): os.environ[k] = v
    t = time.perf_counter(); rc, size, ok = ia.compute(operator, d, ctx=ctx); dt = time.perf_counter() - t
    for k in (env or {}): os.environ.pop(k)
    code, bs, words = tools.verif(d)
    val = tools.verif_interpret({1: 1, 2: 2, 3: 4}[operator], code, bs, words)
    print("%-28s resident key: %.3f s  -> %d" % (opname, dt, val), flush=True)
    return val
assert run("32-bit A+B", 1, 32, 123456789, 987654321) == 123456789 + 987654321
assert run("32-bit A+B (kogge-stone)", 1, 32, 123456789, 987654321, {"IEACHE_ADDER": "kogge-stone"}) == 123456789 + 987654321
assert run("32-bit A-B", 2, 32, 987654321, 123456789) == 987654321 - 123456789
assert run("32-bit A*B", 3, 32, 123456789, 987654321) == 123456789 * 987654321
assert run("64-bit A*B", 3, 64, 2**62 + 12345, 2**61 + 777) == (2**62 + 12345) * (2**61 + 777)
W = {"IEACHE_MULTIPLIER": "wallace"}
assert run("32-bit A*B (carry-save)", 3, 32, 123456789, 987654321, W) == 123456789 * 987654321
assert run("64-bit A*B (carry-save)", 3, 64, 2**62 + 12345, 2**61 + 777, W) == (2**62 + 12345) * (2**61 + 777)
assert run("128-bit A*B", 3, 128, 2**126 + 12345, 2**125 + 777) == (2**126 + 12345) * (2**125 + 777)
assert run("128-bit A*B (carry-save)", 3, 128, 2**126 + 12345, 2**125 + 777, W) == (2**126 + 12345) * (2**125 + 777)
# cold: the `cloud` executable as the reference runs it (key load + upload + spectrum transform included)
tools.alice(d, 0, 32, 5, seed=3); tools.alice(d, 0, 32, 7, seed=4, append=True)
open(os.path.join(d, "operator.txt"), "w").write("1")
t = time.perf_counter(); rc = subprocess.call([os.path.join(os.path.dirname(ia.library_path()), "cloud")], cwd=d, stdout=subprocess.DEVNULL)
print("cold ./cloud 32-bit A+B (incl. 114 MB key load): %.3f s rc=%d" % (time.perf_counter() - t, rc))
# the same executable handing over to a resident-key daemon (IEACHE_DAEMON): no key load per operator
from ieache_amd import daemon
del ctx
sock = os.path.join(d, "cloudd.sock")
t = time.perf_counter(); proc = daemon.spawn(sock, os.path.join(d, "cloud.key")); print("cloudd start (key load + transform) %.2f s" % (time.perf_counter() - t))
for rep in range(2):
    t = time.perf_counter(); rc = subprocess.call([os.path.join(os.path.dirname(ia.library_path()), "cloud")], cwd=d, stdout=subprocess.DEVNULL, env=dict(os.environ, IEACHE_DAEMON=sock))
    print("./cloud via cloudd 32-bit A+B: %.3f s rc=%d" % (time.perf_counter() - t, rc))
assert tools.verif_interpret(1, *tools.verif(d)) == 12
raw = open(os.path.join(d, "cloud.data"), "rb").read()
t = time.perf_counter(); rc, log, ans = daemon.run_data(sock, 1, raw); print("RUN_DATA over the socket 32-bit A+B: %.3f s rc=%d (%d B in, %d B out)" % (time.perf_counter() - t, rc, len(raw), len(ans)))
daemon.shutdown(sock); proc.wait(timeout=60)
