"""Where a cold ./cloud (the reference's per-operator subprocess) spends its time (development aid)."""
import sys, os, time, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ieache_amd as ia
from ieache_amd import tools
d = tempfile.mkdtemp(prefix="ieache_cold_")
tools.keygen_files(d)
tools.alice(d, 0, 32, 5, seed=3); tools.alice(d, 0, 32, 7, seed=4, append=True)
open(os.path.join(d, "operator.txt"), "w").write("1")
exe = os.path.join(os.path.dirname(ia.library_path()), "cloud")
for rep in range(3):
    t = time.perf_counter()
    r = subprocess.run([exe], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=dict(os.environ, IEACHE_TIMING="1"))
    print("run %d: %.3f s rc=%d" % (rep, time.perf_counter() - t, r.returncode))
    print(r.stderr)
t = time.perf_counter(); subprocess.run(["/bin/true"]); print("fork+exec of /bin/true: %.4f s" % (time.perf_counter() - t))
