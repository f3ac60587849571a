"""Parameter sweep over chunk / slice on the add16x4096 workload (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ieache_amd as ia
from ieache_amd import tools
p = ia.default_params()
k = tools.keygen_raw(p, (1, 2, 3))
ctx = ia.Context.from_arrays(p, k["bk"], k["ksk"])
kind, bits, B = 1, 16, 4096
info = ia.circuit_info(kind, bits)
rng = np.random.default_rng(0)
inb = rng.integers(0, 2, size=(B, info.n_inputs), dtype=np.uint8); inb[:, 2*bits:] = 0
stride = ctx.lwe_stride
d_in = torch.zeros((B, info.n_inputs, stride), dtype=torch.int32, device="cuda")
d_in[:, :, :p.n + 1] = torch.from_numpy(tools.encrypt_bits(p, k["lwe_key"], inb, 5)).cuda()
d_out = torch.zeros((B, info.n_outputs, stride), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
def run(label):
    ctx.eval_batch_device(kind, bits, B, d_in.data_ptr(), d_out.data_ptr())
    st = ia.Stats(); t = time.perf_counter()
    ctx.eval_batch_device(kind, bits, B, d_in.data_ptr(), d_out.data_ptr(), st)
    dt = time.perf_counter() - t
    print(label, "%.0f gates/s  BR %.0f ms KS %.0f ms" % (info.bootstraps * B / dt, st.blind_rotate_ms, st.keyswitch_ms), flush=True)
for chunk in (4096, 8192, 16384):
    ctx.set_option("chunk", chunk)
    for sl in (8, 16, 32):
        ctx.set_option("br_slice", sl)
        run("chunk %5d slice %2d" % (chunk, sl))
