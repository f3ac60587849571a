"""The N>1 path on CPU: two -- and eight, the node size the job is written for -- gloo ranks broadcast the cloud key,
shard a batch of expressions with no data-path collective, and gather on the host.  (On the GPU
box the same functions run over RCCL; there is nothing else between ranks.)"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import hashlib, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import ieache_amd as ia
    from ieache_amd import parallel, tools
    rank, world, local_rank, dist = parallel.init_distributed("gloo")
    assert world == int(os.environ["WORLD_SIZE"]) and dist is not None
    p = ia.default_params().copy(n=7, N=32)
    keys = tools.keygen_raw(p, (9, 9, 9)) if rank == 0 else None
    bk, ksk, lwe = parallel.broadcast_cloud_key(p, keys, torch.device("cpu"), dist)
    ref = tools.keygen_raw(p, (9, 9, 9))   # every rank can recompute what rank 0 sent
    assert np.array_equal(bk.numpy(), ref["bk"].ravel()) and np.array_equal(ksk.numpy(), ref["ksk"].ravel())
    assert np.array_equal(lwe.numpy(), ref["lwe_key"])
    # shard 13 expressions (world 2) / 29 (world 8: ragged slices of 3 and 4) of a 16-bit ADD; each rank evaluates only its own slice
    total, bits = (13 if world == 2 else 29), 16
    rng = np.random.default_rng(0)   # same operands on both ranks
    a = rng.integers(0, 1 << bits, size=total); b = rng.integers(0, 1 << bits, size=total)
    sl = parallel.shard_slice(total, rank, world)
    info = ia.circuit_info(ia.CIRC_ADD, bits)
    local = []
    for e in range(sl.start, sl.stop):
        x = np.zeros(info.n_inputs, dtype=np.uint8)
        x[:bits] = tools.int_to_bits(int(a[e]), bits); x[bits:2 * bits] = tools.int_to_bits(int(b[e]), bits)
        local.append(tools.bits_to_int(ia.circuit_simulate(ia.CIRC_ADD, bits, x)))
    got = parallel.gather_to_rank0(dist, np.array(local, dtype=np.int64))
    if rank == 0:
        assert got.tolist() == [int((a[e] + b[e]) %% (1 << bits)) for e in range(total)]
        print("MULTIRANK_OK", sl, flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_shard_slices_cover_batch_exactly():
    sys.path.insert(0, ROOT)
    from ieache_amd.parallel import shard_slice
    for total in (0, 1, 7, 8, 1024, 1025, 8192):
        for world in (1, 2, 3, 8):
            sl = [shard_slice(total, r, world) for r in range(world)]
            assert sl[0].start == 0 and sl[-1].stop == total
            assert all(sl[i].stop == sl[i + 1].start for i in range(world - 1))
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1
    assert shard_slice(1024, 3, 8) == slice(384, 512)  # BASELINE config 4: 128 expressions per GPU


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_broadcast_and_shard(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
        outs.append(out)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    assert "MULTIRANK_OK" in outs[0]


def test_single_rank_process_group_option(tmp_path):
    """IEACHE_DIST_SINGLE=1: a process group of one rank (how a one-GPU box rehearses the collective calls on RCCL);
    here on gloo: the broadcast and the host-side gather go through torch.distributed instead of being skipped."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import ieache_amd as ia\n"
        "from ieache_amd import parallel\n"
        "import torch\n"
        "rank, world, local, dist = parallel.init_distributed('gloo')\n"
        "assert (rank, world) == (0, 1) and dist is not None and dist.is_initialized() and dist.get_world_size() == 1\n"
        "p = ia.default_params().copy(n=7, N=32)\n"
        "keys = {'bk': np.arange(p.bk_count, dtype=np.int32), 'ksk': np.arange(p.ksk_count, dtype=np.int32), 'lwe_key': np.ones(p.n, dtype=np.int32)}\n"
        "bk, ksk, key = parallel.broadcast_cloud_key(p, keys, torch.device('cpu'), dist)\n"
        "assert bk.numel() == p.bk_count and int(bk[-1]) == p.bk_count - 1 and int(key.sum()) == p.n\n"
        "g = parallel.gather_to_rank0(dist, np.arange(6).reshape(3, 2))\n"
        "assert g.shape == (3, 2)\n"
        "dist.destroy_process_group()\n"
        "print('single-rank ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=tmp_path,
                       env=dict(env, IEACHE_DIST_SINGLE="1"))
    assert r.returncode == 0 and "single-rank ok" in r.stdout, r.stderr[-2000:]


SHARDED_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from ieache_amd import parallel
    rank, world, local_rank, dist = parallel.init_distributed("gloo")

    class StandIn:  # what parallel.eval_batch_sharded needs of a Context; stamps each output row with the rank that made it
        calls = 0
        def eval_batch(self, kind, bits, in_lwe, stats=None):
            StandIn.calls += 1
            assert in_lwe.dtype == np.int32 and in_lwe.ndim == 3
            out = in_lwe[:, :2, :] * 3 + kind
            out[:, :, -1] = rank
            return out

    ctx = StandIn()
    for total in (11, 1):   # 11: slices of 6 and 5 (world 2) / 4, 4, 3 (world 3); 1: the other ranks' slices are empty
        full = np.arange(total * 5 * 4, dtype=np.int32).reshape(total, 5, 4) if rank == 0 else None
        got = parallel.eval_batch_sharded(ctx, 7, 16, full, dist)
        if rank == 0:
            exp = full[:, :2, :] * 3 + 7
            assert got.shape == exp.shape and np.array_equal(got[:, :, :-1], exp[:, :, :-1])
            owners = [r for r in range(world) for _ in range(parallel.shard_slice(total, r, world).stop - parallel.shard_slice(total, r, world).start)]
            assert got[:, 0, -1].tolist() == owners, (got[:, 0, -1].tolist(), owners)
        else:
            assert got is None
    assert StandIn.calls == (2 if rank == 0 else 1)   # an empty slice evaluates nothing
    if rank == 0:
        print("SHARDED_OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_eval_batch_sharded_deals_slices_and_reassembles_in_order(tmp_path, world):
    """parallel.eval_batch_sharded on gloo ranks with a stand-in context: contiguous slices out, results back in batch order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "sharded_worker.py"
    script.write_text(SHARDED_WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
        outs.append(out)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    assert "SHARDED_OK" in outs[0]


def test_native_shard_rule_matches_the_python_one(ia):
    """The daemon's per-device slicing (ieache_shard_slice = daemon_shard, csrc/daemon.cpp) is parallel.shard_slice."""
    import ctypes as C
    from ieache_amd.parallel import shard_slice
    L = ia.lib()
    for total in (0, 1, 5, 16, 17, 255, 1024):
        for parts in (1, 2, 3, 8):
            for part in range(parts):
                first, count = C.c_size_t(0), C.c_size_t(0)
                assert L.ieache_shard_slice(total, parts, part, C.byref(first), C.byref(count)) == 0
                sl = shard_slice(total, part, parts)
                assert (first.value, count.value) == (sl.start, sl.stop - sl.start)
    first, count = C.c_size_t(0), C.c_size_t(0)
    assert L.ieache_shard_slice(4, 2, 2, C.byref(first), C.byref(count)) < 0 and L.ieache_shard_slice(4, 0, 0, C.byref(first), C.byref(count)) < 0
