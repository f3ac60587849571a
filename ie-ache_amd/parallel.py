"""Multi-GPU plumbing for the evaluator: one process per GPU, expressions
sharded across ranks, keys broadcast once.

The path has NO data-path collective: batch elements are independent
expressions and the cloud key is read-only, so after the one-time broadcast of
BK/KSK (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests) ranks never
talk again until results are gathered on the host side.
"""
import os

import numpy as np


def init_distributed(backend=None):
    """-> (rank, world, local_rank, dist-or-None) from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # IEACHE_DIST_SINGLE=1: a process group of ONE rank, so that a one-GPU box runs the same collective calls
    # (RCCL initialisation with device_id, device-side broadcast / all_gather / all_reduce, barrier) the N > 1 job makes
    single = world == 1 and os.environ.get("IEACHE_DIST_SINGLE") == "1"
    if world == 1 and not single:
        return rank, world, local_rank, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if single and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank, dist


def shard_slice(total, rank, world):
    """Contiguous total/world slice of the expression batch owned by `rank`
    (the first total % world ranks take one extra)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def broadcast_cloud_key(params, keys, device, dist, src=0):
    """One-time replication of the raw cloud key (+ the LWE secret key used to
    make synthetic inputs in benchmarks) from rank `src`.

    keys: dict(bk, ksk, lwe_key) of numpy int32 on `src`, None elsewhere.
    Returns torch int32 tensors (bk, ksk, lwe_key) on `device`."""
    import torch
    rank = dist.get_rank() if dist is not None else 0
    if rank == src:
        out = [torch.from_numpy(np.ascontiguousarray(keys[k]).reshape(-1)).to(device) for k in ("bk", "ksk", "lwe_key")]
    else:
        out = [torch.empty(c, dtype=torch.int32, device=device) for c in (params.bk_count, params.ksk_count, params.n)]
    if dist is not None:
        for t in out:
            dist.broadcast(t, src)
    return out


def gather_to_rank0(dist, local):
    """Host-side gather of per-rank result arrays (numpy, concatenated along axis 0) on rank 0."""
    if dist is None:
        return local
    objs = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local, objs, dst=0)
    if dist.get_rank() != 0:
        return None
    return np.concatenate(objs, axis=0)


def eval_batch_sharded(ctx, kind, bits, in_lwe, dist, src=0, stats=None):
    """One circuit over a batch of expressions spread over the ranks of a torchrun job (one rank per GPU).

    in_lwe: [batch][n_inputs][n+1] int32 on rank `src`; the other ranks pass None.  Rank `src` deals out contiguous slices
    (shard_slice; host-side scatter, there is no device collective on the data path), every rank evaluates its own slice
    on its own context -- independent expressions, nothing exchanged -- and rank `src` receives the results concatenated in
    batch order ([batch][n_outputs][n+1]); the others return None.  dist None: ctx.eval_batch.  A rank whose slice is empty
    (batch < world) evaluates nothing.  The output does not depend on the number of ranks."""
    if dist is None:
        return ctx.eval_batch(kind, bits, in_lwe, stats)
    rank, world = dist.get_rank(), dist.get_world_size()
    parts = None
    if rank == src:
        in_lwe = np.ascontiguousarray(in_lwe, dtype=np.int32)
        parts = [in_lwe[shard_slice(in_lwe.shape[0], r, world)] for r in range(world)]
    mine = [None]
    dist.scatter_object_list(mine, parts, src=src)
    mine = mine[0]
    local = ctx.eval_batch(kind, bits, mine, stats) if mine.shape[0] else None
    objs = [None] * world if rank == src else None
    dist.gather_object(local, objs, dst=src)
    if rank != src:
        return None
    return np.concatenate([o for o in objs if o is not None], axis=0)
