"""bench.py's host-side arithmetic (no GPU): the algorithmic flop / byte model of SURVEY.md 8(d), the roofline record
built from a synthetic Stats, and the self-launch command line."""
import importlib.util
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_model_matches_the_survey(bench, ia):
    p = ia.default_params()
    # 8 transforms x 5 * 512 * 9 + 12 x 512 complex MACs x 8 flop per CMux step, n = 630 steps
    assert bench.algorithmic_flops_per_gate(p) == 630 * (8 * 5 * 512 * 9 + 12 * 512 * 8) == 630 * 233472
    bk, ksk, io = bench.algorithmic_bytes(p)
    assert (bk, int(ksk), io) == (30965760, 15507456, 7572) and bk + int(ksk) + io == 46480788
    # the provably exact two-limb product (the `exact` leg): forward transforms shared, inverse transforms and MACs doubled
    assert bench.algorithmic_flops_per_gate(p, limbs=2) == 630 * (10 * 5 * 512 * 9 + 24 * 512 * 8) == 630 * 328704


def test_roofline_record_from_synthetic_stats(bench, ia):
    p = ia.default_params()
    # 8 192 gates advanced through one whole rotation in 40 launches of 1 ms: 204 800 gates/s of blind rotation
    st = types.SimpleNamespace(blind_rotate_ms=40.0, blind_rotate_launches=40, keyswitch_ms=0.7, keyswitch_launches=1,
                               bootstraps=8192, chunks=1, total_ms=40.8)
    pmc = {"kernel": "k", "valu_insts_per_gate_step": 3158.0, "fp64_insts_per_gate_step": 2493, "hbm_bytes_per_gate_step": 1264.0,
           "gates_per_launch": 8192, "cmux_steps_per_launch": 16, "shader_cycles_per_gate_step": 17.08, "source": "x",
           "rocprof_avg_launch_ms": 0.82, "rocprof_stats": "y"}
    r = bench.roofline(p, st, 8192 / 40.8e-3, pmc)
    rate = 8192 / 40e-3
    assert r["bound"] == "fp64_valu" and r["peak"] == 78.6 and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - rate * 630 * 233472 * 1e-12) < 1e-9 and abs(r["frac"] - r["achieved"] / 78.6) < 1e-12
    assert r["frac"] == r["frac_algorithmic_flops"] and 0.35 < r["frac"] < 0.45
    assert abs(r["cmux_steps_per_launch"] - 15.75) < 1e-9 and r["gates_per_launch"] == 8192
    vi = r["valu_issue"]
    assert abs(vi["utilisation"] - rate * 3158.0 * 630 / (256 * 4 * 2.4e9 / 4)) < 1e-12 and vi["utilisation"] < 1
    assert "issue_bound_gates_per_s" not in vi  # no issue model in this synthetic counter record
    pmc_im = dict(pmc, issue_model={"issue_bound_gates_per_s": 240000.0, "ns_per_fp64_slot": 2.21, "ns_per_other_vector_slot": 1.875, "source": "s"})
    vim = bench.roofline(p, st, 8192 / 40.8e-3, pmc_im)["valu_issue"]
    assert abs(vim["frac_of_issue_bound"] - rate / 240000.0) < 1e-12 and vim["issue_model"]["ns_per_fp64_slot"] == 2.21
    assert vi["geometry_differs_from_pmc_run"] is False  # 8 192 gates, slices of 16 steps (15.75 on average: the last one is short)
    st2 = types.SimpleNamespace(**dict(vars(st), bootstraps=4096))
    assert bench.roofline(p, st2, 1e5, pmc)["valu_issue"]["geometry_differs_from_pmc_run"] is True
    assert r["rocprof_avg_launch_ms"] == 0.82 and r["traffic"] == 1264.0 * 8192 * 15.75
    assert r["hbm_model"]["algorithmic_bytes_per_gate"] == 46480788 and r["hbm_model"]["reuse_factor"] > 10
    # a leg whose launch size selects another kernel than the one the counters were collected on: no counter figures attached
    rk = bench.roofline(p, st2, 1e5, pmc, "k_blind_rotate_w2r<3,7>")
    assert rk["kernel"] == "k_blind_rotate_w2r<3,7>" and "valu_issue" not in rk and rk["traffic"] is None and "pmc_note" in rk
    assert bench.roofline(p, st, 8192 / 40.8e-3, pmc, pmc["kernel"].split("<")[0] + "<3,7>")["valu_issue"]
    # the exact leg's record prices the two-limb product's flops on the same launch times
    rx = bench.roofline(p, st, 8192 / 40.8e-3, None, "k_blind_rotate_x1<3,7>", limbs=2)
    assert rx["algorithmic_flops_per_gate"] == 630 * 328704 and abs(rx["frac"] / r["frac"] - 328704 / 233472) < 1e-12
    assert "10 transforms" in rx["note"] and "24 M complex MACs" in rx["note"]
    # without committed counters the record still carries the algorithmic fraction
    r0 = bench.roofline(p, st, 8192 / 40.8e-3, None)
    assert r0["frac"] == r["frac"] and "valu_issue" not in r0 and r0["traffic"] is None


def test_self_launch_builds_a_torchrun_command_and_relays(bench, monkeypatch, capsys):
    """`python bench.py --gpus N` outside torch.distributed.run starts its own ranks as a CHILD process (never exec)."""
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(['{"n_gpus": 3}\n'])

        def wait(self):
            return 7

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    rc = bench.self_launch(types.SimpleNamespace(gpus=3, backend="gloo"))
    cmd = seen["cmd"]
    assert rc == 7 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "3"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "3", "--steps", "2"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert capsys.readouterr().out == '{"n_gpus": 3}\n'
    # more RCCL ranks than GPUs on the node: refused before anything is started.  The launcher counts GPUs from the kernel
    # driver's topology files, never through HIP / torch (it must not hold a GPU context while its ranks run) ...
    seen.clear()
    monkeypatch.setattr(bench, "count_gpus_without_runtime", lambda: 2)
    assert bench.self_launch(types.SimpleNamespace(gpus=3, backend="nccl")) == 2 and not seen
    # ... and where those files do not say (no KFD sysfs, as in this container) it starts the ranks and lets them report
    monkeypatch.undo()
    assert bench.count_gpus_without_runtime() in (None, 0) or bench.count_gpus_without_runtime() > 0
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    monkeypatch.setattr(bench, "count_gpus_without_runtime", lambda: None)
    assert bench.self_launch(types.SimpleNamespace(gpus=3, backend="nccl")) == 7 and seen["cmd"][1:3] == ["-m", "torch.distributed.run"]


def test_launcher_counts_gpus_from_the_kfd_topology(bench, tmp_path, monkeypatch):
    """The launcher's GPU count comes from sysfs (nodes with SIMDs are GPUs, the CPU node has none), narrowed by the
    visible-devices variables; no HIP call."""
    for node, simds in (("0", 0), ("1", 1024), ("2", 1024), ("3", 1024)):
        d = tmp_path / node
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simds == 0 else 0, simds))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.count_gpus_without_runtime(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus_without_runtime(str(tmp_path)) == 2
    assert bench.count_gpus_without_runtime(str(tmp_path / "missing")) is None
