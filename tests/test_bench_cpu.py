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


def _synthetic_full_record(bench, ia):
    """A full record with every leg present, built from synthetic stats the way main() builds it."""
    p = ia.default_params()
    st = types.SimpleNamespace(blind_rotate_ms=40.0, blind_rotate_launches=40, keyswitch_ms=0.7, keyswitch_launches=1,
                               bootstraps=8192, chunks=1, total_ms=40.8)
    pmc = {"kernel": "k_blind_rotate_w1b<3,7,guard on 1 coefficient in 4>", "valu_insts_per_gate_step": 3158.0, "fp64_insts_per_gate_step": 2493,
           "hbm_bytes_per_gate_step": 1264.0, "gates_per_launch": 8192, "cmux_steps_per_launch": 16, "shader_cycles_per_gate_step": 17.08,
           "source": "profiles/r4_pmc_summary.txt", "rocprof_avg_launch_ms": 0.811508,
           "rocprof_stats": "profiles/r4_c_add16_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs none --exact-leg off)",
           "issue_model": {"issue_bound_gates_per_s": 240000.0, "ns_per_fp64_slot": 2.21, "ns_per_other_vector_slot": 1.875, "source": "s"}}
    roof = bench.roofline(p, st, 8192 / 40.8e-3, pmc, "k_blind_rotate_w1b<3,7>")
    leg = {"workload": bench.WORKLOADS["mul32"][3], "circuit": "mul32", "batch_per_gpu": 1024, "bootstraps_per_expr": 11264, "levels": 255,
           "passes": 2, "ms_per_pass": 55004.915184981655, "gate_ops_per_s": 209696.46005652408, "expressions_per_s": 18.616518115813573,
           "per_pass_gate_ops_per_s": [209596.46005652408, 209796.46005652408], "spread": 0.00095, "per_rank_gate_ops_per_s": [209696.46005652408] * 8,
           "checked": "x" * 80, "warmup": "y" * 300, "roofline": roof, "mul32_per_s": 18.616518115813573, "note": "z" * 200}
    full = {"metric": "bootstrapped gate ops/sec; encrypted 32-bit MUL/sec", "value": 209309.12345678, "unit": "gate ops/s", "n_gpus": 8,
            "steps": 20, "warmup": 5, "ms_per_step": 1565.4321, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": bench.WORKLOADS["add16"][3] + " + " + bench.WORKLOADS["mul32"][3], "value_workload": bench.WORKLOADS["add16"][3],
                       "metric_leg_workload": bench.WORKLOADS["mul32"][3], "circuit": "add16", "batch_per_gpu": 4096, "bootstraps_per_expr": 80,
                       "levels": 48, "params": "n=630 N=1024 k=1 l=3 Bgbit=7 ks_t=8 ks_basebit=2", "arithmetic": "a" * 400,
                       "parallelism": "batch-sharded x8", "kernel": "w1x64-radix8-onelimb", "key_broadcast_s": 0.1234, "rccl_ranks": 8,
                       "collective_backend": "nccl", "per_rank_gate_ops_per_s": [26163.640432098] * 8},
            "roofline": roof,
            "fft_guard": {"max_rounding_deviation": 0.0123, "reruns_on_two_limb_kernel": 0, "limit": 0.0625, "wrong_bit_at": 0.5,
                          "audit": {"audits": 30, "gates_compared": 1920, "mismatches": 0, "note": "n" * 300}},
            "mul32": leg, "muladd64": dict(leg, sub_batch_of=128), "mul128": dict(leg, sub_batch_of=1024, full_share_estimate_s=600.0),
            "exact": {"what": "w" * 300, "workload": "c", "batch_per_gpu": 4096, "passes": 1, "ms_per_pass": 2230.0, "gate_ops_per_s": 146893.0,
                      "per_rank_gate_ops_per_s": [18361.0] * 8, "kernel": "x1x64-radix8-twolimb", "bit_identical_to_primary_leg": True,
                      "checked": "c" * 100, "vs_primary": 0.7018, "roofline": bench.roofline(p, st, 1e5, None, "k_blind_rotate_x1<3,7>", limbs=2)},
            "metric_leg": {"workload": leg["workload"], "batch_per_gpu": 1024, "gate_ops_per_s": leg["gate_ops_per_s"], "mul32_per_s": leg["mul32_per_s"],
                           "passes": 2, "per_pass_gate_ops_per_s": leg["per_pass_gate_ops_per_s"], "spread": 0.00095,
                           "ms_per_pass": leg["ms_per_pass"], "roofline_frac": roof["frac"], "details": "mul32"},
            "mul32_per_s": leg["mul32_per_s"],
            "skipped_legs": [{"leg": "mul128", "reason": "r" * 80}],
            "cpu_baseline": {"value": 68.03448771528342, "unit": "bootstrapped gate ops/s", "cores": 1, "kind": "port",
                             "sample": "613 AND/XOR gates (n=630,N=1024) in 9.0 s with the oracle's FP64-FFT back-end " + "s" * 200,
                             "all_cores": {"value": 886.2, "cores": 16, "sample": "t" * 200},
                             "real_libtfhe": "unavailable on this host (no libtfhe* in /etc/ld.so.cache = `ldconfig -p | grep tfhe` empty)",
                             "note": "n" * 400}}
    return full


def test_line_is_compact(bench, ia):
    """The ONE line stays below 6 000 bytes with every leg present at N = 8 (round 4's 21 KB line was not parsed by the driver),
    parses back, and carries the contract's keys plus `roofline` and `cpu_baseline`; everything wordy is in bench_details.json."""
    import json
    full = _synthetic_full_record(bench, ia)
    assert len(json.dumps(full)) > 12000  # the full record is what no longer fits
    line = bench.compact_line(full, "bench_details.json")
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 6000
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "metric_leg", "exact", "mul32", "muladd64", "mul128", "details"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5 and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-2
    r = d["roofline"]
    for k in ("bound", "unit", "achieved", "peak", "frac", "traffic", "kernel", "avg_launch_ms", "gates_per_launch", "cmux_steps_per_launch",
              "rocprof_avg_launch_ms", "rocprof_stats", "hbm_frac", "measured_hbm_GBps"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and r["rocprof_stats"] == "profiles/r4_c_add16_kernel_stats.csv"
    assert "note" not in r and "valu_issue" not in r and "hbm_model" not in r and "per_kernel" not in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["all_cores"] == {"value": 886.2, "cores": 16} and "note" not in c
    assert c["sample"].startswith("613 AND/XOR gates") and len(c["sample"]) < 80
    m = d["metric_leg"]
    assert m["passes"] == 2 and len(m["per_pass_gate_ops_per_s"]) == 2 and m["workload"] == bench.WORKLOADS["mul32"][3]
    assert d["config"]["value_workload"] == bench.WORKLOADS["add16"][3] and len(d["config"]["per_rank_gate_ops_per_s"]) == 8
    assert d["exact"]["bit_identical_to_primary_leg"] is True and d["mul128"]["sub_batch_of"] == 1024 and d["skipped_legs"] == ["mul128"]
    assert not any(isinstance(v, str) and len(v) > 200 for v in d["config"].values())
    # a record that would still be too long sheds its optional objects rather than print an unreadable line
    fat = dict(full, config=dict(full["config"], per_rank_gate_ops_per_s=[1.5] * 1200))
    shed = json.loads(bench.compact_line(fat, "bench_details.json"))
    assert "mul128" not in shed and "roofline" in shed and "cpu_baseline" in shed


def test_details_file_holds_the_full_record(bench, ia, tmp_path, monkeypatch):
    import json
    full = _synthetic_full_record(bench, ia)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.write_details(full) == "bench_details.json"
    back = json.load(open(tmp_path / "bench_details.json"))
    assert back["roofline"]["note"] == full["roofline"]["note"] and back["mul32"]["roofline"]["hbm_model"]["note"]
