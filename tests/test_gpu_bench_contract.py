"""bench.py's one-line JSON contract (metric/value/unit/n_gpus/steps/warmup/ms_per_step/higher_is_better/scaling/vs_baseline/
dtype/data/config + roofline + cpu_baseline, and the other BASELINE configs under "configs"), on small batches so that it
runs in seconds; plus the multi-rank launch (torch.distributed.run, two ranks sharing the one GPU of the box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline")


def run_bench(argv, launcher=(), timeout=1500):
    """runs bench.py; checks what the DRIVER sees -- the last line of stdout is one compact JSON object of at most 6 KB carrying the
    contract keys (round 5's 22 KB line was not parsed) -- and returns the FULL record bench.py wrote next to it, with the compact
    line under "_compact" """
    import tempfile
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as td:
        full_path = os.path.join(td, "full.json")
        p = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *argv, "--full-out", full_path], capture_output=True, text=True,
                           timeout=timeout, cwd=ROOT, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        lines = [ln for ln in out_lines if ln.startswith("{")]
        assert len(lines) == 1, "exactly one JSON line on stdout"
        assert out_lines[-1] == lines[0], "the JSON line is the LAST line of stdout"
        assert len(lines[0]) <= 6144, f"compact line is {len(lines[0])} bytes"
        c = json.loads(lines[0])
        for k in COMPACT_KEYS:
            assert k in c, k
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic", "hbm_effective"):
            assert k in c["roofline"], k
        assert "workload" in c["config"] and "batch_per_gpu" in c["config"]
        with open(full_path) as fh:
            full = json.load(fh)
    assert abs(full["value"] - c["value"]) <= 1e-5 * full["value"] and full["n_gpus"] == c["n_gpus"] and full["scaling"] == c["scaling"]
    full["_compact"] = c
    return full


def check_roofline(r, cfg):
    """the record names the roof that binds the launch (instruction issue; C4: the memory system) and quotes counter-derived
    numbers only from a committed pass stamped with the loaded library's build; the section-8(d) effective rate sits apart"""
    for k in ("bound", "bound_is", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic", "hbm_effective", "floor_frac", "library"):
        assert k in r, k
    assert r["bound"] == ("hbm" if cfg == "C4" else "issue")
    assert r["peak"] == (8000.0 if cfg == "C4" else 1024 * 2.4)
    e = r["hbm_effective"]
    assert e["peak"] == 8000.0 and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-12
    assert abs(e["achieved"] - e["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * e["achieved"]
    assert 0 < r["floor_frac"] < e["frac"] * 1.5 + 1
    assert len(r["library"]["csrc_sha16"]) == 16 and r["library"]["version"].startswith("daqp_amd")
    if r.get("stale"):      # (these tests run at batch sizes no counter pass was taken at)
        assert r["traffic"] is None and r["frac"] is None and r["achieved"] is None and "issue" not in r and r["stale_why"]
    else:
        assert r["traffic"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        if cfg == "C4":     # the memory system is the roof: measured HBM-side bytes of the launch over its time
            assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"] and 0 < r["frac"] <= 1.0
            assert abs(r["issue"]["frac"] - r["issue"]["attainable_ms"] / r["avg_launch_ms"]) < 1e-9
        else:
            assert abs(r["frac"] - r["issue"]["attainable_ms"] / r["avg_launch_ms"]) < 1e-9 and 0 < r["frac"] <= 1.0


def test_bench_json_line(gpu_lib):
    d = run_bench(["--steps", "2", "--warmup", "1", "--batch", "4096", "--cpu-sample", "512", "--side-configs", "C3,C5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in d, k
    assert d["unit"] == "QPs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["metric"].endswith("n=50 m=150") and d["config"]["workload"].startswith("C2: 4096 ")
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "traffic" in d["roofline"]
    assert d["checks"]["all_optimal"] and d["checks"]["rechecked_in_exact_arithmetic"] == 0      # the default-mode kernels' own verdicts
    check_roofline(d["roofline"], "C2")
    assert d["roofline"].get("stale") is True      # batch 4096: no committed counter pass at that size
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "cpu", "wall_s", "passes"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    assert c["kind"] == "port" or c["wall_s"] >= 1.0, "a timed CPU leg lasts at least a second"
    assert d["parity_vs_cpu"]["identical_active_set"] == 1.0 and d["parity_vs_cpu"]["max_abs_dx"] < 1e-9
    assert set(d["configs"]) == {"C3", "C5"}
    for name, s in d["configs"].items():
        for k in ("metric", "value", "unit", "ms_per_step", "workload", "roofline", "cpu_baseline", "parity_vs_cpu", "checks"):
            assert k in s, (name, k)
        assert s["workload"].startswith(name + ": ") and s["value"] > 0 and s["checks"]["all_optimal"] and s["checks"]["rechecked_in_exact_arithmetic"] == 0
        check_roofline(s["roofline"], name)
        assert s["cpu_baseline"]["kind"] == "port" or s["cpu_baseline"]["wall_s"] >= 1.0
    # "H2D/D2H reported separately" -- per configuration, never part of `value`
    for name, s in [("C2", d)] + list(d["configs"].items()):
        t = s["transfers"]
        for k in ("h2d_ms", "d2h_ms", "h2d_GBps", "d2h_GBps", "h2d_bytes_per_qp", "d2h_bytes_per_qp", "sample_qps", "value_with_transfers"):
            assert k in t and t[k] > 0, (name, k, t)
        assert t["value_with_transfers"] < s["value"]
    assert d["transfers"]["h2d_bytes_per_qp"] == 8 * (50 * 50 + 50 + 150 * 50 + 300) and d["configs"]["C5"]["transfers"]["h2d_bytes_per_qp"] == 8 * 50
    assert "n=12 m=48" in d["configs"]["C3"]["metric"] and d["configs"]["C3"]["parity_vs_cpu"]["identical_iter"] == 1.0
    # what the driver parses: cpu_baseline and the parity sample of the headline, one short summary per side configuration, no prose
    c = d["_compact"]
    for k in ("value", "unit", "cores", "kind", "cpu", "wall_s", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["parity_vs_cpu"]["identical_active_set"] == 1.0 and set(c["configs"]) == {"C3", "C5"}
    for name, s in c["configs"].items():
        assert s["value"] > 0 and s["ms_per_step"] > 0 and "frac" in s["roofline"] and s["cpu_baseline"]["value"] > 0, name
        assert abs(s["value"] - d["configs"][name]["value"]) <= 1e-5 * s["value"]
    assert not any(k in c for k in ("transfers", "batch_sweep", "exact", "arith")) and "bound_is" not in c["roofline"]
    assert d["configs"]["C5"]["unit"] == "warm solves/s" and d["configs"]["C5"]["parity_vs_cpu"]["identical_iter_last_step"] == 1.0
    assert d["configs"]["C5"]["parity_vs_cpu"]["max_abs_dx_last_step"] < 1e-9


def test_bench_headline_carries_the_batch_sweep(gpu_lib):
    """the headline configuration at its own batch size also reports where the device saturates: C2 at 1 000 / 10 000 / 100 000 problems
    (`batch_sweep`), and -- when a counter pass of this build is committed -- the setup launch's own roofline record (`roofline.setup`)"""
    d = run_bench(["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--side-configs", "none", "--no-exact"])
    sw = d["batch_sweep"]
    assert [e["batch"] for e in sw] == [1000, 10000, 100000] and all(e["value"] > 0 and e["solve_ms"] > 0 for e in sw)
    assert sw[2]["value"] == d["value"] and sw[0]["value"] < sw[2]["value"]
    r = d["roofline"]
    if not r.get("stale"):
        st = r["setup"]
        assert st["bound"] == "hbm" and st["peak"] == 8000.0 and abs(st["frac"] - st["achieved"] / st["peak"]) < 1e-12
        assert abs(st["achieved"] - st["traffic"] / (st["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * st["achieved"]
        assert st["traffic"] >= 0.9 * st["algorithmic_bytes_per_launch"] and "k_setup" in st["kernel"]


def test_bench_multi_entry_one_process(gpu_lib):
    """`bench.py --multi-entry --gpus G`: the single-process path of the C ABI (daqp_batch_*_multi_shards, a host thread and a stream per
    shard) in the same line format as the rank-per-GPU path; here with the one device of the box listed twice"""
    d = run_bench(["--multi-entry", "--gpus", "2", "--single-device", "--steps", "2", "--warmup", "1", "--batch", "2048"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["entry"] == "multi" and d["config"]["batch_per_gpu"] == 2048
    assert d["checks"]["all_optimal"] and d["checks"]["max_abs_x_minus_analytic_optimum"] < 1e-9
    assert abs(d["value"] - 2 * 2048 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    s = run_bench(["--multi-entry", "--gpus", "3", "--single-device", "--steps", "2", "--warmup", "1", "--config", "C3", "--batch", "4096"])
    assert s["n_gpus"] == 3 and "n=12 m=48" in s["metric"] and s["checks"]["all_optimal"]


def test_bench_two_ranks_one_device_weak_and_strong(gpu_lib):
    """the driver's multi-GPU launch shape on the one GPU of this box: two ranks (both on cuda:0, gloo rendezvous) --
    barrier, MAX over ranks and the strong-scaling split QP k -> rank k mod 2 of ONE batch"""
    port = 29600 + os.getpid() % 300
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    common = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--single-device", "--backend", "gloo", "--cpu-sample", "0"]
    d = run_bench(common + ["--batch", "2048"], launcher)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["batch_per_gpu"] == 2048
    assert "configs" not in d and d["checks"]["all_optimal"]
    s = run_bench(common + ["--config", "C3", "--strong", "--batch", "10001"], launcher)
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["config"]["batch_per_gpu"] == 5001   # rank 0 of 10001 interleaved
    assert "n=12 m=48" in s["metric"] and abs(s["value"] - 10001 * 2 / (s["ms_per_step"] * 2e-3)) < 1e-6 * s["value"]


def test_bench_rccl_process_group_on_the_device(gpu_lib):
    """the driver's launch shape with the REAL collective backend: torch.distributed.run, backend nccl (= RCCL on ROCm),
    init_process_group(device_id=cuda:0), barrier and MAX all-reduce of the elapsed time on the device -- one rank, because
    the box has one GPU (two RCCL ranks cannot share a device); the code path is the one every world size takes"""
    port = 29900 + os.getpid() % 90
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    d = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--backend", "nccl", "--cpu-sample", "0", "--batch", "2048",
                   "--side-configs", "none"], launcher)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["checks"]["all_optimal"]
    assert "RCCL" in d["config"]["parallelism"], d["config"]["parallelism"]
    s = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--backend", "nccl", "--cpu-sample", "0", "--config", "C3", "--strong",
                   "--batch", "4099", "--side-configs", "none"], launcher)
    assert s["scaling"] == "strong" and s["config"]["batch_per_gpu"] == 4099 and "RCCL" in s["config"]["parallelism"]


def test_bench_gpus_flag_starts_the_ranks_itself(gpu_lib):
    """plain `python bench.py --gpus 2` -- no launcher, what a driver that only knows the flag runs -- must BE two ranks: bench.py
    re-executes itself under torch.distributed.run (rank r on device r; here both on the one GPU of the box over gloo), weak and
    strong; a launcher whose world size contradicts --gpus is refused loudly"""
    common = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--single-device", "--backend", "gloo", "--cpu-sample", "0"]
    d = run_bench(common + ["--batch", "2048"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["batch_per_gpu"] == 2048 and d["checks"]["all_optimal"]
    assert "2 rank(s)" in d["config"]["parallelism"] and "gloo" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 2048 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    s = run_bench(common + ["--config", "C3", "--strong", "--batch", "10001"])
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["config"]["batch_per_gpu"] == 5001
    assert abs(s["value"] - 10001 * 2 / (s["ms_per_step"] * 2e-3)) < 1e-6 * s["value"]
    one = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2048", "--cpu-sample", "0", "--side-configs", "none"])
    assert one["n_gpus"] == 1 and "RCCL" not in one["config"]["parallelism"]          # --gpus 1: this process, no launcher, as before
    port = 29700 + os.getpid() % 200
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0 and "--gpus 2 but the launcher started WORLD_SIZE=1" in (p.stderr + p.stdout)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0 and "HIP device(s) are visible" in (p.stderr + p.stdout)       # one rank per GPU: two need two devices
