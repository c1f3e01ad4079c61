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


def run_bench(argv, launcher=(), timeout=1500):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def check_roofline(r):
    for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "floor_frac"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert 0 < r["floor_frac"] < r["frac"] * 1.5 + 1


def test_bench_json_line(gpu_lib):
    d = run_bench(["--steps", "2", "--warmup", "1", "--batch", "4096", "--cpu-sample", "512", "--side-configs", "C3,C5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "configs"):
        assert k in d, k
    assert d["unit"] == "QPs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["metric"].endswith("n=50 m=150") and d["config"]["workload"].startswith("C2: 4096 ")
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "traffic" in d["roofline"]
    check_roofline(d["roofline"])
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    assert d["parity_vs_cpu"]["identical_active_set"] == 1.0 and d["parity_vs_cpu"]["max_abs_dx"] < 1e-9
    assert set(d["configs"]) == {"C3", "C5"}
    for name, s in d["configs"].items():
        for k in ("metric", "value", "unit", "ms_per_step", "workload", "roofline", "cpu_baseline", "parity_vs_cpu", "checks"):
            assert k in s, (name, k)
        assert s["workload"].startswith(name + ": ") and s["value"] > 0 and s["checks"]["all_optimal"]
        check_roofline(s["roofline"])
    assert "n=12 m=48" in d["configs"]["C3"]["metric"] and d["configs"]["C3"]["parity_vs_cpu"]["identical_iter"] == 1.0
    assert d["configs"]["C5"]["unit"] == "warm solves/s" and d["configs"]["C5"]["parity_vs_cpu"]["identical_iter_last_step"] == 1.0
    assert d["configs"]["C5"]["parity_vs_cpu"]["max_abs_dx_last_step"] < 1e-9


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
