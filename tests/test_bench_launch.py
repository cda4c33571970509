"""bench.py --gpus N: rank fan-out (reference: training_go.py:317-347 starts one actor process per slot).
CPU tier: --launch-check rendezvouses the ranks over gloo and reports what joined; the measurement itself needs a GPU."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_n_self_launches_n_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size_env"] == 2 and line["backend"] == "gloo"


def test_gpus_n_without_n_devices_fails_loudly():
    """On a box with fewer than N devices `--gpus N` must not print a number (here: 0 devices)."""
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout) and "{" not in r.stdout


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_self_launch_command_shape():
    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse_args(["--gpus", "4", "--steps", "7"])
    cmd = bench.self_launch_cmd(a, ["--gpus", "4", "--steps", "7"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_broadcast_weights_two_ranks_gloo(tmp_path):
    script = os.path.join(ROOT, "tests", "bcast_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(tmp_path)], env=env) for r in range(2)]
    assert all(p.wait(timeout=300) == 0 for p in procs)
    w0, w1 = (np.load(os.path.join(str(tmp_path), f"w{r}.npz")) for r in range(2))
    assert set(w0.files) == set(w1.files) and len(w0.files) > 20
    for k in w0.files:
        if np.issubdtype(w0[k].dtype, np.floating):
            assert np.array_equal(w0[k], w1[k]), k


def test_cpu_baseline_calibration_fixture(golden_dir):
    """SURVEY 8d last row: the port-vs-reference ratio is a committed fixture (tools/calibrate_baseline.py) that bench.py reports."""
    doc = json.load(open(os.path.join(golden_dir, "cpu_baseline_calibration.json")))
    r = doc["ratios"]["go9_p8_s200_10x128"]
    assert 0.8 <= r["ratio"] <= 1.4 and r["reference_moves_per_s"] > 0 and sum(x["moves"] for x in r["runs"]) >= 100
    assert abs(r["ratio"] - r["port_moves_per_s"] / r["reference_moves_per_s"]) < 1e-3


def test_bench_measurement_flow_two_ranks_gloo(tmp_path):
    """The N-rank control flow of bench.py (what the driver runs on 8 GPUs) on 2 gloo ranks with host-twin actors: both ranks leave
    the pre-roll at the same round although one needs twice as many rounds per move, every collective pairs up (no hang), rank 0
    gathers samples inside the timed region, totals are sums over ranks and the time is the maximum."""
    script = os.path.join(ROOT, "tests", "bench_flow_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(tmp_path)], env=env) for r in range(2)]
    assert all(p.wait(timeout=600) == 0 for p in procs)
    f0, f1 = (json.load(open(os.path.join(str(tmp_path), f"flow{r}.json"))) for r in range(2))
    assert f0["preroll"] == f1["preroll"] >= 10
    assert f0["total_moves"] == f1["total_moves"] == f0["local_moves"] + f1["local_moves"] > 0
    assert f0["elapsed_max"] == f1["elapsed_max"] == max(f0["elapsed"], f1["elapsed"])
    assert f0["gathered"] > 0 and f1["gathered"] == 0  # samples arrive on rank 0 only
    # the per-rank report every rank computes from an all_gather: identical on both ranks, one entry per rank, own-clock moves/s
    assert f0["per_rank"] == f1["per_rank"] and len(f0["per_rank"]["moves_per_s"]) == 2
    for r, f in enumerate((f0, f1)):
        assert abs(f0["per_rank"]["moves_per_s"][r] - f["local_moves"] / f["elapsed"]) <= 0.06
    assert f0["per_rank"]["harvest_gather_calls"] == 240 // 7 and min(f0["per_rank"]["harvest_gather_ms_mean"]) > 0


def test_bench_measurement_flow_eight_ranks_gloo(tmp_path):
    """The same control flow at the driver's world size (8 gloo ranks, host-twin actors with four different budgets): one common
    pre-roll exit, no hang in any collective, samples on rank 0 only, totals = sums over the 8 ranks, time = the maximum, the per-rank
    report identical on every rank with 8 entries -- what `value(8) ~ 8 value(1) (1 - harvest_gather_share)` is computed from."""
    W, steps = 8, 120
    script = os.path.join(ROOT, "tests", "bench_flow_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29567", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(W), str(tmp_path), str(steps)], env=env) for r in range(W)]
    assert all(p.wait(timeout=900) == 0 for p in procs)
    f = [json.load(open(os.path.join(str(tmp_path), f"flow{r}.json"))) for r in range(W)]
    assert len({x["preroll"] for x in f}) == 1 and f[0]["preroll"] >= 10
    assert all(x["total_moves"] == sum(y["local_moves"] for y in f) for x in f) and f[0]["total_moves"] > 0
    assert all(x["elapsed_max"] == max(y["elapsed"] for y in f) for x in f)
    assert f[0]["gathered"] > 0 and all(x["gathered"] == 0 for x in f[1:])
    pr = f[0]["per_rank"]
    assert all(x["per_rank"] == pr for x in f) and len(pr["moves_per_s"]) == W == len(pr["harvest_gather_ms_mean"])
    for r, x in enumerate(f):
        assert abs(pr["moves_per_s"][r] - x["local_moves"] / x["elapsed"]) <= 0.06
    assert pr["min_moves_per_s"] == min(pr["moves_per_s"]) and pr["max_moves_per_s"] == max(pr["moves_per_s"])
    assert pr["harvest_gather_calls"] == steps // 7 and 0 < pr["harvest_gather_share_of_time"] < 1
    # whole-job value = sum of moves / max time: at most the sum of the ranks' own rates, and what the formula in the bench line predicts
    value = f[0]["total_moves"] / f[0]["elapsed_max"]
    assert value <= sum(pr["moves_per_s"]) + 1e-6 and value >= W * pr["min_moves_per_s"] * 0.5


def test_bench_exports_dmabuf_ipc_mode_for_itself_and_its_ranks(monkeypatch):
    """Multi-process GPU work on this pool needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; RCCL otherwise fails in hipIpcGetMemHandle):
    bench.py puts it in its own environment at import (before the HSA runtime starts) -- never overriding a caller's choice -- and hands
    it to the ranks it self-launches (the nccl tests exported it, the self-launch did not)."""
    import importlib
    import subprocess
    import sys

    src = ("import os; os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); import bench; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY']);"
           "os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '1'; import importlib; importlib.reload(bench); print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])")
    out = subprocess.run([sys.executable, "-c", src], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["0", "1"]
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert 'subprocess.call(self_launch_cmd(args, argv), env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=' in text
