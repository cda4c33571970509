"""RCCL on hardware with the one GPU a test box has: a single-rank "nccl" process group (VERDICT r2 "Next" #3a).  CPU tier twins of
the same code paths on 2 gloo ranks: tests/test_actor_host.py::test_sample_gather_two_ranks_gloo, tests/test_bench_launch.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env():
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.gpu
def test_gpu_gather_and_broadcast_through_single_rank_nccl_group():
    """gather_samples / broadcast_weights on device tensors through an initialised nccl group: the collectives run (counted), the
    packed wire format round-trips bit-exactly on the device."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py")], env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert rep["ok"] and rep["backend"] == "nccl" and rep["collectives"]["gather"] >= 1 and rep["collectives"]["broadcast"] == 1


@pytest.mark.gpu
def test_gpu_bench_end_to_end_through_single_rank_nccl_group():
    """`bench.py --gpus 1` started the way the driver starts N ranks (torch.distributed.run, here with one): RCCL init, barriers,
    the pre-roll's all_reduce, harvest + packed gather inside the timed region and the per-rank report all execute on the device."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "2", "--games", "512", "--sims", "32",
           "--blocks", "2", "--preroll-rounds", "20", "--harvest-every", "6", "--no-fp32", "--no-cpu-baseline", "--no-fresh-tree"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["process_group"] == "nccl" and line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and line["samples_gathered"] > 0
    assert len(line["per_rank"]["moves_per_s"]) == 1 and line["per_rank"]["harvest_gather_calls"] == 2
