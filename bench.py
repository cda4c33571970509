"""bench.py -- self-play moves/sec of the batched engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: WORLD_SIZE must then equal N) or, when WORLD_SIZE is not set, bench.py re-executes itself under
torch.distributed.run with N ranks -- the fan-out of training_go.py:317-347 (one actor per process), one process per GPU here.
It exits non-zero when fewer than N devices are visible: a 1-GPU number is never reported as an N-GPU one.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 9x9 Go, G = 4096
concurrent games per GPU, 200 sims/move with the reference budget semantics (root.N >= sims + P,
inherited visits count: mcts_v2.py:568), num_parallel P = 8, 10-block x 128-filter AlphaZeroNet
(random Kaiming init, torch.manual_seed(1)), Dirichlet root noise, sub-tree reuse, resign disabled.
The evaluator runs at the REFERENCE'S precision class (fp32: pipeline.py:91-123, no autocast anywhere in the reference) on the
hand-written split-precision kernels (az_conv_sp.h: every value a hi + lo f16 pair, a product = three f16 MFMAs into fp32
accumulators; as close to fp64 as the library's fp32 convolution, tests/test_split_tower.py).  `--net-dtype bf16 / fp16` select the
lower-precision evaluators; the default run reports the bf16 one as a labelled companion, never as `value`.
One "step" = one engine round over all games: expand/backup of the previous G*P leaf batch, end-of-move
work, selection of the next P leaves per game, observation planes, and the network forward on G*P rows.
All inputs live in HBM; nothing crosses PCIe inside the timed region except the harvests of finished games.

Before --warmup / --steps apply, an untimed, argument-independent PRE-ROLL brings the engine to its steady state: slots start
from staggered random openings, then rounds run until every slot has committed >= 2 searched moves and >= 300 rounds have passed
(a fresh tree needs 26 rounds to its first move; afterwards inherited sub-trees of different sizes de-phase the slots), so that
move completions are spread evenly over rounds and a 20-round window measures the same moves/s as a 300-round one.  The timed
region always contains at least two harvests + gathers (every min(--harvest-every, --steps / 2) rounds).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# Multi-process GPU work on this pool needs dmabuf IPC (the host driver supports no legacy IPC handles: without this RCCL fails in
# hipIpcGetMemHandle).  It is exported in the build container and on the GPU boxes; bench.py sets it for itself and for the ranks it
# launches so that an N-GPU run does not depend on the caller's environment.  Must be in place before the HSA runtime initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}
MFMA_PEAK_CLOCK_GHZ = 2.4  # the clock the 2.5 PF/s dense f16 / bf16 peak is quoted at (256 CUs x 4 SIMDs x 1024 flop / cycle)
SPLIT_PRODUCTS = 3  # f16 MFMA products per fp32-class multiply-add of the split-precision kernels (w_hi x_hi, w_hi x_lo, w_lo x_hi)
DTYPE_LABEL = {"fp32": "fp32-class evaluator (f16 hi+lo pairs, 3 MFMA products, fp32 accumulate) / f32-f64 tree",
               "bf16": "bf16 evaluator / f32-f64 tree", "fp16": "fp16 evaluator / f32-f64 tree"}


def net_flops_per_eval(n, A, blocks, filters, fc, gomoku):
    """2*MAC of the convolutions and linear layers."""
    s = (n + 4) if gomoku else n
    f = 2 * 17 * filters * 9 * s * s + blocks * 2 * (2 * filters * filters * 9 * s * s)
    f += 2 * filters * 3 * s * s + 2 * (2 * s * s) * A + 2 * (s * s) * fc + 2 * fc
    return float(f)


def usable_host_cores():
    """Host threads this job may actually use: min(os.cpu_count(), cgroup CPU quota).  The GPU boxes expose 256
    threads but cap the container at 16 CPUs (cpu.max = "1600000 100000"); oversubscribing them only slows the
    baseline down (measured: 16 procs 54 moves/s, 64 procs 44, 256 procs 38)."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--game", default="go")
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--parallel", type=int, default=8)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--net-dtype", default="fp32", choices=["bf16", "fp16", "fp32"],
                    help="evaluator precision class: fp32 (default) = the reference's (pipeline.py:91-123 evaluates in fp32) on the hand-written "
                         "split-precision kernels (hi + lo f16 pairs, three MFMA products, fp32 accumulation); bf16 / fp16 = the lower-precision evaluators")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fused-block", action="store_true", help="A/B: two convolution launches per ResNetBlock even where a one-launch block kernel exists")
    ap.add_argument("--stagger", type=int, default=60, help="random opening plies per slot so game phases are mixed from the start")
    ap.add_argument("--preroll-rounds", type=int, default=300, help="minimum untimed rounds after the stagger (steady state, see module docstring)")
    ap.add_argument("--preroll-moves", type=int, default=2, help="every slot must have committed this many searched moves before timing")
    ap.add_argument("--harvest-every", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=60.0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="0 = all usable host cores (cgroup quota aware)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", "--no-fp32", dest="no_companions", action="store_true",
                    help="skip the companion measurements (lower-precision bf16 evaluator, library fp32 evaluator)")
    ap.add_argument("--no-fresh-tree", action="store_true",
                    help="skip the sub-tree-reuse-off companion measurement (fresh_tree_moves_per_s, SURVEY 8d's upper-work variant)")
    ap.add_argument("--no-miopen-find", action="store_true", help="disable torch.backends.cudnn.benchmark (MIOpen find) for the convs")
    ap.add_argument("--split-round", action="store_true", help="diagnostic: launch expand/backup and select as two kernels and time each")
    ap.add_argument("--launch-check", action="store_true",
                    help="only rendezvous the ranks (RCCL on GPUs, gloo without a device), print the world size that joined and exit")
    return ap.parse_args(argv)


def self_launch_cmd(args, argv):
    """The command bench.py re-executes itself with when --gpus N > 1 and no launcher started it (WORLD_SIZE unset): one rank per
    GPU under torch.distributed.run on this node (training_go.py:317-347 starts one actor process per slot the same way)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_check(world, rank, local_rank):
    """--launch-check: ranks rendezvous, count themselves with one all_reduce and rank 0 prints what joined."""
    import torch.distributed as dist

    if world > 1:
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            one = torch.ones(1, device="cuda")
        else:
            dist.init_process_group("gloo")
            one = torch.ones(1)
        dist.all_reduce(one)
        joined, backend = int(one.item()), dist.get_backend()
        dist.barrier()
        dist.destroy_process_group()
    else:
        joined, backend = 1, None
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": joined, "world_size_env": world, "backend": backend}), flush=True)


# ---- the measurement flow (module level: tests/bench_flow_worker.py drives exactly these functions under gloo, 2 ranks, with a
# host-twin actor, so the N-rank control flow the driver runs on 8 GPUs -- paired collectives, synchronised pre-roll exit, gather
# inside the timed region, max-over-ranks timing -- is exercised on CPU; bench.py itself only ever builds GPU actors) -------------
def pg_active():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def barrier(world, dev):
    if pg_active():  # every launcher-started run has a process group, a single-rank one included
        import torch.distributed as dist

        dist.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


HG_MS = []  # wall-clock of every harvest + gather of this rank, measured from an idle stream (the rounds queued before it have drained)


def harvest_and_gather(act):
    from alpha_zero_amd.core.gather import gather_samples

    if act.device.type == "cuda":
        torch.cuda.synchronize(act.device)  # the harvest would wait for the queued rounds anyway: keep their time out of this number
    t0 = time.perf_counter()
    st, pi, z, games = act.harvest_tensors()
    res = gather_samples(st, pi, z, games, dst=0)
    HG_MS.append((time.perf_counter() - t0) * 1e3)
    return 0 if res is None else int(res[0].shape[0])


def preroll(act, args, world, dev, min_rounds=None):
    """Untimed and independent of --steps / --warmup: rounds until every slot has committed >= preroll_moves searched moves
    (ply advanced, or a game finished) and >= preroll_rounds rounds have passed.  Returns the rounds it took."""
    e = act.engine
    st0, _ = e.status()
    ply0, done0 = st0[:, 1].copy(), st0[:, 5].copy()
    min_rounds = args.preroll_rounds if min_rounds is None else min_rounds
    rounds, cap = 0, max(min_rounds, 40 * (args.sims // max(1, args.parallel) + 2))
    while rounds < cap:
        act.run_round()
        rounds += 1
        if rounds % args.harvest_every == 0:
            harvest_and_gather(act)
        if rounds >= min_rounds and rounds % 10 == 0:
            st, _ = e.status()
            moved = np.where(st[:, 5] > done0, args.preroll_moves, st[:, 1] - ply0)
            ok = torch.tensor([1.0 if moved.min() >= args.preroll_moves else 0.0], device=dev)
            if pg_active():  # all ranks leave the pre-roll together (their harvest / gather calls must stay paired)
                import torch.distributed as dist

                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() > 0:
                break
    else:  # the cap was reached without the steady-state criterion: say so instead of silently timing an unsettled engine
        print(f"bench.py: WARNING pre-roll stopped at its cap of {cap} rounds before every slot had committed {args.preroll_moves} searched moves",
              file=sys.stderr, flush=True)
    return rounds


def timed(act, args, world, dev, warmup, steps):
    """--warmup untimed rounds, then exactly --steps rounds between barrier + synchronize on both sides."""
    h_every = max(1, min(args.harvest_every, steps // 2))  # the timed region always pays for >= 2 harvests + gathers (1 when steps == 1)
    for i in range(warmup):
        act.run_round()
        if (i + 1) % h_every == 0:
            harvest_and_gather(act)
    barrier(world, dev)
    harvest_and_gather(act)
    act.counters(reset=True)
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(steps)] if dev.type == "cuda" else [None] * steps
    gathered = 0
    barrier(world, dev)
    del HG_MS[:]
    t0 = time.perf_counter()
    for i in range(steps):
        act.run_round(evs[i])
        if (i + 1) % h_every == 0:
            gathered += harvest_and_gather(act)
    barrier(world, dev)
    return time.perf_counter() - t0, act.counters(), evs, gathered


def reduce_totals(cnt, elapsed, world, dev):
    """Whole-job totals: moves / sims / evaluations summed over ranks, elapsed = MAX over ranks."""
    tot = torch.tensor([float(cnt["moves"]), float(cnt["sims"]), float(cnt["leaves"])], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if pg_active():
        import torch.distributed as dist

        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_moves, total_sims, total_evals = (float(x) for x in tot.tolist())
    return float(tmax.item()), total_moves, total_sims, total_evals


def per_rank_report(cnt, elapsed, world, dev):
    """What every rank did inside its own timed region, so that an N-GPU line explains itself: moves/s per rank (own clock), and the
    mean wall-clock of one harvest + gather call (the only inter-rank exchange of the data path) per rank."""
    mine = torch.tensor([cnt["moves"] / max(elapsed, 1e-9), float(np.mean(HG_MS)) if HG_MS else 0.0, float(np.max(HG_MS)) if HG_MS else 0.0,
                         float(len(HG_MS)), float(np.sum(HG_MS)) / max(elapsed * 1e3, 1e-9)], dtype=torch.float64, device=dev)
    rows = [mine]
    if pg_active():
        import torch.distributed as dist

        rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
    t = torch.stack(rows).cpu().numpy()
    return {"moves_per_s": [round(float(v), 1) for v in t[:, 0]], "min_moves_per_s": round(float(t[:, 0].min()), 1),
            "max_moves_per_s": round(float(t[:, 0].max()), 1), "harvest_gather_ms_mean": [round(float(v), 3) for v in t[:, 1]],
            "harvest_gather_ms_max": round(float(t[:, 2].max()), 3), "harvest_gather_calls": int(t[0, 3]),
            "harvest_gather_share_of_time": round(float(t[:, 4].max()), 5)}  # of each rank's own timed region, the largest


def tower_replay(actor, args, dev, reps=5):
    """Times the dominant kernel: the forward's own tower launch sequence, one HIP event after every launch (recorded on the stream the
    kernels are launched on), on the REAL activations of the last forward -- the tower input that forward's stem produced is kept in a
    copy and every repetition restarts from it, so each timed launch sees exactly the data it sees inside the step (round 4 chained the
    repetitions on their own output: a 50-block-deep random network whose activations left f16's range and tripped the range record).
    Returns None when the tower is not on hand-written kernels.  The network's range record is left as it was found."""
    import ctypes

    inf, dll = actor.infer, actor.binding.dll
    n = args.board
    S_t = n + 2 * (inf.stem_pad - 1)  # tower planes (17x17 behind the Gomoku pad-3 stem)
    rows = actor.engine.rows
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    split = args.net_dtype == "fp32" and getattr(inf, "_split", None) is not None and inf.supports_split_features(n, dev)
    tiled = args.net_dtype != "fp32" and getattr(inf, "_tiled", None) is not None and actor.tiled_features
    if not (split or tiled):
        return None
    fused = inf.use_fused_block and ((tiled and (args.filters, S_t) in ((64, 17), (64, 9)) and args.net_dtype == "bf16")  # one launch per ResNetBlock
                                     or (split and (args.filters, S_t) in inf.SPLIT_FUSED_SHAPES))
    e = actor.engine
    # re-create the forward's tower input: run the stem of the engine-facing forward once more on the live features (slot 0's buffer `a`)
    if split:
        (a, m, o), _, _, _ = inf._split_buffers(rows, S_t, args.filters, dev, 0)  # slot 0 = the engine-facing forward's buffers
        rr = inf._range_ptr(dev)
        assert dll.azsp_stem_split_exact(e.features.data_ptr(), inf.stem_wsp.data_ptr(), inf.stem_b_sp.data_ptr(), a.data_ptr(), rows, n, args.filters,
                                         inf.stem_pad, 1, rr, st) == 0
        conv_fn = lambda x, i, r, y: dll.azsp_conv3x3_split(x.data_ptr(), inf.wsp[i].data_ptr(), inf.b_sp[i].data_ptr(),  # noqa: E731
                                                            None if r is None else r.data_ptr(), y.data_ptr(), rows, S_t, args.filters, 1, rr, st)
    else:
        a, m, o = inf._tiled
        stem = dll.azsp_stem_tiled_f16 if args.net_dtype == "fp16" else dll.azsp_stem_tiled
        assert stem(e.features.data_ptr(), inf.stem_wp.data_ptr(), inf.stem_b32.data_ptr(), a.data_ptr(), rows, n, args.filters, inf.stem_pad, 1, st) == 0
        fn = dll.azsp_conv3x3_tiled_f16 if args.net_dtype == "fp16" else dll.azsp_conv3x3_tiled
        conv_fn = lambda x, i, r, y: fn(x.data_ptr(), inf.wp[i].data_ptr(), inf.b32[i].data_ptr(), None if r is None else r.data_ptr(),  # noqa: E731
                                        y.data_ptr(), rows, S_t, args.filters, 1, st)
    x0 = a.clone()  # the tower input of a real forward
    a0, o0 = a, o
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2 * inf.n_blocks + 1)] for _ in range(reps)]
    torch.cuda.synchronize(dev)
    d = []
    for rep in range(reps):
        a, o = a0, o0
        a.copy_(x0)  # every repetition = the forward's tower on the forward's own input (untimed copy)
        k = 0
        ev[rep][0].record()
        for i in range(inf.n_blocks):  # the forward's own launch sequence
            if fused and split:
                assert dll.azsp_resblock_split(a.data_ptr(), inf.wsp[2 * i].data_ptr(), inf.b_sp[2 * i].data_ptr(), inf.wsp[2 * i + 1].data_ptr(),
                                               inf.b_sp[2 * i + 1].data_ptr(), o.data_ptr(), rows, S_t, args.filters, rr, st) == 0
                k += 1
                ev[rep][k].record()
                a, o = o, a
                continue
            if fused:
                assert dll.azsp_resblock_tiled(a.data_ptr(), inf.wp[2 * i].data_ptr(), inf.b32[2 * i].data_ptr(), inf.wp[2 * i + 1].data_ptr(),
                                               inf.b32[2 * i + 1].data_ptr(), o.data_ptr(), rows, S_t, args.filters, st) == 0
                k += 1
                ev[rep][k].record()
                a, o = o, a
                continue
            assert conv_fn(a, 2 * i, None, m) == 0
            k += 1
            ev[rep][k].record()
            assert conv_fn(m, 2 * i + 1, a, o) == 0
            k += 1
            ev[rep][k].record()
            a, o = o, a
        torch.cuda.synchronize(dev)
        d += [ev[rep][j].elapsed_time(ev[rep][j + 1]) for j in range(k)]
    del x0
    return {"launches": len(d), "avg_ms": float(np.mean(d)), "planes": S_t, "fused_block": fused, "split": split,
            "avg_ms_plain": None if fused else float(np.mean(d[0::2])), "avg_ms_residual": None if fused else float(np.mean(d[1::2])),
            "data": "the real activations of a forward of the timed run (every repetition restarts from the stem's output)"}


def evaluator_range(actor):
    """Range record of the actor's fp32-class evaluator since the record was last read (InferenceNet.range_rec, this network's own):
    events = kernel lanes that met |v| > 65504 (scaled units) and clamped it.  None for evaluators without the record."""
    inf = actor.infer
    if actor.net_dtype != torch.float32 or not hasattr(inf, "range_rec") or not inf.range_rec.is_cuda:
        return None
    ev, mx = inf.split_range_status(reset=True)
    return {"events": ev + actor.range_events, "largest_abs": max(mx * 2.0 ** inf.act_shift, actor.range_max_abs), "act_shift": inf.act_shift,
            "calibrated_max_abs": round(inf.act_max_abs, 4), "rescales_during_run": actor.range_rescales, "games_in_a_clamp_window": actor.clamped_games,
            "fallback": inf.split_fallback_reason or None}


def mfma_ceilings(split):
    """What the matrix cores alone sustain with this kernel's instruction form and operand mix, from profiles/mfma_power_probe.json (the
    output of tools/probes/mfma_power_probe.hip on MI355X; nothing is hard-coded here): `issue_ceiling` = the form with ALL-ZERO operands,
    where the package power limit cannot bind -- the rate at which the instruction can be issued at all (v_mfma_f32_16x16x32 issues every
    ~20 cycles for 16 cycles of work: 0.79 of the 2.5 PF/s peak; v_mfma_f32_32x32x16: 0.985); `sustained` = the form on the kernel's own
    operand mix (real data: the smaller of the issue ceiling and the power limit).  `limit` says which of the two `sustained` is.
    (Rounds 4-5 published `sustained` of the split mix as "power_limited_mfma_only_tflops": for the 16x16x32 form it is the ISSUE ceiling.)"""
    pth = os.path.join(ROOT, "profiles", "mfma_power_probe.json")
    try:
        pj = json.load(open(pth))
        cases = {(c["mfma"], c["operands"].split(" (")[0]): c["tflops"] for c in pj["cases"]}
        if split:
            form, pipe_cycles = "v_mfma_f32_16x16x32_f16", 16.0
            issue = float(pj.get("issue_ceiling_16x16x32_tflops", cases[("v_mfma_f32_16x16x32_bf16", "zeros")]))
            sustained = float(pj["split_mix_mfma_only_tflops"])
            what = "split-precision operand mix, f16"
        else:
            form, pipe_cycles = "v_mfma_f32_32x32x16_bf16", 32.0
            issue = float(pj.get("issue_ceiling_32x32x16_tflops", cases[("v_mfma_f32_32x32x16_bf16", "zeros")]))
            sustained = float(cases[("v_mfma_f32_32x32x16_bf16", "A dense, B half zeros")])
            what = "bf16, A dense, B half zeros"
        out = {"instruction_form": form, "pipe_cycles_per_mfma_at_peak": pipe_cycles, "issue_ceiling_tflops": issue, "sustained_tflops": sustained,
               "limit": "issue" if sustained >= 0.97 * issue else "power", "source": f"from_profiles: profiles/mfma_power_probe.json ({what}; all-zero operands for the issue ceiling)"}
        if split and "split_mix_32x32x16_tflops" in pj:  # the other instruction form on the same operand mix
            out["other_form"] = {"instruction_form": "v_mfma_f32_32x32x16_f16", "sustained_tflops": float(pj["split_mix_32x32x16_tflops"]),
                                 "issue_ceiling_tflops": float(pj["split_mix_32x32x16_zero_operands_tflops"])}
        return out
    except Exception:
        return None


def kernel_pmc(cname):
    """A PMC summary from profiles/ with its freshness: `stale` is True when the summary records the digest of the kernel sources it was
    measured on and that digest differs from the sources of this tree (the kernel was edited after the counters were collected), None
    when the summary predates the digest (rounds 1-5: unknown)."""
    cprof = os.path.join(ROOT, "profiles", cname)
    if not os.path.exists(cprof):
        return None, None
    try:
        pj = json.load(open(cprof))
    except Exception:
        return None, None
    stale = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_digest import PMC_FILE_FAMILY, kernel_source_digest

        if pj.get("kernel_source_sha256") and cname in PMC_FILE_FAMILY:
            stale = pj["kernel_source_sha256"] != kernel_source_digest(PMC_FILE_FAMILY[cname])
    except Exception:
        stale = None
    return pj, stale


def tower_roofline(conv, args, step_ms):
    """`roofline` object of the step's dominant kernel from tower_replay's timings."""
    n = args.board
    rows = args.games * args.parallel
    S_t, fused, split = conv["planes"], conv["fused_block"], conv["split"]
    convs_per_launch = 2 if fused else 1
    # algorithmic work per launch = the dense 3x3 convolution (padding taps counted, the usual convention): 2 * rows * S^2 * C * C * 9 flop
    conv_flops = 2.0 * rows * S_t * S_t * args.filters * args.filters * 9 * convs_per_launch
    launches_per_step = args.blocks * (1 if fused else 2)
    # HBM traffic per launch is NOT measured in this run: it comes from a separate `rocprofv3 --pmc` pass (the guide's recipe: counters in
    # their own run) whose summary is committed under profiles/ -- labelled as such in `traffic_source`
    if split:
        cname = (("splitblock9_64_kernel_pmc.json" if fused else "split_kernel_pmc.json") if S_t == 9 else
                 ("splitblock17_kernel_pmc.json" if fused else "split17_kernel_pmc.json"))
    else:
        cname = "block64_kernel_pmc.json" if fused else {(9, 128): "conv_kernel_pmc.json", (19, 256): "conv19_kernel_pmc.json"}.get((S_t, args.filters), "conv64_kernel_pmc.json")
    ctraffic, csrc, cstale = None, None, None
    pj, cstale = kernel_pmc(cname)
    if pj is not None and pj.get("board") == n and pj.get("channels") == args.filters and pj.get("rows") and pj.get("hbm_bytes_per_launch"):
        # per-launch bytes scale with the boards of a launch (every board is read / written once): a pass at another batch
        # size is scaled linearly and labelled
        ctraffic = round(pj["hbm_bytes_per_launch"] * rows / pj["rows"])
        age = {True: "STALE: the kernel sources were edited after this pass -- not evidence for this binary; ", False: "kernel sources unchanged since the pass; ",
               None: "pass of an earlier round, no source digest recorded; "}[cstale]
        csrc = "from_profiles: profiles/" + cname + " (" + age + "separate rocprofv3 --pmc pass, not this run" + ("" if pj["rows"] == rows else f"; measured at {pj['rows']} boards per launch, scaled linearly to {rows}") + ")"
    if split:
        # fp32-class arithmetic on the f16 matrix pipe: every multiply-add of the algorithm is SPLIT_PRODUCTS f16 MFMA products (hi x hi, hi x lo,
        # lo x hi).  `achieved` counts the products the kernel has to issue -- that is what the MFMA roofline bounds -- and the
        # fp32-equivalent rate (algorithmic flops / time) is reported beside it.
        issued = SPLIT_PRODUCTS * conv_flops
        tf = issued / (conv["avg_ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS["fp16"]
        if fused:
            kname = ("k_resblock_sp<" + ("Sb17" if S_t == 17 else "Sb9") + "> (one whole split-precision ResNetBlock per launch: both 3x3 convolutions, hi + lo f16 "
                     "pairs, three f16 MFMA products per multiply, fp32 accumulation; intermediate activation in LDS, skip from the x image in LDS)")
            elem, passes = 4, 2.0  # x in, y out
        else:
            kname = (("k_conv3x3_sp2" if args.filters == 128 else "k_conv3x3_sp") if S_t == 9 else "k_conv3x3_sp17") + (
                " (split-precision 3x3 convolution of the residual tower: hi + lo f16 pairs, three f16 MFMA products per multiply, fp32 accumulation; "
                "weight-stationary" + ("; 2 x 2 split of a CU's work between its waves: 32 couts x half of cin per wave, fp32 hand-over through LDS)" if S_t == 9 and args.filters == 128 else ")"))
            elem, passes = 4, 2.5  # hi + lo f16 = 4 bytes per activation element; x in, y out, residual on every second layer
        extra = {"mfma_products_per_multiply": SPLIT_PRODUCTS, "alg_flops_per_launch": conv_flops, "issued_f16_mfma_flops_per_launch": issued,
                 "fp32_equivalent_tflops": round(conv_flops / (conv["avg_ms"] * 1e-3) / 1e12, 2), "fp32_mfma_peak_tflops": MFMA_PEAK_TFLOPS["fp32"],
                 "fp32_equivalent_over_fp32_mfma_peak": round(conv_flops / (conv["avg_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["fp32"], 3),
                 "peak_note": "2500 TFLOP/s = dense f16 MFMA peak; achieved = issued f16 MFMA products (3 per multiply of the fp32-class algorithm)"}
    else:
        tf = conv_flops / (conv["avg_ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.net_dtype]
        if fused:
            kname = "k_resblock64 (one whole ResNetBlock per launch: both 3x3 convolutions, intermediate activation in LDS, skip from the resident input tile)"
            passes = 2.0   # x in, y out
        else:
            kname = {(9, 128): "k_conv3x3_tiled", (17, 64): "k_conv3x3_t64", (9, 64): "k_conv3x3_t64",
                     (19, 256): "k_conv3x3_op19q (one launch per convolution: 64 couts x 256 cin per CU, four cin quarters meet through LDS)"}.get((S_t, args.filters), "conv3x3")
            kname += f" (weight-stationary MFMA 3x3 convolution of the residual tower, {'f16' if args.net_dtype == 'fp16' else 'bf16'})"
            passes = 2.5  # x in, y out, residual on every second layer (19x19 x 256 since round 6 too: one pass)
        elem = 2
        extra = {"alg_flops_per_launch": conv_flops}
    roofline = {"kernel": kname, "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 5),
                "traffic": ctraffic, "traffic_source": csrc, "alg_hbm_bytes_per_launch": round(rows * S_t * S_t * args.filters * elem * passes),
                "avg_launch_ms": round(conv["avg_ms"], 4),
                "avg_launch_ms_plain": round(conv["avg_ms_plain"], 4) if conv["avg_ms_plain"] is not None else None,
                "avg_launch_ms_residual": round(conv["avg_ms_residual"], 4) if conv["avg_ms_residual"] is not None else None,
                "launches_per_step": launches_per_step, "share_of_step": round(launches_per_step * conv["avg_ms"] / step_ms, 4),
                "timed_on": conv.get("data"),
                "traffic_stale": cstale}
    # annotations, not measurements of this run: what the matrix cores alone sustain with this instruction form (issue ceiling, all-zero
    # operands) and on this operand mix, measured by tools/probes/mfma_power_probe.hip and read from its committed output; and the
    # decomposition of `frac` from the PMC pass: frac ~ instruction_form_ceiling x issue_efficiency x clock_fraction
    ceil = mfma_ceilings(split)
    roofline["mfma_ceilings"] = ceil
    issued_tf = tf
    roofline["frac_algorithmic"] = round(conv_flops / (conv["avg_ms"] * 1e-3) / 1e12 / peak, 5)  # algorithmic flops / dense peak of the MFMA dtype
    if ceil:
        roofline["frac_of_issue_ceiling"] = round(issued_tf / ceil["issue_ceiling_tflops"], 4)
        roofline["instruction_form_ceiling"] = round(ceil["issue_ceiling_tflops"] / peak, 4)
        if (pj is not None and pj.get("cycles_per_mfma") and pj.get("gpu_cycles_per_launch_mean") and pj.get("rows")
                and pj.get("board") == n and pj.get("channels") == args.filters):  # (a pass of THIS kernel shape only)
            issue_cycles = ceil["pipe_cycles_per_mfma_at_peak"] / (ceil["issue_ceiling_tflops"] / peak)  # cycles between two issues of the form at its ceiling
            roofline["cycles_per_mfma"] = pj["cycles_per_mfma"]
            roofline["issue_efficiency"] = round(issue_cycles / pj["cycles_per_mfma"], 4)
            # the clock THIS run's launches ran at: shader cycles of a launch (a property of the code path, from the PMC pass, scaled to this
            # run's boards per launch) / this run's measured launch time
            clk = pj["gpu_cycles_per_launch_mean"] * rows / pj["rows"] / (conv["avg_ms"] * 1e-3) / 1e9
            roofline["effective_clock_GHz"] = round(clk, 4)
            roofline["clock_fraction"] = round(clk / MFMA_PEAK_CLOCK_GHZ, 4)
            # MFMAs the kernel issues per MFMA the algorithm needs (halo rows recomputed by the fused 17x17 block, padding slots of the column
            # tiles): the counters see the former, `frac` counts the latter
            flop_per_mfma = 2.0 * 16 * 16 * 32 * (ceil["pipe_cycles_per_mfma_at_peak"] / 16.0)  # 16x16x32: 16 384 flop in 16 pipe cycles; 32x32x16: 32 768 in 32
            over = (pj.get("SQ_INSTS_MFMA") or 0) * rows / pj["rows"] * flop_per_mfma / (issued_tf * conv["avg_ms"] * 1e-3 * 1e12)
            roofline["issued_over_algorithmic_mfma"] = round(over, 4) if over > 0 else None
            roofline["decomposition_product"] = round(roofline["instruction_form_ceiling"] * roofline["issue_efficiency"] * roofline["clock_fraction"] / (over if over > 0 else 1.0), 4)
            roofline["decomposition_source"] = ("from_profiles: profiles/" + cname + " (SQ_INSTS_MFMA, GRBM_GUI_ACTIVE / 8 XCDs and the launch time of the same passes: "
                                                "cycles per MFMA; effective clock = those cycles / this run's launch time; " + ("STALE" if cstale else "kernel sources unchanged" if cstale is False else "no source digest") + ")")
    roofline.update(extra)
    return roofline


def _c1_dropin_worker(args):
    """One measurement of c1_dropin in its own process -- like every reference actor (training_gomoku.py:8-17 exports OMP_NUM_THREADS=1
    before torch is imported, one process per actor) and like the CPU port it is compared with (oracle/baseline.py)."""
    kind, seconds = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    import torch

    torch.set_num_threads(1)
    from alpha_zero_amd import _lib
    from alpha_zero_amd.core.mcts_v2 import uct_search
    from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet, widen_for_kernels
    from alpha_zero_amd.envs.gomoku import GomokuEnv

    n, sims = 13, 100
    net = AlphaZeroNet((17, n, n), n * n, 10, 40, 80, gomoku=True)
    ck = os.path.join(ROOT, "tests", "golden", "gomoku13_ckpt200000_network.pt")
    weights = "random init (10 x 40, fc 80)"
    if os.path.exists(ck):
        net.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True)["network"], strict=True)
        weights = "the reference's shipped checkpoint checkpoints/gomoku/13x13/training_steps_200000.ckpt (10 x 40)"
    net = net.eval()
    note = ""
    if kind == "null":  # no evaluator at all: what one simulation costs in the engine + the host round trip (uniform priors, value 0)
        uni = np.full(n * n, 1.0 / (n * n), dtype=np.float32)

        def eval_func(state, batched=False):
            return ([uni] * state.shape[0], [0.0] * state.shape[0]) if batched else (uni, 0.0)
    elif kind == "cpu":
        @torch.no_grad()
        def eval_func(state, batched=False):  # pipeline.py:91-123
            x = torch.from_numpy(state if batched else state[None, ...]).to(dtype=torch.float32)
            logits, v = net(x)
            pi = torch.softmax(logits, dim=-1).cpu().numpy()
            v = np.squeeze(v.cpu().numpy(), axis=1).tolist()
            pi = [pi[i] for i in range(pi.shape[0])]
            return (pi, v) if batched else (pi[0], v[0])
    else:
        wnet, wnote = widen_for_kernels(net, n, torch.float32)  # 40 -> 64 filters, function-preserving: the hand-written fp32-class kernels
        inf = InferenceNet(wnet, dtype=torch.float32, binding=_lib.load()).cuda()
        note = inf.evaluator_path(n, torch.device("cuda")) + wnote
        resident = None
        if kind == "resident":  # the product's DeviceEvaluator: uct_search keeps the leaves on the device (core/mcts_v2.py _simulate_on_device)
            from alpha_zero_amd.core.evaluate import DeviceEvaluator

            resident = DeviceEvaluator(inf)

        @torch.no_grad()
        def eval_func(state, batched=False):
            x = torch.from_numpy(state if batched else state[None, ...]).to(device="cuda", dtype=torch.float32, non_blocking=True)
            pri, v = inf(x)
            pri, v = pri.float().cpu().numpy(), v.float().cpu().numpy().tolist()
            pi = [pri[i] for i in range(pri.shape[0])]
            return (pi, v) if batched else (pi[0], v[0])

    cb, inner = [0, 0.0], eval_func  # callback calls, seconds inside the callback: what is left of the wall time is the drop-in itself

    def timed_eval(state, batched=False):
        tc = time.perf_counter()
        r = inner(state, batched)
        cb[0] += 1
        cb[1] += time.perf_counter() - tc
        return r

    if kind == "resident":
        timed_eval = resident  # (no host callback to account for: the evaluator object itself is handed to uct_search)

    def play(budget):
        np.random.seed(1)
        env = GomokuEnv(board_size=n)
        moves, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            env.reset()
            root, done = None, False
            while not done and time.perf_counter() - t0 < budget:
                mv, pi, rq, cq, root = uct_search(env=env, eval_func=timed_eval, root_node=root, c_puct_base=19652.0, c_puct_init=1.25,
                                                  num_simulations=sims, root_noise=True, warm_up=not (env.steps > 16))
                _, _, done, _ = env.step(mv)
                moves += 1
        return moves / (time.perf_counter() - t0), moves, time.perf_counter() - t0

    play(0.5)  # engine creation, first launches
    cb[0], cb[1] = 0, 0.0
    c0 = resident._calls if kind == "resident" else 0
    v, m, wall = play(seconds)
    if kind == "resident":
        acct = {"forwards": resident._calls - c0, "us_per_forward_all_in": round(1e6 * wall / max(resident._calls - c0, 1), 1),
                "hip_graph_forward": bool(resident._graphs)}
    else:
        acct = {"callback_calls": cb[0], "callback_ms_per_call": round(1e3 * cb[1] / max(cb[0], 1), 4),
                "outside_callback_us_per_call": round(1e6 * (wall - cb[1]) / max(cb[0], 1), 1)}
    return v, m, weights, note, acct


def c1_dropin(seconds=8.0):
    """BASELINE configs[0] (C1) through the KEPT ENTRY POINT: `alpha_zero_amd.core.mcts_v2.uct_search` (the reference's signature,
    mcts_v2.py:301-450) in the reference actor's loop shape (pipeline.py:289-346: warm-up temperature for the first 16 moves, root noise,
    sub-tree reuse) on 13x13 Gomoku, 100 simulations per move, with the reference's shipped trained 10 x 40 checkpoint when the golden
    copy of its weights travelled (tests/golden/, data), a random network of that shape otherwise.  The tree lives on the GPU engine; the
    evaluator is the caller's callback: (a) `cpu_eval_func` = the reference's own eval_position (pipeline.py:91-123: fp32 torch-CPU module, one
    torch thread in a process of its own, as training_gomoku.py runs its actors), (b) `device_eval_func` = the product evaluator (InferenceNet,
    fp32-class kernels) behind the same HOST callback signature: one launch + one stream synchronisation per simulation (azsp_dropin_step)
    + the callback's own upload / read-back, (c) `resident_eval_func` = the same evaluator handed over as a DeviceEvaluator object
    (core/evaluate.py): the leaves never leave the device, the forward is a hipGraph between the engine's tensors, the host polls the
    status ~8 times per move.  moves/s of each.  Each measurement runs in a spawned process, like the CPU port it stands beside."""
    import multiprocessing as mp

    out = {"config": "BASELINE C1: 13x13 Gomoku, uct_search (P = 1), 100 sims/move, one game at a time, sub-tree reuse, root noise",
           "entry_point": "alpha_zero_amd.core.mcts_v2.uct_search (same signature and return tuple as the reference's mcts_v2.uct_search)",
           "host_round_trips_per_simulation": 1}
    ctx = mp.get_context("spawn")
    for kind in ("cpu", "device", "resident", "null"):
        with ctx.Pool(1) as pool:
            v, m, weights, note, acct = pool.map(_c1_dropin_worker, [(kind, seconds if kind != "null" else min(seconds, 3.0))])[0]
        out[f"{kind}_eval_func_moves_per_s"], out[f"{kind}_eval_func_moves"], out["weights"] = round(v, 3), m, weights
        out[f"{kind}_eval_func_accounting"] = acct  # where a move's time goes: inside the caller's callback vs in the drop-in (engine + round trip)
        if note:
            out["device_evaluator"] = note
    # a search of 100 simulations with sub-tree reuse runs ~70-100 new simulations: an upper bound of the engine + round-trip cost per simulation
    out["engine_step_us_per_simulation_upper_bound"] = round(1e6 / max(out["null_eval_func_moves_per_s"], 1e-9) / 100.0, 1)
    out["reference_moves_per_s_dev_container"] = 3.1  # BASELINE.md section 2 (the imported reference on this configuration; it cannot travel to the GPU box)
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    rc_fail = False
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.launch_check or torch.cuda.is_available():
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible; refusing to report fewer ranks than asked")
        import subprocess

        raise SystemExit(subprocess.call(self_launch_cmd(args, argv), env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ["HSA_ENABLE_IPC_MODE_LEGACY"])))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")
    if args.launch_check:
        return launch_check(world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback for the product path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} has no HIP device ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks, process_group = 1, None
    if "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ:
        # started by a launcher (torch.distributed.run), with ANY world size: rendezvous over RCCL.  A single-rank group is still a
        # group -- barriers, the count all_gather and the packed sample gather all execute (tests/test_nccl_single_rank.py)
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)
        rccl_ranks, process_group = dist.get_world_size(), dist.get_backend()

    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    n, game = args.board, args.game
    A = n * n + (1 if game == "go" else 0)
    torch.manual_seed(1)
    net = AlphaZeroNet((17, n, n), A, args.blocks, args.filters, args.filters, gomoku=(game != "go"))
    DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
    torch.backends.cudnn.benchmark = not args.no_miopen_find

    def make_actor(dtype_name, reuse_tree=True, split=True):
        """Engine + evaluator in the steady state: staggered openings, then the argument-independent pre-roll."""
        act = SelfPlayActor(net, game=game, board_size=n, num_games=args.games, num_simulations=args.sims, num_parallel=args.parallel,
                            warm_up_steps=16 if n <= 13 else 30, resign_threshold=-1.0, seed=1, rank=rank, device=dev, net_dtype=DT[dtype_name],
                            use_graph=not args.no_graph, engine_kw=None if reuse_tree else {"reuse_tree": False}, use_split_evaluator=split)
        if args.no_fused_block:
            act.infer.use_fused_block = False
        e = act.engine
        if args.stagger > 0:  # mixed game phases from the first round (documented in DESIGN.md "Measurement")
            rng = np.random.Generator(np.random.PCG64(1234 + rank))
            plies = rng.integers(0, args.stagger + 1, size=args.games)
            out = e.env_step(None)
            for t in range(int(plies.max())):
                legal = out["legal"][:, : n * n].astype(bool)
                r = rng.random(legal.shape) * legal
                acts = np.where((plies > t) & legal.any(axis=1) & (out["scalars"][:, 5] == 0), r.argmax(axis=1), -2).astype(np.int32)
                out = e.env_step(acts)
        return act

    actor = make_actor(args.net_dtype)
    eng = actor.engine
    preroll_rounds = preroll(actor, args, world, dev)
    elapsed, cnt, evs, samples_at_root = timed(actor, args, world, dev, args.warmup, args.steps)
    # range record of the HEADLINE actor's evaluator over pre-roll + warm-up + timed region, read before anything else touches the network
    erange = evaluator_range(actor)
    bk_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))  # expand/backup + end-of-move kernels
    k_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))   # select kernel (the dominant hand-written kernel)
    nn_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))

    elapsed_max, total_moves, total_sims, total_evals = reduce_totals(cnt, elapsed, world, dev)
    ranks = per_rank_report(cnt, elapsed, world, dev)
    if total_moves <= 0:
        raise SystemExit("bench.py: no move was committed inside the timed region -- the pre-roll did not reach the steady state")

    # ---- dominant kernel (the tower convolution): average launch duration, HIP events on the launch stream -----------
    conv = tower_replay(actor, args, dev) if rank == 0 else None
    replay_range = None
    if rank == 0 and erange is not None:  # (the replay runs the forward's own launches on its own data: expected 0 too); leaves the record clean
        ev_r, mx_r = actor.infer.split_range_status(reset=True)
        replay_range = {"events": ev_r, "largest_abs": mx_r * 2.0 ** actor.infer.act_shift}

    if args.split_round and rank == 0:
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ta = tb = 0.0
        for _ in range(20):
            ea[0].record()
            eng.expand_backup()
            ea[1].record()
            eng.select()
            ea[2].record()
            actor._graph.replay() if actor._graph is not None else actor._forward()
            torch.cuda.synchronize(dev)
            ta += ea[0].elapsed_time(ea[1]) / 20
            tb += ea[1].elapsed_time(ea[2]) / 20
        print(json.dumps({"split_round_ms": {"expand_backup_endmove": round(ta, 4), "select_features": round(tb, 4)}}), flush=True)
    if rank == 0:
        # ---- roofline of the dominant hand-written kernel: the fused round kernel (HBM bound) ----------
        e_bytes = {"bf16": 2, "fp16": 2, "fp32": 4}[args.net_dtype]
        W = (n * n + 63) // 64
        steps = max(1, args.steps)
        expanded = cnt["leaves"] - cnt["dup_leaves"] + cnt["root_evals"]
        created = cnt["nodes_created"]
        vloss_edges = cnt["backup_edges"] - (cnt["sims"] - cnt["leaves"] + expanded)  # informational
        # select kernel, reference-equivalent dense layout: N, W, P rows of every visited node, the new
        # node's position, virtual loss read-modify-write of W per path edge, the leaf's 17 planes + 8-board history
        # observation planes per leaf: 17 planes in the network dtype, or (tiled evaluator layout) 3 written 8-channel chunks
        feat_bytes = 3 * n * n * 16 if (actor.tiled_features or getattr(actor, "split_features", False)) else 17 * n * n * e_bytes
        alg_bytes = (cnt["node_visits"] * 12 * A + created * 64 + cnt["leaves"] * (cnt["backup_edges"] / max(1, cnt["sims"])) * 8
                     + (cnt["leaves"] + cnt["root_evals"]) * (feat_bytes + 16 * W * 8)) / steps
        # expand/backup kernel: prior + value in, P/N/W rows out, N and W read-modify-write per path edge (+ vloss revert)
        bk_bytes = (expanded * (12 * A + 4 * A + 4) + cnt["backup_edges"] * 16 + cnt["leaves"] * 8 * 4.5) / steps
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic, tsrc = None, None
        prof = os.path.join(ROOT, "profiles", "select_kernel_pmc.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                if pj.get("games") == args.games and pj.get("board") == n:
                    traffic, tsrc = pj.get("hbm_bytes_per_launch"), "from_profiles: profiles/select_kernel_pmc.json (separate rocprofv3 --pmc pass, not this run)"
            except Exception:
                traffic = None
        engine_roof = {"kernel": "k_game<OpSelect> (PUCT descents + virtual loss + observation planes)", "bound": "hbm",
                       "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                       "traffic": traffic, "traffic_source": tsrc, "alg_bytes_per_launch": round(alg_bytes), "avg_launch_ms": round(k_ms, 4),
                       "share_of_step": round(k_ms / (bk_ms + k_ms + nn_ms), 4),
                       "backup_kernels": {"avg_ms": round(bk_ms, 4), "alg_bytes_per_launch": round(bk_bytes),
                                          "achieved_GBs": round(bk_bytes / (bk_ms * 1e-3) / 1e9, 2)}}
        flops_eval = net_flops_per_eval(n, A, args.blocks, args.filters, args.filters, game != "go")
        nn_tflops = flops_eval * args.games * args.parallel / (nn_ms * 1e-3) / 1e12
        split_eval = conv is not None and conv["split"]
        # fp32-class evaluator: every multiply is SPLIT_PRODUCTS f16 MFMA products, so the bound on ALGORITHMIC flops is the f16 peak / 3
        peak = MFMA_PEAK_TFLOPS["fp16"] / SPLIT_PRODUCTS if split_eval else MFMA_PEAK_TFLOPS[args.net_dtype]
        nn_roof = {"kernel": "whole evaluator forward on G*P rows (stem + tower + heads)" + (", algorithmic (fp32-equivalent) flops; peak = f16 MFMA peak / 3 products" if split_eval else ""),
                   "bound": "mfma", "achieved": round(nn_tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                   "frac": round(nn_tflops / peak, 5), "avg_forward_ms": round(nn_ms, 3),
                   "share_of_step": round(nn_ms / (bk_ms + k_ms + nn_ms), 4),
                   "batch_fill": round((cnt["leaves"] + cnt["root_evals"]) / (steps * args.games * args.parallel), 4)}
        roofline = tower_roofline(conv, args, bk_ms + k_ms + nn_ms) if conv is not None else engine_roof
        # ---- companions (world == 1 only): the same engine and workload with another evaluator -- labelled, never `value` ----------
        lowp = fp32_lib = None
        if world == 1 and not args.no_companions:
            def companion(dtype_name, split, steps, pre, label):
                a = make_actor(dtype_name, split=split)
                prer = preroll(a, args, world, dev, min_rounds=pre)
                el, c, ev, _ = timed(a, args, world, dev, 5, steps)
                path = a.infer.evaluator_path(n, dev)
                cv = tower_replay(a, argparse.Namespace(**{**vars(args), "net_dtype": dtype_name}), dev, reps=2) if (dtype_name != "fp32" or split) else None
                r = {"moves_per_s": round(c["moves"] / el, 2), "sims_per_sec": round(c["sims"] / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
                     "sims_per_move": round(c["sims"] / max(1, c["moves"]), 2), "steps": steps, "warmup": 5, "preroll_rounds": prer, "protocol": label,
                     "forward_ms": round(float(np.mean([e[2].elapsed_time(e[3]) for e in ev])), 3), "dtype": DTYPE_LABEL[dtype_name] if (dtype_name != "fp32" or split) else "fp32 (library convolutions) / f32-f64 tree",
                     "evaluator": path,
                     "tower_kernel_avg_launch_ms": round(cv["avg_ms"], 4) if cv else None}
                del a, ev
                torch.cuda.empty_cache()
                return r

            if args.net_dtype == "fp32":
                # (a) the lower-precision evaluator (bf16 hand-written kernels): what rounds 1-3 reported as the headline.  Narrower arithmetic
                # than the reference's, bounded against it in tests/test_precision_parity.py; full protocol (same pre-roll as the headline)
                lowp = companion("bf16", True, args.steps, args.preroll_rounds, "full (the headline's pre-roll and window)")
                # (b) the reference's precision on LIBRARY convolutions (+ azsp_bias_act): what the hand-written fp32-class kernels replace
                if conv is not None and conv["split"]:
                    fp32_lib = companion("fp32", False, min(20, args.steps), 30, "short protocol (30-round pre-roll, <= 20 steps: 180 ms per round)")
            else:
                # a lower-precision run: the reference-precision evaluator as the companion
                lowp = companion("fp32", True, min(40, args.steps), 60, "short protocol (60-round pre-roll, <= 40 steps)")
        fresh = None
        if world == 1 and not args.no_fresh_tree:
            # SURVEY 8d's "fresh-tree" variant: sub-tree reuse off (mcts_v2.py:436-446 never runs), every move pays the full budget of
            # sims + P root visits instead of inheriting the chosen child's (mcts_v2.py:378 semantics discounted in `value`).  Without
            # inheritance every move takes the same number of rounds, so the slots stay in phase: the window is 5 whole move periods.
            period = (args.sims + args.parallel + args.parallel - 1) // args.parallel + 1
            af = make_actor(args.net_dtype, reuse_tree=False)
            af.run_rounds(2 * period)
            elf, cf, evf, _ = timed(af, args, world, dev, period, 5 * period)
            fresh = {"moves_per_s": round(cf["moves"] / elf, 2), "sims_per_sec": round(cf["sims"] / elf, 1), "sims_per_move": round(cf["sims"] / max(1, cf["moves"]), 2),
                     "ms_per_step": round(elf / (5 * period) * 1e3, 3), "steps": 5 * period, "warmup": period, "reuse_tree": False,
                     "note": "same engine / evaluator / workload with sub-tree reuse disabled: the upper-work variant, labelled, never `value`"}
            del af, evf
            torch.cuda.empty_cache()
        cpu, c1, c1d = None, None, None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import baseline

            cores = args.cpu_cores or usable_host_cores()
            cpu = baseline.run(cores, seconds=args.cpu_seconds, game=game, n=n, sims=args.sims, P=args.parallel, blocks=args.blocks,
                               filters=args.filters, stagger=args.stagger)
            cpu["value"] = round(cpu["value"], 3)
            cpu["per_core"] = round(cpu["per_core"], 4)
            # BASELINE configs[0] (C1): the reference's own CPU-runnable case -- 13x13 Gomoku, mcts_v2.uct_search (P = 1), ONE actor, 100
            # sims/move, 6x64 net -- timed with the same port on one host core for a few seconds (the reference measured
            # 2.7 moves/s in the development container)
            c1 = baseline.run(1, seconds=min(8.0, args.cpu_seconds), game="gomoku", n=13, sims=100, P=1, blocks=6, filters=64, stagger=0)
            c1["value"], c1["per_core"] = round(c1["value"], 3), round(c1["per_core"], 4)
            c1["config"] = "BASELINE C1: 13x13 Gomoku, uct_search, 1 CPU self-play actor, 100 sims/move (reference path, no GPU)"
            # the same port on the shape of the checkpoint C1 names (10 x 40), and the kept entry point on the GPU engine beside it
            ck = os.path.join(ROOT, "tests", "golden", "gomoku13_ckpt200000_network.pt")  # the same trained weights c1_dropin evaluates
            c1b = baseline.run(1, seconds=min(8.0, args.cpu_seconds), game="gomoku", n=13, sims=100, P=1, blocks=10, filters=40, stagger=0, fc=80,
                               ckpt=ck if os.path.exists(ck) else None)
            try:
                c1d = c1_dropin(seconds=min(8.0, args.cpu_seconds))
                c1d["cpu_port_moves_per_s_10x40"] = round(c1b["value"], 3)
                c1d["cpu_port_sample"] = c1b["sample"]
                c1d["cpu_port_accounting"] = {k: c1b[k] for k in ("callback_calls", "callback_ms_per_call", "outside_callback_us_per_call")}
                c1d["cpu_eval_func_over_cpu_port"] = round(c1d["cpu_eval_func_moves_per_s"] / max(c1b["value"], 1e-9), 3)
                c1d["device_eval_func_over_cpu_port"] = round(c1d["device_eval_func_moves_per_s"] / max(c1b["value"], 1e-9), 3)
                c1d["resident_eval_func_over_cpu_port"] = round(c1d["resident_eval_func_moves_per_s"] / max(c1b["value"], 1e-9), 3)
            except Exception as ex:  # the companion must never take the headline down
                c1d = {"error": repr(ex)}
            # port vs the imported reference on identical seeded moves, measured in the development container by
            # tools/calibrate_baseline.py (the reference cannot travel to the GPU box): ratio = port moves/s / reference moves/s
            cal = os.path.join(ROOT, "tests", "golden", "cpu_baseline_calibration.json")
            if os.path.exists(cal):
                cj = json.load(open(cal))
                key = f"{game}{n}_p{args.parallel}_s{args.sims}_{args.blocks}x{args.filters}"
                ratio = cj.get("ratios", {}).get(key, {}).get("ratio")
                cpu["calibration_ratio"] = ratio
                cpu["reference_equivalent_value"] = round(cpu["value"] / ratio, 3) if ratio else None
                cpu["calibration"] = "port / reference moves/s on the same seeded moves, tools/calibrate_baseline.py, key " + key
        line = {
            "metric": "self-play moves/sec (whole node), 9x9 Go @ 200 sims/move" if (game == "go" and n == 9 and args.sims == 200)
            else f"self-play moves/sec (whole node), {n}x{n} {game} @ {args.sims} sims/move",
            "value": round(total_moves / elapsed_max, 2), "unit": "moves/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "process_group": process_group,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.net_dtype] if (args.net_dtype != "fp32" or (conv is not None and conv["split"])) else (
                "fp32-class tower (f16 hi+lo pairs, 3 MFMA products, fp32 accumulate) behind library fp32 stem and heads / f32-f64 tree"
                if "split-precision tower" in actor.evaluator_path else "fp32 (library convolutions) / f32-f64 tree"),
            "data": "synthetic",
            "config": {"workload": f"{n}x{n} {game}, {args.games} games/GPU, {args.sims} sims/move (reference budget semantics), P={args.parallel}, "
                                   f"{args.blocks}x{args.filters} net", "net_dtype": args.net_dtype, "tree_dtype": "f32 (f64 noisy root)",
                       "evaluator": actor.evaluator_path,
                       "games_per_gpu": args.games, "stagger_plies": args.stagger, "hip_graph_forward": not args.no_graph,
                       "parallelism": f"games sharded x{world}, sample gather to rank 0"},
            "sims_per_sec": round(total_sims / elapsed_max, 1), "evals_per_sec": round(total_evals / elapsed_max, 1),
            "sims_per_move": round(total_sims / max(1.0, total_moves), 2),
            "select_nodes_per_sim": round(cnt["node_visits"] / max(1, cnt["sims"]), 3),
            "backup_nodes_per_sim": round(cnt["backup_edges"] / max(1, cnt["sims"]), 3),
            "select_hint_prefetches_per_sim": round(cnt.get("hint_prefetches", 0) / max(1, cnt["sims"]), 3),
            "select_hint_hit_rate": round(cnt.get("hint_hits", 0) / max(1, cnt.get("hint_prefetches", 0)), 3),
            "serial_step_ms": round(bk_ms + k_ms + nn_ms, 3),
            "samples_gathered": samples_at_root, "preroll_rounds": preroll_rounds, "per_rank": ranks,
            "dup_leaf_rate": round(cnt["dup_leaves"] / max(1, cnt["leaves"]), 5), "terminal_hit_rate": round(cnt["terminal_hits"] / max(1, cnt["sims"]), 5),
            ("bf16_moves_per_s" if args.net_dtype == "fp32" else "fp32_moves_per_s"): lowp["moves_per_s"] if lowp else None,
            ("lower_precision_companion" if args.net_dtype == "fp32" else "fp32_companion"): lowp,
            "fp32_library_moves_per_s": fp32_lib["moves_per_s"] if fp32_lib else None, "fp32_library_companion": fp32_lib,
            "speedup_vs_library_fp32": round(total_moves / elapsed_max / fp32_lib["moves_per_s"], 2) if fp32_lib and fp32_lib["moves_per_s"] > 0 else None,
            "fresh_tree_moves_per_s": fresh["moves_per_s"] if fresh else None, "fresh_tree_companion": fresh,
            "speedup_vs_cpu_baseline": round(total_moves / elapsed_max / cpu["value"], 1) if cpu and cpu["value"] > 0 else None,
            # the same ratio against the REFERENCE's own speed: the CPU baseline is a port (C rules) that runs calibration_ratio x faster than
            # the imported reference on identical seeded moves (tools/calibrate_baseline.py) -- the raw ratio above is the conservative one
            "speedup_vs_reference_equivalent_cpu": (round(total_moves / elapsed_max / cpu["reference_equivalent_value"], 1)
                                                    if cpu and cpu.get("reference_equivalent_value") else None),
            "roofline": roofline, "engine_roofline": engine_roof, "nn_roofline": nn_roof,
            "cpu_baseline": cpu if world == 1 else "measured at N=1 only", "c1_cpu_reference_path": c1, "c1_dropin": c1d,
            # fp32-class evaluator: lanes that met a value beyond the f16-pair format's range during pre-roll + warm-up + timed region of
            # THIS run's headline actor (its own range record, read before the kernel replay); must be 0 for the line to be valid
            "evaluator_range_events": erange["events"] if erange else None, "evaluator_range": erange,
            "kernel_replay_range_events": replay_range["events"] if replay_range else None,
            # weak scaling: value(N) = sum over ranks of moves / max over ranks of time; the only inter-rank work is harvest + gather
            "expected_value_formula": "value(N) ~= N * value(1) * (1 - per_rank.harvest_gather_share_of_time)" if world > 1 else None,
        }
        print(json.dumps(line), flush=True)
        if erange and erange["events"]:
            # the timed forward clamped activations: the number is not the fp32-class evaluator's (the actor has rescaled / fallen back
            # for the rounds after the event, see evaluator_range) -- fail loudly instead of reporting it as valid
            print(f"bench.py: ERROR the evaluator clamped {erange['events']} activation lanes inside the measured run", file=sys.stderr, flush=True)
            rc_fail = True
    if pg_active():
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    if rc_fail:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
