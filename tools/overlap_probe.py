"""Can the engine kernels (select / expand-backup) hide behind the evaluator forward on a second stream?  ("Next" #2)

Times, on the BASELINE C3 workload after a short pre-roll:  (a) the forward alone, (b) expand/backup + end-of-move + select alone,
(c) both launched together on two streams.  If (c) ~ (a) + (b) the two do not overlap: the weight-stationary convolution kernels
hold one persistent workgroup per CU that owns all 512 registers of every SIMD and 160 KB of LDS, so no other wave can become
resident on a CU while a convolution runs.  Timing probe only: the forward of (c) reads features the select kernel is rewriting."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402

torch.manual_seed(1)
net = AlphaZeroNet((17, 9, 9), 82, 10, 128, 128)
act = SelfPlayActor(net, game="go", board_size=9, num_games=4096, num_simulations=200, num_parallel=8, resign_threshold=-1.0, seed=1,
                    device="cuda", use_graph=True, net_dtype=torch.bfloat16)
act.run_rounds(120)
torch.cuda.synchronize()
e = act.engine
s_main, s_eng = torch.cuda.current_stream(), torch.cuda.Stream()
REPS = 20


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s_main)
    for _ in range(REPS):
        fn()
    s_main.wait_stream(s_eng)
    e1.record(s_main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


def fwd():
    act._graph.replay()


def eng():
    e.expand_backup()
    e.select()


def eng_side():
    with torch.cuda.stream(s_eng):
        e.expand_backup()
        e.select()


def both():
    eng_side()
    act._graph.replay()


def serial():
    eng()
    fwd()


out = {"forward_ms": timed(fwd), "engine_ms": timed(eng), "serial_ms": timed(serial), "two_streams_ms": timed(both)}
out["hidden_ms"] = out["forward_ms"] + out["engine_ms"] - out["two_streams_ms"]
print(json.dumps({k: round(v, 4) for k, v in out.items()}))
