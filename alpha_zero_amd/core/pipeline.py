"""Self-play actor on the batched engine (reference: alpha_zero/core/pipeline.py:83-382).

`game_stats_from_row` rebuilds the per-game `stats` dict of play_and_record_one_game
(pipeline.py:367-380) from the 16-int game record the engine emits at harvest time.
"""
from typing import Any, Dict

import numpy as np


def _result_string(row, game, komi):
    """go.py:194-200 + go_engine.py:527-534 / gomoku.py:138-147"""
    winner = int(row[2])
    if game == "go":
        if int(row[6]):  # resigned
            return "B+R" if winner == 1 else "W+R"
        score = float(int(row[3])) - (float(int(row[4])) + komi)
        if score > 0:
            return "B+" + "%.1f" % score
        if score < 0:
            return "W+" + "%.1f" % abs(score)
        return "DRAW"
    return "B+1.0" if winner == 1 else "W+1.0" if winner == 2 else "DRAW"


def game_stats_from_row(row, game: str, komi: float = 7.5, resign_threshold: float = -1.0) -> Dict[str, Any]:
    stats: Dict[str, Any] = {"game_length": int(row[1]), "game_result": _result_string(row, game, komi)}
    if game == "go":  # has_pass_move / has_resign_move (pipeline.py:372-380)
        stats["num_passes"] = int(row[5])
        stats["is_resign_disabled"] = bool(row[7])
        stats["is_marked_for_resign"] = bool(row[8])
        stats["is_could_won"] = bool(row[9])
        mp = int(row[10])
        stats["marked_resign_player"] = "B" if mp == 1 else "W" if mp == -1 else None
        stats["resign_threshold"] = resign_threshold
    return stats


# =====================================================================================================
# Batched self-play actor
# =====================================================================================================
import time  # noqa: E402

import torch  # noqa: E402

from .. import _abi  # noqa: E402
from .engine import Engine, EngineConfig  # noqa: E402
from .network import AlphaZeroNet, InferenceNet, widen_for_kernels  # noqa: E402
from .replay import Transition  # noqa: E402

_FEAT_OF = {torch.float32: _abi.FEAT_F32, torch.bfloat16: _abi.FEAT_BF16, torch.float16: _abi.FEAT_F16}


class ClampWindow:
    """Which games may hold a move that was searched on a CLAMPING fp32-class evaluator (Weak #1).  The range record says
    that some kernel lane clamped since the last poll, not which board: every game that ran a round inside the window
    (last clean poll, detecting poll] is suspect -- the games in progress at the detecting poll and the finished games not handed out
    yet.  The reference's fp32 forward (pipeline.py:102-109) has no such window, so none of their samples may reach the learner
    unmarked.  Games are identified by the engine's uid = (games finished before in that slot) * G + slot."""

    _NONE_LO, _NONE_HI = np.iinfo(np.int64).max, -1

    def __init__(self, num_games):
        self.G = int(num_games)
        self.next_unharvested = np.zeros(self.G, dtype=np.int64)  # per slot: first game index no harvest has handed out yet
        self.lo = np.full(self.G, self._NONE_LO, dtype=np.int64)   # per slot: suspect game indices [lo, hi]
        self.hi = np.full(self.G, self._NONE_HI, dtype=np.int64)
        self.events = 0

    def on_event(self, games_done):
        """games_done int[G]: azsp_get_status column 5 at the detecting poll (= the index of each slot's game in progress)."""
        gd = np.asarray(games_done, dtype=np.int64).reshape(self.G)
        self.lo = np.minimum(self.lo, self.next_unharvested)
        self.hi = np.maximum(self.hi, gd)
        self.events += 1

    def mask(self, uids):
        """bool per harvested game (uids = column 11 of the harvest rows): suspect?  Advances the per-slot harvest front."""
        u = np.asarray(uids, dtype=np.int64)
        slot, idx = u % self.G, u // self.G
        m = (idx >= self.lo[slot]) & (idx <= self.hi[slot])
        np.maximum.at(self.next_unharvested, slot, idx + 1)
        spent = self.next_unharvested > self.hi  # intervals wholly behind the harvest front are used up
        self.lo[spent], self.hi[spent] = self._NONE_LO, self._NONE_HI
        return m


class SelfPlayActor:
    """G concurrent self-play games on one GPU: replaces G `run_selfplay_actor_loop` processes
    (pipeline.py:166-286).  One round = engine kernel (expand/backup of the previous leaf batch, end-of-move
    work, selection of the next P leaves per game, observation planes) + one network forward on G*P rows.
    Nothing returns to the host during a round; finished games are collected with `harvest()`."""

    def __init__(self, network: AlphaZeroNet, *, game="go", board_size=9, num_games=4096, num_simulations=200, num_parallel=8,
                 c_puct_base=19652.0, c_puct_init=1.25, warm_up_steps=16, check_resign_after_steps=40, disable_resign_ratio=0.1,
                 resign_threshold=-1.0, komi=7.5, num_to_win=5, seed=1, rank=0, device="cuda", net_dtype=torch.float32,
                 use_graph=True, training_steps=0, binding=None, root_noise=True, deterministic=False, tiled_features=None, engine_kw=None,
                 use_split_evaluator=True, auto_widen=None):
        """net_dtype: precision class of the leaf evaluator.  The default is the REFERENCE'S: fp32 (pipeline.py:91-123 evaluates in fp32,
        no autocast anywhere) -- on the hand-written split-precision kernels (hi + lo f16 pairs, three MFMA products, fp32 accumulation:
        include/azsp.h azsp_conv3x3_split) for 9x9 x {128, 64} and 13x13 Gomoku x 64 networks, on library fp32 convolutions (announced
        by a RuntimeWarning) for any other shape.  torch.bfloat16 / torch.float16 are the opt-in lower-precision evaluators.
        tiled_features: None = the engine writes its observation planes in the evaluator's own input layout whenever the network / board
        shape has hand-written kernels (tiled bf16 / f16, or the fp32-class stem's split layout); False = always NCHW planes.
        use_split_evaluator: False = an fp32 network is evaluated by the LIBRARY's fp32 convolutions even where the split-precision
        kernels exist (comparison runs: bench.py's fp32_library_companion, tests/test_precision_parity.py).
        auto_widen: networks of a width without hand-written kernels run as a function-preserving widened copy (network.widen_for_kernels).
        None = on the GPU whenever the widened network actually reaches hand-written kernels (not for library comparison runs, not when
        the caller disabled the kernels' feature layouts); True / False force it."""
        from .. import _lib

        self.binding = binding or _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.game, self.komi, self.resign_threshold = game, komi, resign_threshold
        self.net_dtype = net_dtype
        self.use_graph = use_graph and self.device.type == "cuda"
        self.use_split_evaluator = bool(use_split_evaluator)
        if auto_widen is None:  # widening only pays when the wider network lands on hand-written kernels
            auto_widen = (self.device.type == "cuda" and tiled_features is not False
                          and (net_dtype != torch.float32 or self.use_split_evaluator))
        self.auto_widen = bool(auto_widen)
        self.board_size = board_size
        wnet, self._widen_note = widen_for_kernels(network, board_size, net_dtype) if self.auto_widen else (network, "")
        probe = InferenceNet(wnet, dtype=net_dtype, binding=self.binding if self.device.type == "cuda" else None)
        probe.use_split_tower = self.use_split_evaluator
        self.tiled_features = probe.supports_tiled_features(board_size, self.device) if tiled_features is None else bool(tiled_features)
        # fp32-class evaluator: the engine writes the stem's input layout itself (AZSP_FEAT_F16_SPLIT; 0 / 1 planes are exact f16 values)
        self.split_features = tiled_features is None and not self.tiled_features and probe.supports_split_features(board_size, self.device)
        self.evaluator_path = (probe.evaluator_path(board_size, self.device) if self.tiled_features or tiled_features is None else
                               "library stem (tiled features disabled by the caller)") + self._widen_note
        if self.device.type == "cuda" and "hand-written" not in self.evaluator_path:
            import warnings

            warnings.warn(f"alpha_zero_amd: evaluator falls back to {self.evaluator_path}", RuntimeWarning, stacklevel=2)
        self.cfg = EngineConfig(
            game=game, board_size=board_size, num_games=num_games, num_parallel=num_parallel, num_simulations=num_simulations,
            c_puct_base=c_puct_base, c_puct_init=c_puct_init, warm_up_steps=warm_up_steps, komi=komi, num_to_win=num_to_win,
            resign_threshold=resign_threshold, check_resign_after_steps=check_resign_after_steps,
            disable_resign_ratio=disable_resign_ratio, root_noise=root_noise, deterministic=deterministic,
            feature_dtype=((_abi.FEAT_F16_TILED if net_dtype == torch.float16 else _abi.FEAT_BF16_TILED) if self.tiled_features
                           else _abi.FEAT_F16_SPLIT if self.split_features else _FEAT_OF[net_dtype]),
            training_steps=training_steps, seed=seed, rank=rank,
            device_index=self.device.index or 0)
        for k, v in (engine_kw or {}).items():  # further EngineConfig fields (move logs, max_plies, ...: tests and diagnostics)
            if not hasattr(self.cfg, k):
                raise TypeError(f"unknown engine option {k}")
            setattr(self.cfg, k, v)
        self.engine = Engine(self.binding, self.cfg, device=self.device)
        self.engine.reset_games()
        self._graph = None
        self.rounds = 0
        self.straddled_games = 0  # harvested games that were in progress across a weight hot-swap (see harvest())
        # fp32-class evaluator: clamped out-of-range activations seen so far and the rescalings they triggered (_check_evaluator_range)
        self.range_events, self.range_max_abs, self.range_rescales = 0, 0.0, 0
        # games that ran a round between a clamp event and its repair (ClampWindow): harvest() marks them `evaluator_clamped` or, with
        # drop_clamped_games = True (run_selfplay_actor_loop), drops them; harvest_tensors() leaves a bool per game in last_harvest_clamped
        self.clamp_window = ClampWindow(num_games)
        self.drop_clamped_games, self.clamped_games, self.last_harvest_clamped = False, 0, np.zeros(0, dtype=bool)
        self.drop_straddling_games = False
        self.set_network(network, training_steps)

    # -- weights ---------------------------------------------------------------------------------------
    def set_network(self, network: AlphaZeroNet, training_steps=0):
        """Checkpoint hot-swap (pipeline.py:232-239): new weights take effect at the next round."""
        if self.auto_widen:
            network, _ = widen_for_kernels(network, self.board_size, self.net_dtype)
        self.infer = InferenceNet(network, dtype=self.net_dtype, binding=self.binding if self.device.type == "cuda" else None).to(self.device)
        self.infer.use_split_tower = self.use_split_evaluator
        self.training_steps = training_steps
        self.engine.set_actor_state(self.resign_threshold, training_steps)  # games that start from now on carry this tag
        self._graph = None
        if getattr(self, "rounds", 0) and "library stem (tiled features disabled" not in self.evaluator_path:
            # a hot-swapped network starts uncalibrated at shift 0: drop the previous network's "activations carried x 2^-k" note
            self.evaluator_path = self.infer.evaluator_path(self.board_size, self.device) + self._widen_note

    def set_resign_threshold(self, resign_threshold):
        """var_resign_threshold as the reference actor reads it before EVERY game (pipeline.py:241-242): games that start after this
        call use (and report) the new value, games in progress keep theirs.  <= -1 disables resignation (pipeline.py:449-459)."""
        self.resign_threshold = float(resign_threshold)
        self.engine.set_actor_state(self.resign_threshold, self.training_steps)

    def _forward(self):
        e = self.engine
        if e.features_tiled:
            self.infer.forward_tiled(e.features, e.rows, e.N, e.priors, e.values)
        elif e.features_split:
            if self.infer.split_fallback_reason or self.infer.stem_fallback_reason:  # the fp32-class kernels (or their stem) were given up for this network
                self.infer._forward_after_split_fallback(e.features, e.priors, e.values, (e.rows, e.N))
                return
            if not self.infer.supports_split_features(e.N, self.device):  # (someone switched the split kernels off on the live InferenceNet)
                raise RuntimeError("the engine writes the split-precision stem's input layout; build the actor with use_split_evaluator=False "
                                   "to evaluate an fp32 network on the library")
            self.infer.forward_split(e.features, e.priors, e.values, split_features=(e.rows, e.N))
        else:
            self.infer(e.features, e.priors, e.values)

    def _calibrate(self):
        """fp32-class evaluator: one calibration pass on the leaf batch the engine has just written (InferenceNet.calibrate_activation_scale)
        fixes the power-of-two activation scale of this network BEFORE its first forward is used (and before a hipGraph is captured)."""
        e, inf = self.engine, self.infer
        if not (e.features_split and not inf.act_calibrated and inf.supports_split_features(e.N, self.device)):
            return
        inf.calibrate_activation_scale(e.features, split_features=(e.rows, e.N), slot=0)
        self._note_evaluator_path()

    def _note_evaluator_path(self):
        inf = self.infer
        if inf.split_fallback_reason:
            import warnings

            self.evaluator_path = inf.evaluator_path(self.board_size, self.device) + self._widen_note
            warnings.warn(f"alpha_zero_amd: evaluator falls back to {self.evaluator_path}", RuntimeWarning, stacklevel=3)
            self._graph = None
        elif inf.act_shift and "activations carried" not in self.evaluator_path:
            self.evaluator_path += f"; activations carried x 2^-{inf.act_shift} (exact rescaling, calibrated max |v| = {inf.act_max_abs:.4g})"

    def _capture(self):
        e = self.engine
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(3):  # let MIOpen pick its kernels before capture
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._forward()
        self._graph = g

    # -- rounds ----------------------------------------------------------------------------------------
    def run_round(self, evs=None):
        """evs: optional 4 torch.cuda.Events recorded around (expand/backup + end-of-move), select, forward."""
        e = self.engine
        if evs is not None:
            evs[0].record()
        e.expand_backup()
        if evs is not None:
            evs[1].record()
        e.select()
        if evs is not None:
            evs[2].record()
        if not self.infer.act_calibrated:
            self._calibrate()
        if self.use_graph:
            if self._graph is None:
                self._capture()
            self._graph.replay()
        else:
            self._forward()
        if evs is not None:
            evs[3].record()
        self.rounds += 1

    def run_rounds(self, n):
        for _ in range(n):
            self.run_round()

    # -- output ----------------------------------------------------------------------------------------
    def harvest_tensors(self, clone=False):
        """(states, pi, z, games) of the finished games as device tensors.  The tensors are views of buffers the engine re-uses: the
        next harvest overwrites them in place.  clone=True returns private copies (for consumers that keep them across harvests,
        e.g. an asynchronous learner or a replay insert on another stream)."""
        st, pi, z, games = self.engine.harvest()
        self._check_evaluator_range()
        self.last_harvest_clamped = self.clamp_window.mask(games[:, 11]) if len(games) else np.zeros(0, dtype=bool)
        self.clamped_games += int(self.last_harvest_clamped.sum())
        return (st.clone(), pi.clone(), z.clone(), games) if clone else (st, pi, z, games)

    def poll_evaluator_range(self):
        """Poll the fp32-class evaluator's range record now (two words; synchronises the stream) instead of waiting for the next
        harvest: an event is repaired at once and the games that ran inside the window are remembered (ClampWindow).  The actor loop
        calls this every few rounds, which bounds the window by that many rounds.  Returns the events seen so far."""
        self._check_evaluator_range()
        return self.range_events

    def _check_evaluator_range(self):
        """fp32-class evaluator: its kernels carry values as f16 pairs and clamp what exceeds +-65504 (in units of 2^act_shift) -- the
        reference's fp32 network would carry such a value on.  The kernels record every such event in THIS network's range record
        (InferenceNet.range_rec; another actor or evaluator in the same process has its own); it is polled here, once per harvest (the
        harvest has synchronised the stream already).  An event is counted, announced and ACTED on: the network is re-calibrated on
        the current leaf batch (a larger exact power-of-two activation scale), or, if the format cannot carry it, handed to the
        library's fp32 convolutions -- the actor never keeps playing on a clamping evaluator PAST THE POLL; what ran
        between the event and the poll is marked (ClampWindow: harvest() marks or drops those games)."""
        inf = self.infer
        if (self.device.type != "cuda" or self.net_dtype != torch.float32 or inf.binding is None or not inf.use_split_tower
                or inf.split_fallback_reason or not hasattr(inf, "range_rec")):
            return
        ev, mx = inf.split_range_status(reset=True)
        if not ev:
            return
        import warnings

        self.range_events += ev
        self.range_max_abs = max(self.range_max_abs, mx * 2.0 ** inf.act_shift)
        old_shift = inf.act_shift
        e = self.engine
        # every game that ran a round since the last clean poll is suspect: the ones in progress now + the finished ones not handed out yet
        self.clamp_window.on_event(e.status()[0][:, 5])
        if e.features_split and inf.supports_split_features(e.N, self.device):
            inf.act_calibrated = False
            inf.set_act_shift(min(inf.MAX_ACT_SHIFT, old_shift + 2))  # at least 4x more room, then whatever the calibration asks for
            inf.calibrate_activation_scale(e.features, split_features=(e.rows, e.N), slot=0)
            self.range_rescales += 1
            self._note_evaluator_path()
            what = (f"falls back to the library's fp32 convolutions ({inf.split_fallback_reason})" if inf.split_fallback_reason
                    else f"activation scale raised 2^-{old_shift} -> 2^-{inf.act_shift}")
        else:
            # the split tower behind a library stem / heads (shapes without split stem or head kernels): no layer-by-layer calibration
            # pass exists for it -- raise the scale by what the record shows (a lower bound: clamped values hide the true maximum) + 16x
            import math

            k = min(inf.MAX_ACT_SHIFT, old_shift + max(2, math.ceil(math.log2(max(mx, 65504.0) / 65504.0)) + 4))
            if k > old_shift:
                inf.set_act_shift(k)
                self._graph = None  # (this path multiplies by 2^-k with a host constant: a captured graph holds the old one)
                self.range_rescales += 1
                what = f"activation scale raised 2^-{old_shift} -> 2^-{inf.act_shift}"
            else:
                what = "the scale is at its limit: evaluate this network with use_split_tower = False"
        warnings.warn(f"alpha_zero_amd: the fp32-class evaluator clamped {ev} activation lanes beyond f16's range (largest |v| = "
                      f"{mx * 2.0 ** old_shift:.6g}); the reference's fp32 network would have carried them -- {what}",
                      RuntimeWarning, stacklevel=3)

    def harvest(self, with_moves=False):
        """Finished games as the reference actor emits them: [(game_seq: list[Transition], stats: dict)]
        (pipeline.py:283, :356-380).  pi_prob is float64 for Go and float32 for Gomoku like the reference.
        with_moves=True yields (game_seq, stats, moves): the game's move list as env.history holds it (flat actions, N*N = pass;
        a final resignation is not a history move, base.py:224-226) -- what to_sgf() needs (pipeline.py:276-281)."""
        got = self.engine.harvest(with_moves=with_moves)
        self._check_evaluator_range()
        states, pi, z, games = got[:4]
        extra = self.engine.last_extra
        if len(games) == 0:
            self.last_harvest_clamped = np.zeros(0, dtype=bool)
            return []
        states, pi, z = states.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
        moves = got[4].cpu().numpy() if with_moves else None
        out = []
        self.last_harvest_clamped = self.clamp_window.mask(games[:, 11])
        for row, ex, clamped in zip(games, extra, self.last_harvest_clamped):
            s0, ln = int(row[0]), int(row[1])
            if clamped:
                # the game ran a round between a clamp event of the fp32-class evaluator and its repair: some of its searches may have
                # used a clamped activation where the reference's fp32 forward (pipeline.py:102-109) carries the value on
                self.clamped_games += 1
                if self.drop_clamped_games:
                    continue
            if int(ex[3]):
                # The game was in progress across a weight hot-swap -- impossible in the reference, whose actor only reloads between
                # games (pipeline.py:232-239).  It keeps the tag of the weights that STARTED it (pipeline.py:237 -> :271) and is
                # counted; drop_straddling_games=True discards it instead.
                self.straddled_games += 1
                if self.drop_straddling_games:
                    continue
            pis = pi[s0:s0 + ln].astype(np.float64) if self.game == "go" else pi[s0:s0 + ln]
            seq = [Transition(state=states[s0 + i].copy(), pi_prob=pis[i].copy(), value=float(z[s0 + i])) for i in range(ln)]
            # the threshold this very game was played with (pipeline.py:241-242, :379), exact double
            thr = float(np.array([ex[1], ex[2]], dtype=np.int32).view(np.float64)[0])
            stats = game_stats_from_row(row, self.game, self.komi, thr)
            stats["training_steps"] = int(row[12])  # weights in use when the game started (pipeline.py:237, :271, :492)
            if clamped:
                stats["evaluator_clamped"] = True  # (only ever present on such a game: the reference's stats keys stay as they are)
            out.append((seq, stats, [int(m) for m in moves[s0:s0 + ln] if m >= 0]) if with_moves else (seq, stats))
        return out

    def game_sgf(self, stats, moves, date=""):
        """SGF text of a harvested game in the reference's format (go.py:202-210 / gomoku.py:149-157 through sgf_wrapper.make_sgf)."""
        from ..envs.base import PlayerMove
        from ..utils.sgf import make_sgf

        n = self.cfg.board_size
        hist = [PlayerMove("B" if i % 2 == 0 else "W", m) for i, m in enumerate(moves)]
        is_go = self.game == "go"
        return make_sgf(n, hist, stats["game_result"], ruleset="Chinese" if is_go else "", komi=self.komi if is_go else "", date=date)

    def counters(self, reset=False):
        return self.engine.counters(reset)


def load_checkpoint_state(path, allow_pickle=None):
    """Checkpoint dictionary of the learner (pipeline.py:597-606: 'network', 'training_steps', optimizer / scheduler state) from `path`.
    Loaded with weights_only=True: tensors, numbers, strings, containers -- no pickled code runs.  Only when the safe loader REJECTS the
    file's content (pickle.UnpicklingError: an object type outside its allow-list, e.g. a pickled scheduler object) and the caller opted
    in (allow_pickle=True, or the environment variable AZSP_ALLOW_PICKLE_CKPT=1) is the file re-read with the full unpickler, and that
    is logged.  I/O errors and corrupt files propagate; they are never retried with the unsafe loader."""
    import logging
    import os
    import pickle

    import collections

    try:  # MultiStepLR.state_dict() (training_go.py:273) holds its milestones as a collections.Counter: plain data, allow-listed
        with torch.serialization.safe_globals([collections.Counter, collections.OrderedDict]):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if allow_pickle is None:
            allow_pickle = os.environ.get("AZSP_ALLOW_PICKLE_CKPT", "0") == "1"
        if not allow_pickle:
            raise
        logging.getLogger("alpha_zero_amd").warning("checkpoint %s needs the full unpickler (%s): loading with weights_only=False", path, e)
        return torch.load(path, map_location="cpu", weights_only=False)


def run_selfplay_actor_loop(seed, rank, network, device, data_queue, env, num_simulations, num_parallel, c_puct_base, c_puct_init,
                            warm_up_steps, check_resign_after_steps, disable_resign_ratio, save_sgf_dir=None, save_sgf_interval=0,
                            logs_dir=None, load_ckpt=None, log_level="INFO", var_ckpt=None, var_resign_threshold=None,
                            ckpt_event=None, stop_event=None, num_games=4096, net_dtype=torch.float32, harvest_every=64, binding=None):
    """Same role and argument list as the reference actor entry point (pipeline.py:166-189), extended by
    `num_games`: one call drives `num_games` games on `device` and puts (game_seq, stats) tuples on
    `data_queue` exactly as `num_games` reference actors would.  `net_dtype` defaults to the reference's evaluator precision (fp32,
    see SelfPlayActor)."""
    import os

    game = "go" if env.has_pass_move else "gomoku"
    training_steps = 0
    if load_ckpt is not None and os.path.exists(load_ckpt):  # pipeline.py:208-212
        st = load_checkpoint_state(load_ckpt)
        network.load_state_dict(st["network"])
        training_steps = st["training_steps"]
    thr = var_resign_threshold.value if (var_resign_threshold is not None and env.has_resign_move) else -1.0
    actor = SelfPlayActor(network, game=game, board_size=env.board_size, num_games=num_games, num_simulations=num_simulations,
                          num_parallel=num_parallel, c_puct_base=c_puct_base, c_puct_init=c_puct_init, warm_up_steps=warm_up_steps,
                          check_resign_after_steps=check_resign_after_steps, disable_resign_ratio=disable_resign_ratio,
                          resign_threshold=thr, komi=getattr(env, "komi", 7.5), num_to_win=getattr(env, "num_to_win", 5),
                          seed=seed, rank=rank, device=device, net_dtype=net_dtype, training_steps=training_steps, binding=binding)
    actor.drop_clamped_games = True  # the learner never sees a game that ran on a clamping evaluator (the reference's fp32 has no range)
    writer = None
    if logs_dir:  # per-actor statistics file with the reference's columns (pipeline.py:196, :268-271; logs/go/9x9/actor0.csv)
        from ..utils.csv_writer import CsvWriter
        from ..utils.sgf import get_time_stamp

        writer = CsvWriter(os.path.join(logs_dir, f"actor{rank}.csv"))
    last_ckpt, t_last, played_games = None, time.time(), 0
    while stop_event is None or not stop_event.is_set():
        if ckpt_event is not None and ckpt_event.is_set():  # the learner is writing a checkpoint (pipeline.py:228-230)
            time.sleep(0.001)
            continue
        if var_resign_threshold is not None and env.has_resign_move and var_resign_threshold.value != actor.resign_threshold:
            actor.set_resign_threshold(var_resign_threshold.value)  # pipeline.py:241-242: read before every game
        if var_ckpt is not None:
            new_ckpt = var_ckpt.value.decode("utf-8") if isinstance(var_ckpt.value, bytes) else str(var_ckpt.value)
            if new_ckpt != "" and new_ckpt != last_ckpt and os.path.exists(new_ckpt):  # pipeline.py:232-239
                st = load_checkpoint_state(new_ckpt)
                network.load_state_dict(st["network"])
                actor.set_network(network, st["training_steps"])
                last_ckpt = new_ckpt
        # Rounds in small chunks, polling the checkpoint event between them.  The reference actor queues every game that finished before
        # the event, drops the one game that ends while it is set (pipeline.py:264-267) and starts no new game until it clears
        # (pipeline.py:228-230).  Here: the moment the event is seen, whatever has finished so far is harvested and queued (those games
        # ended before it was noticed), then the engine pauses at the top of the loop -- no game ends during the event, none is dropped;
        # the games in progress resume under the new weights and are counted as straddling (SelfPlayActor.straddled_games).
        done_rounds, poll = 0, max(1, min(8, harvest_every))
        while done_rounds < harvest_every and not (ckpt_event is not None and ckpt_event.is_set()) and not (stop_event is not None and stop_event.is_set()):
            actor.run_rounds(min(poll, harvest_every - done_rounds))
            done_rounds += poll
            actor.poll_evaluator_range()  # two words: a clamp event is repaired within `poll` rounds, the games of the window are remembered
        want_sgf = bool(save_sgf_dir) and save_sgf_interval > 0 and os.path.isdir(save_sgf_dir)
        finished = actor.harvest(with_moves=want_sgf)
        now = time.time()
        if stop_event is not None and stop_event.is_set():
            break
        if want_sgf:  # every save_sgf_interval-th finished game is dumped like the reference does (pipeline.py:276-281)
            from ..utils.sgf import get_time_stamp as _ts

            for seq, stats, moves in finished:
                played_games += 1
                if played_games % save_sgf_interval == 0:
                    with open(os.path.join(save_sgf_dir, f"actor{rank}_{_ts(True)}_{played_games}.sgf"), "w") as f:
                        f.write(actor.game_sgf(stats, moves, date=_ts()))
            finished = [(seq, stats) for seq, stats, _ in finished]
        for seq, stats in finished:
            ts = stats.pop("training_steps")
            stats["time_per_game"] = round((now - t_last) * num_games / max(1, len(finished)), 4)
            stats["training_steps"] = ts  # key order of the reference's stats / CSV columns (pipeline.py:268-271)
            if writer is not None:
                writer.write({"datetime": get_time_stamp(), **stats})
            data_queue.put((seq, stats))
        if finished:
            t_last = now
    if writer is not None:
        writer.close()
