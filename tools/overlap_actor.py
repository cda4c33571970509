"""EXPERIMENT (tools only, not part of the product): SelfPlayActor with the games run as two half-batches on two HIP streams, so that
the select / backup kernels of one half could execute while the other half's leaf batch is inside the evaluator (VERDICT r2 "Next" #2).

Measured on MI355X in round 3 (profiles/r03_overlap_*.json) and NOT adopted:
  * no gain: 9x9 Go / 10x128 17 573 vs 17 483 moves/s (+0.5 %), 13x13 Gomoku / 6x64 25 374 vs 25 307 (+0.3 %).  The weight-stationary
    convolution kernels hold one persistent workgroup per CU that owns all 512 registers of every SIMD and 160 KB of LDS, so engine
    waves only get CUs in the tails between launches, and the halved launches pay their prologue twice;
  * not bit-exact: with two forwards in flight at once the head planes of the forward that was launched FIRST differ in a few
    neighbouring positions per round (tools/concurrency_probe2.py: its stem and tower outputs are identical to a serial run, k_head_tiled's
    output is not; the second forward, whose head kernel runs alone, is exact).  The same half-batch forwards one after the other on
    ONE stream are bit-identical to the whole batch (tools/overlap_debug.py seq_fwd / seq_rev, 250 rounds), engine kernels of disjoint
    ranges on two streams are exact, engine kernels beside a forward are exact.  The difference disappears when k_head_tiled cannot
    share a CU with another workgroup (96 KB of extra dynamic LDS in a probe build); an agent-scope acquire fence did not change it and
    LDS-DMA does not land in a neighbour's allocation (tools/probes/lds_dma_coresidency_probe.hip).  Mechanism open; the product runs
    one stream per engine.
The range launches this experiment needs (azsp_select_range / azsp_expand_backup_range) ARE product API: they are exact
(tests/test_actor_host.py, tests/test_engine_gpu.py run disjoint game ranges one after the other on one stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402


class OverlapActor(SelfPlayActor):
    def __init__(self, network, *a, overlap=True, one_stream=False, **kw):
        kw.setdefault("net_dtype", torch.bfloat16)  # the experiment is about the tiled bf16 evaluator (SelfPlayActor's own default is fp32)
        super().__init__(network, *a, **kw)
        self._halves, self._streams, self._hgraphs, self._forked = None, None, [None, None], False
        self.overlap = False
        if overlap:
            gA = self._plan_halves()
            self._halves = [(0, gA), (gA, self.engine.G)]
            s0 = torch.cuda.Stream(self.device)
            self._streams = [s0, s0 if one_stream else torch.cuda.Stream(self.device)]
            self.engine.on_launch = self._on_engine_launch
            self.overlap = True

    def _plan_halves(self):
        """First game of the second half: a multiple of 32 (sub-range launches keep every game on its XCD), its first leaf row on a
        feature-tile boundary, chosen so that the two halves' persistent convolution grids need no more tile-times per CU than the
        undivided batch (4096 games x 8 leaves at 9x9: 5376 = 21 x 256 tiles + 5547 <= 22 x 256 tiles, against 43 x 256 undivided)."""
        e = self.engine
        tb = max(1, 256 // (e.N * e.N))
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        best = None
        for gA in range(32, e.G, 32):
            if (gA * e.P) % tb:
                continue
            tA, tB = gA * e.P // tb, -(-((e.G - gA) * e.P) // tb)
            key = (-(-tA // n_cu) + -(-tB // n_cu), abs(2 * gA - e.G))
            if best is None or key < best[0]:
                best = (key, gA)
        return best[1]

    def _on_engine_launch(self, cur):
        """Engine.on_launch: anything that reaches the engine from another stream first waits for the two half-batch streams."""
        if self._forked and all(cur.cuda_stream != s.cuda_stream for s in self._streams):
            for s in self._streams:
                cur.wait_stream(s)
            self._forked = False

    def _forward_half(self, k):
        e = self.engine
        g0, g1 = self._halves[k]
        r0, r1 = g0 * e.P, g1 * e.P
        tb = max(1, 256 // (e.N * e.N))
        feat = e.features[(r0 // tb) * (32 * tb * e.N * e.N):]  # [tile][4 chunks][tb N^2 positions][8 channels]
        self.infer.forward_tiled(feat, r1 - r0, e.N, e.priors[r0:r1], e.values[r0:r1], slot=1 + k)

    def _capture_half(self, k):
        for _ in range(2):
            self._forward_half(k)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self._streams[k]):
            self._forward_half(k)
        self._hgraphs[k] = g

    def set_network(self, network, training_steps=0):
        if getattr(self, "_streams", None) is not None:
            self._on_engine_launch(torch.cuda.current_stream(self.device))
            torch.cuda.synchronize(self.device)
        super().set_network(network, training_steps)
        self._hgraphs = [None, None]

    def run_round(self, evs=None):
        if not self.overlap:
            return super().run_round(evs)
        e = self.engine
        if not self._forked:
            main = torch.cuda.current_stream(self.device)
            for s in self._streams:
                s.wait_stream(main)
            self._forked = True
        for k, (g0, g1) in enumerate(self._halves):
            with torch.cuda.stream(self._streams[k]):
                e.expand_backup(g0, g1)
                e.select(g0, g1)
                if self.use_graph:
                    if self._hgraphs[k] is None:
                        self._capture_half(k)
                    self._hgraphs[k].replay()
                else:
                    self._forward_half(k)
        self.rounds += 1
