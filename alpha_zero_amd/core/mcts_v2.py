"""Drop-in `uct_search` / `parallel_uct_search` (reference: alpha_zero/core/mcts_v2.py:301-657) on the GPU engine.

Same signatures, same return tuple `(move, search_pi, root_Q, best_child_Q, next_root_node)`, same
errors, same use of the global NumPy random state (one `np.random.dirichlet` per call when
`root_noise`, then `np.random.choice` draws until the move is acceptable), so an actor or evaluator
written against the reference runs unchanged.  The tree lives in HBM inside a one-game engine
(stop_after_move mode of the C ABI); `eval_func` is the caller's Python callback and receives the
observation planes the select kernel produced, exactly the arrays the reference would pass it.
`next_root_node` is an opaque handle (callers only pass it back).
"""
import weakref

import numpy as np

from .. import _abi
from ..envs.base import BoardGameEnv
from .engine import Engine, EngineConfig

_POOL = {}  # config key -> idle _Searcher objects


class Node:
    """Opaque handle of a re-usable sub-tree (the reference returns its Node object here)."""

    def __init__(self, searcher, board, to_play, steps):
        self._searcher, self._board, self.to_play, self._steps = searcher, board, to_play, steps
        self._alive = True
        # Returns the engine to the pool when a LIVE handle is dropped (the caller abandoned the tree).  A handle that is handed
        # back to the next search passes its engine on to the new handle: its finalizer is detached there, otherwise dropping the
        # old handle would put an engine that is still in use into the pool (and a second search would overwrite the live tree).
        self._fin = weakref.finalize(self, _release, searcher)

    @property
    def is_expanded(self):
        return True


def _release(searcher):
    idle = _POOL.setdefault(searcher.key, [])
    if not any(x is searcher for x in idle):  # never pool the same engine twice
        idle.append(searcher)


class _Searcher:
    def __init__(self, key, env, sims, P, base, init, root_noise):
        self.key = key
        binding, device = getattr(env, "_binding", None), getattr(env, "_device", None)
        if binding is None:
            from .. import _lib

            binding, device = _lib.load(require_gpu=True), "cuda"
        game = "go" if env.has_pass_move else "gomoku"
        cfg = EngineConfig(game=game, board_size=env.board_size, num_games=1, num_parallel=P, num_simulations=sims, c_puct_base=base,
                           c_puct_init=init, root_noise=root_noise, komi=getattr(env, "komi", 7.5),
                           max_steps=getattr(env, "max_steps", 0) or 0, num_to_win=getattr(env, "num_to_win", 5),
                           stop_after_move=True, feature_dtype=_abi.FEAT_I8, log_moves=False)
        self.eng = Engine(binding, cfg, device=device)
        self.P = P


def _get_searcher(env, sims, P, base, init, root_noise):
    key = ("go" if env.has_pass_move else "gomoku", env.board_size, sims, P, float(base), float(init), bool(root_noise),
           getattr(env, "komi", None), getattr(env, "max_steps", None), getattr(env, "num_to_win", None), id(getattr(env, "_binding", None)))
    idle = _POOL.get(key)
    if idle:
        return idle.pop()
    return _Searcher(key, env, sims, P, base, init, root_noise)


def _load_position(s, env):
    hist = np.stack([np.asarray(b, dtype=np.int8) for b in env.board_deltas])
    ko, caps = getattr(env, "ko", -1), getattr(env, "_caps", (0, 0))
    pos = getattr(env, "position", None)  # a reference GoEnv keeps these on its Position
    if pos is not None:
        ko = -1 if pos.ko is None else pos.ko[0] * env.board_size + pos.ko[1]
        caps = tuple(pos.caps)
    last_pass = bool(env.has_pass_move and len(env.history) > 0 and env.history[-1].move == env.pass_move)
    s.eng.set_state(0, np.asarray(env.board, dtype=np.int8), hist, env.to_play, env.steps, ko, last_pass, caps)


def _simulate_with_callback(eng, eval_func, P, num_parallel, A):
    """The simulation loop (mcts_v2.py:378-421 / :568-625) with the caller's HOST callback.  One host round trip per leaf batch:
    azsp_dropin_step uploads eval_func's outputs, runs expand / backup + the selection of the next leaves, and returns status, valid
    flags and the leaves' observation planes from one packed read-back (rounds 1-5 paid eight stream synchronisations per simulation here)."""
    pri = np.zeros((eng.rows, A), dtype=np.float32)
    val = np.zeros(eng.rows, dtype=np.float32)
    st, q, valid, obs = eng.dropin_step(None, None, P)
    for _ in range(1 << 20):
        if st[0, 0] == _abi.ST_MOVE_DONE:
            return
        if valid.any():
            if st[0, 6] or num_parallel == 1:  # root evaluation / uct_search leaves: unbatched call (mcts_v2.py:365, :414, :555)
                p, v = eval_func(obs[0], False)
                pri[0], val[0] = np.asarray(p, dtype=np.float32), v
            else:
                rows = np.flatnonzero(valid)
                ps, vs = eval_func(obs[rows], True)  # mcts_v2.py:614
                for r, p, v in zip(rows, ps, vs):
                    pri[r], val[r] = np.asarray(p, dtype=np.float32), v
        st, q, valid, obs = eng.dropin_step(pri, val, P)
    raise RuntimeError("the search did not finish")


def _simulate_on_device(eng, evaluator, num_simulations, num_parallel):
    """The same loop for an evaluator that lives on the engine's device (anything with `device_eval(x) -> (priors, values)` on device
    tensors, e.g. core/evaluate.py DeviceEvaluator): select writes the leaves' observation planes into `eng.features`, the evaluator reads
    them there and its outputs go straight into `eng.priors` / `eng.values`, expand / backup reads those -- NO host round trip per
    simulation.  The host only polls the status words: a search that still owes `left` simulations cannot finish in fewer than
    left / (simulations one iteration can complete) iterations (1 for uct_search -- a terminal leaf makes select descend again in the
    same launch, mcts_v2.py:378-411 --, at most 2P attempts for parallel_uct_search, :572), so that many iterations are queued without
    looking (half of them: a search with terminal leaves needs fewer, and iterations queued after the search is done would be no-ops in the engine -- nothing pending, status
    MOVE_DONE -- but each still costs a forward).  ~8 polls per 100-simulation move instead of 100 round trips.  Every row is evaluated,
    valid or not, like the batched actor does: the engine ignores outputs of rows it did not ask for.  Same searches as the callback
    loop with the same evaluator (tests/dropin_checks.py)."""
    per_iter = 1 if num_parallel == 1 else 2 * num_parallel
    into = getattr(evaluator, "device_eval_into", None)

    def iterate(n):
        for _ in range(n):
            if into is not None:
                into(eng.features, eng.priors, eng.values)
            else:
                pri, val = evaluator.device_eval(eng.features)
                eng.priors.copy_(pri.reshape(eng.priors.shape))
                eng.values.copy_(val.reshape(eng.values.shape))
            eng.round()

    eng.round()  # nothing to back up yet: selects the first leaf (or asks for the root's evaluation)
    for _ in range(1 << 20):
        st, _ = eng.status()
        if st[0, 0] == _abi.ST_MOVE_DONE:
            return
        left = max(num_simulations - int(st[0, 2]), 1)
        iterate(max(1, (left // 2) // per_iter))
    raise RuntimeError("the search did not finish")


def _search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations, num_parallel, root_noise, warm_up, deterministic):
    if not isinstance(env, BoardGameEnv) and not (hasattr(env, "board_deltas") and hasattr(env, "legal_actions")):
        raise ValueError(f"Expect `env` to be a valid BoardGameEnv instance, got {env}")
    if not 1 <= num_simulations:
        raise ValueError(f"Expect `num_simulations` to a positive integer, got {num_simulations}")
    if env.is_game_over():
        raise RuntimeError("Game is over.")
    if root_node is not None:
        if not isinstance(root_node, Node) or not root_node._alive:
            raise ValueError("`root_node` must be the handle returned by the previous search (or None)")
        if root_node.to_play != env.to_play or root_node._steps != env.steps or not np.array_equal(root_node._board, env.board):
            raise ValueError("`root_node` does not belong to this position")
        s = root_node._searcher
        root_node._alive = False
        root_node._fin.detach()  # the engine moves on with this search; only the NEXT handle (or game end) releases it
    else:
        s = _get_searcher(env, num_simulations, num_parallel, c_puct_base, c_puct_init, root_noise)
        _load_position(s, env)
    eng, A = s.eng, s.eng.A
    root_legal = env.legal_actions
    noise = None
    if root_noise:  # add_dirichlet_noise (mcts_v2.py:259-260): the engine applies the legal mask itself
        noise = np.random.dirichlet(np.ones_like(root_legal) * 0.03)
    eng.begin_move(noise, warm_up=1 if warm_up else 0)
    if getattr(eval_func, "device_eval", None) is not None and not (eng.features_tiled or eng.features_split):
        # a device-resident evaluator (core/evaluate.py DeviceEvaluator): the leaves never visit the host -- see _simulate_on_device
        _simulate_on_device(eng, eval_func, num_simulations, num_parallel)
    else:
        _simulate_with_callback(eng, eval_func, s.P, num_parallel, A)
    pi64, child_n, _ = eng.get_search(0, 0)
    search_pi = pi64 if env.has_pass_move else pi64.astype(np.float32)  # float64 for Go, float32 for Gomoku
    move = None
    if deterministic:
        move = np.argmax(child_n)
    else:
        while move is None or (warm_up and env.has_pass_move and move == env.pass_move) or root_legal[move] != 1:
            move = np.random.choice(np.arange(search_pi.shape[0]), p=search_pi)
    eng.commit_move([int(move)])
    st, q = eng.status()
    next_root = None
    if st[0, 0] == _abi.ST_SEARCH:
        out = eng.env_step(None)  # export the new root position for the hand-over check of the next call
        next_root = Node(s, out["board"][0].copy(), int(out["scalars"][0][4]), int(out["scalars"][0][3]))
    else:
        _release(s)
    assert root_legal[move] == 1
    return (move, search_pi, float(q[0, 0]), float(q[0, 1]), next_root)


def uct_search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations=800, root_noise=False, warm_up=False,
               deterministic=False):
    """mcts_v2.py:301-450"""
    return _search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations, 1, root_noise, warm_up, deterministic)


def parallel_uct_search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations, num_parallel, root_noise=False,
                        warm_up=False, deterministic=False):
    """mcts_v2.py:485-657"""
    return _search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations, num_parallel, root_noise, warm_up,
                   deterministic)
