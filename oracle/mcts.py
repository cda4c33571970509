"""NumPy restatement of the reference PUCT search (alpha_zero/core/mcts_v2.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It is the checker for the
HIP engine and the CPU baseline of bench.py, never part of the product path.

The statistics live in one growable structure-of-arrays pool per search tree
(rows of float32 N/W/P, an int32 child table) instead of one Python object per
node, but every arithmetic expression keeps the operand dtypes and the
evaluation order of the reference so that results are bit-identical under
NumPy 2.x promotion rules ("reference under numpy 2.2.6"):

  * child_N/child_W/child_P are float32 rows                          (mcts_v2.py:90-92)
  * the root's own N and W live outside the rows: Python floats for a
    freshly created root, np.float32 scalars after a re-root          (mcts_v2.py:56-62, :439-443)
  * root priors become float64 once Dirichlet noise is mixed in       (mcts_v2.py:259-262)
  * pb_c is a Python double computed from the node's own visit count  (mcts_v2.py:99-102)

Parity is pinned by tests/test_oracle_mcts.py against golden vectors produced
by the reference itself (tools/gen_golden_mcts.py).
"""
import copy
import math

import numpy as np


class Rand:
    """Randomness source.  Default = the reference's global np.random calls
    (mcts_v2.py:260 dirichlet, :434/:641 choice).  Tests inject recorded values."""

    def dirichlet(self, alphas):
        return np.random.dirichlet(alphas)

    def uniform(self):
        return np.random.random_sample()


class InjectedRand(Rand):
    def __init__(self, noise, uniforms):
        self.noise, self.uniforms, self.k = noise, list(uniforms), 0

    def dirichlet(self, alphas):
        return np.asarray(self.noise, dtype=np.float64)

    def uniform(self):
        u = self.uniforms[self.k]
        self.k += 1
        return u


class Tree:
    """Pool of nodes for one game.  Node 0.. ; `root` is the current root index."""

    def __init__(self, num_actions, to_play):
        self.A = num_actions
        self.N = []        # per node: float32[A]   (children visit counts)
        self.W = []        # per node: float32[A]
        self.P = []        # per node: float32[A] (float64 at a noisy root)
        self.child = []    # per node: dict move -> node index (lazy creation, mcts_v2.py:182-183)
        self.parent = []   # node index or -1
        self.move = []     # action that led here (None for the root)
        self.to_play = []
        self.expanded = []
        self.vloss = []
        self.root = self._new(-1, None, to_play)
        # DummyNode slots (mcts_v2.py:56-62): defaultdict(float) -> Python floats
        self.root_N = 0.0
        self.root_W = 0.0

    def _new(self, parent, move, to_play):
        self.N.append(np.zeros(self.A, dtype=np.float32))
        self.W.append(np.zeros(self.A, dtype=np.float32))
        self.P.append(np.zeros(self.A, dtype=np.float32))
        self.child.append({})
        self.parent.append(parent)
        self.move.append(move)
        self.to_play.append(to_play)
        self.expanded.append(False)
        self.vloss.append(0)
        return len(self.N) - 1

    # a node's own statistics are stored in its parent's rows (mcts_v2.py:111-135)
    def get_N(self, i):
        return self.root_N if i == self.root else self.N[self.parent[i]][self.move[i]]

    def get_W(self, i):
        return self.root_W if i == self.root else self.W[self.parent[i]][self.move[i]]

    def add_N(self, i, d):
        if i == self.root:
            self.root_N = self.root_N + d
        else:
            self.N[self.parent[i]][self.move[i]] += d

    def add_W(self, i, d):
        if i == self.root:
            self.root_W = self.root_W + d
        else:
            self.W[self.parent[i]][self.move[i]] += d

    def Q(self, i):
        n = self.get_N(i)
        return self.get_W(i) / n if n > 0 else 0.0

    def size(self):
        return len(self.N)


def _scores(t, i, c_puct_base, c_puct_init):
    """-Q + U for every action of node i (mcts_v2.py:99-109, :173)."""
    n_self = t.get_N(i)
    pb_c = math.log((1 + n_self + c_puct_base) / c_puct_base) + c_puct_init
    u = pb_c * t.P[i] * (math.sqrt(n_self) / (1 + t.N[i]))
    q = t.W[i] / np.where(t.N[i] > 0, t.N[i], 1)
    return -q + u


def _best_child(t, i, legal, c_puct_base, c_puct_init, child_to_play):
    """mcts_v2.py:142-185: masked argmax (first maximum), lazy child creation."""
    s = np.where(legal == 1, _scores(t, i, c_puct_base, c_puct_init), -9999)
    move = int(np.argmax(s))
    assert legal[move] == 1
    if move not in t.child[i]:
        t.child[i][move] = t._new(i, move, child_to_play)
    return t.child[i][move]


def _expand(t, i, prior):
    """mcts_v2.py:188-210: priors stored as given, no masking, no renormalisation."""
    if t.expanded[i]:
        raise RuntimeError("Node already expanded.")
    if not isinstance(prior, np.ndarray) or prior.ndim != 1 or prior.dtype not in (np.float32, np.float64):
        raise ValueError(f"Expect `prior_prob` to be a 1D float numpy.array, got {prior}")
    t.P[i] = prior
    t.expanded[i] = True


def _backup(t, i, value):
    """mcts_v2.py:213-232: N += 1, W += v, flip sign, up to and including the root slot."""
    if not isinstance(value, float):
        raise ValueError(f"Expect `value` to be a float type, got {type(value)}")
    while i != -1:
        t.add_N(i, 1)
        t.add_W(i, value)
        i = t.parent[i] if i != t.root else -1
        value = -1 * value


def _add_vloss(t, i):
    """mcts_v2.py:453-467: +1 on W only, every node of the path including the root slot."""
    while i != -1:
        t.vloss[i] += 1
        t.add_W(i, +1)
        i = t.parent[i] if i != t.root else -1


def _revert_vloss(t, i):
    """mcts_v2.py:470-482"""
    while i != -1:
        if t.vloss[i] > 0:
            t.vloss[i] -= 1
            t.add_W(i, -1)
        i = t.parent[i] if i != t.root else -1


def _add_noise(t, legal, rand, eps=0.25, alpha=0.03):
    """mcts_v2.py:235-262: alphas over ALL actions, noise masked but not renormalised -> float64 priors."""
    alphas = np.ones_like(legal) * alpha
    noise = legal * rand.dirichlet(alphas)
    t.P[t.root] = t.P[t.root] * (1 - eps) + noise * eps


def search_policy(child_N, temperature, legal):
    """mcts_v2.py:265-298"""
    n = legal * child_N
    if temperature > 0.0:
        n = np.power(n, max(1.0, min(5.0, 1.0 / temperature)))
    s = np.sum(n)
    if s > 0:
        n /= s
    return n


def _sample(pi, rand):
    """np.random.choice(arange(A), p=pi) -- numpy's legacy algorithm (mtrand.pyx RandomState.choice):
    cdf = cumsum(p as float64); cdf /= cdf[-1]; searchsorted(cdf, u, side='right')."""
    cdf = np.asarray(pi, dtype=np.float64).cumsum()
    cdf /= cdf[-1]
    return int(cdf.searchsorted(rand.uniform(), side="right"))


def _make_root(env, eval_func):
    """mcts_v2.py:364-368 / :554-558"""
    prior, value = eval_func(env.observation(), False)
    t = Tree(env.action_dim, env.to_play)
    _expand(t, t.root, prior)
    _backup(t, t.root, value)
    return t


def _finish(t, env, root_legal, warm_up, deterministic, rand):
    """Shared tail of both searches: mcts_v2.py:421-450 / :628-657."""
    pi = search_policy(t.N[t.root], 1.0 if warm_up else 0.1, root_legal)
    move = None
    best_child_q = 0.0
    if deterministic:
        move = int(np.argmax(t.N[t.root]))
    else:
        while move is None or (warm_up and env.has_pass_move and move == env.pass_move) or root_legal[move] != 1:
            move = _sample(pi, rand)
    root_q = t.Q(t.root)
    next_tree = None
    if move in t.child[t.root]:
        c = t.child[t.root][move]
        n, w = copy.copy(t.get_N(c)), copy.copy(t.get_W(c))  # np.float32 scalars (mcts_v2.py:439)
        t.root = c
        t.root_N, t.root_W = n, w
        best_child_q = -t.Q(c)
        next_tree = t
    assert root_legal[move] == 1
    return move, pi, root_q, best_child_q, next_tree


def _descend(t, env, c_puct_base, c_puct_init):
    """One selection from the root on a private copy of the env (mcts_v2.py:379-404 / :576-601)."""
    node = t.root
    sim_env = copy.deepcopy(env)
    obs = sim_env.observation()
    done = sim_env.is_game_over()
    reward = 0.0
    while t.expanded[node]:
        node = _best_child(t, node, sim_env.legal_actions, c_puct_base, c_puct_init, sim_env.opponent_player)
        obs, reward, done, _ = sim_env.step(t.move[node])
        if done:
            break
    assert t.to_play[node] == sim_env.to_play
    return node, obs, reward, done


def _check_args(env, num_simulations):
    if not 1 <= num_simulations:
        raise ValueError(f"Expect `num_simulations` to a positive integer, got {num_simulations}")
    if env.is_game_over():
        raise RuntimeError("Game is over.")


def uct_search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations=800, root_noise=False, warm_up=False,
               deterministic=False, rand=None):
    """mcts_v2.py:301-450.  `root_node` is a Tree (or None)."""
    rand = rand or Rand()
    _check_args(env, num_simulations)
    t = root_node if root_node is not None else _make_root(env, eval_func)
    assert t.to_play[t.root] == env.to_play
    root_legal = env.legal_actions
    if root_noise:
        _add_noise(t, root_legal, rand)
    while t.root_N < num_simulations:
        node, obs, reward, done = _descend(t, env, c_puct_base, c_puct_init)
        if done:
            _backup(t, node, -reward)  # :407-411 terminal child is never expanded
            continue
        prior, value = eval_func(obs, False)
        _expand(t, node, prior)
        _backup(t, node, value)
    return _finish(t, env, root_legal, warm_up, deterministic, rand)


def parallel_uct_search(env, eval_func, root_node, c_puct_base, c_puct_init, num_simulations, num_parallel, root_noise=False,
                        warm_up=False, deterministic=False, rand=None):
    """mcts_v2.py:485-657: leaves are gathered one after another under virtual loss, then one batched evaluation."""
    rand = rand or Rand()
    _check_args(env, num_simulations)
    t = root_node if root_node is not None else _make_root(env, eval_func)
    assert t.to_play[t.root] == env.to_play
    root_legal = env.legal_actions
    if root_noise:
        _add_noise(t, root_legal, rand)
    while t.root_N < num_simulations + num_parallel:  # :568
        leaves, failsafe = [], 0
        while len(leaves) < num_parallel and failsafe < num_parallel * 2:  # :572
            failsafe += 1
            node, obs, reward, done = _descend(t, env, c_puct_base, c_puct_init)
            if done:
                _backup(t, node, -reward)
                continue
            _add_vloss(t, node)
            leaves.append((node, obs))
        if leaves:
            nodes, obs_list = zip(*leaves)
            priors, values = eval_func(np.stack(obs_list, axis=0), True)
            for leaf, prior, value in zip(nodes, priors, values):
                _revert_vloss(t, leaf)
                if t.expanded[leaf]:  # :621-622 duplicate leaf: evaluation wasted
                    continue
                _expand(t, leaf, prior)
                _backup(t, leaf, value)
    return _finish(t, env, root_legal, warm_up, deterministic, rand)
