"""Statistical checks of the PRODUCTION randomness (no injection): the Philox streams behind add_dirichlet_noise and
np.random.choice ("handled by injection (parity) + statistical tests (production)")."""
import ctypes as C

import numpy as np
from scipy import stats

import engine_util as eu
from alpha_zero_amd.core.engine import Engine, EngineConfig


def _probe(kind, G, plies, tries, seed, rank=0, n=9):
    binding, dev = eu.backend(kind)
    eng = Engine(binding, EngineConfig(game="go", board_size=n, num_games=G, num_parallel=2, num_simulations=8, seed=seed, rank=rank), device=dev)
    eng.reset_games()
    A = eng.A
    noise = np.zeros((G, plies, A))
    unif = np.zeros((G, plies, tries))
    rc = binding.dll.azsp_rng_probe(eng.h, plies, tries, noise.ctypes.data_as(C.c_void_p), unif.ctypes.data_as(C.c_void_p), None)
    assert rc == 0
    eng.close()
    return noise, unif


def check_production_rng(kind):
    G, plies, tries, alpha = 64, 40, 8, 0.03
    noise, unif = _probe(kind, G, plies, tries, seed=1234)
    A = noise.shape[-1]
    # --- Dirichlet(alpha * 1_A): simplex, moments, marginal law -------------------------------------------------------
    assert np.all(noise >= 0) and np.allclose(noise.sum(-1), 1.0, atol=1e-12)
    flat = noise.reshape(-1, A)  # 2560 independent vectors
    a0 = alpha * A
    mean, var = 1.0 / A, (alpha * (a0 - alpha)) / (a0 * a0 * (a0 + 1.0))
    se = np.sqrt(var / flat.shape[0])
    assert np.all(np.abs(flat.mean(0) - mean) < 6 * se), np.abs(flat.mean(0) - mean).max() / se
    assert abs(flat.var(0).mean() / var - 1.0) < 0.05
    # component marginal is Beta(alpha, a0 - alpha); a handful of actions incl. the pass column, Kolmogorov-Smirnov on each
    for a in (0, 1, 40, A - 2, A - 1):
        p = stats.kstest(flat[:, a], stats.beta(alpha, a0 - alpha).cdf).pvalue
        assert p > 1e-4, (a, p)
    # a joint statistic against NumPy's own Dirichlet sampler: the mean of the largest component (0.441 for alpha 0.03, A 82)
    ref = np.random.Generator(np.random.PCG64(1)).dirichlet(np.full(A, alpha), size=20000).max(1)
    assert abs(flat.max(1).mean() - ref.mean()) < 6 * ref.std() * np.sqrt(1.0 / flat.shape[0] + 1.0 / ref.size)
    # distinct across slots / plies, reproducible, and keyed by seed and rank
    assert len({v.tobytes() for v in flat}) == flat.shape[0]
    again, u2 = _probe(kind, G, plies, tries, seed=1234)
    assert np.array_equal(again, noise) and np.array_equal(u2, unif)
    other, _ = _probe(kind, 8, 2, 1, seed=1235)
    ranked, _ = _probe(kind, 8, 2, 1, seed=1234, rank=1)
    assert not np.array_equal(other, noise[:8, :2]) and not np.array_equal(ranked, noise[:8, :2])
    assert np.array_equal(ranked, other)  # the key is seed + rank (pipeline.py:193)
    # --- uniforms behind np.random.choice ----------------------------------------------------------------------------
    u = unif.reshape(-1)
    assert np.all((u >= 0) & (u < 1)) and len(np.unique(u)) == u.size
    assert stats.kstest(u, "uniform").pvalue > 1e-4
    assert abs(np.corrcoef(unif[:, :, 0].reshape(-1)[:-1], unif[:, :, 0].reshape(-1)[1:])[0, 1]) < 0.05
    # the sampler built on them: searchsorted(cumsum(pi) / sum, u, 'right') reproduces pi (chi-square)
    pi = np.random.Generator(np.random.PCG64(5)).dirichlet(np.full(10, 0.7))
    cdf = np.cumsum(pi)
    picks = np.searchsorted(cdf / cdf[-1], u, side="right")
    chi = stats.chisquare(np.bincount(picks, minlength=10), pi * u.size)
    assert chi.pvalue > 1e-4, chi
