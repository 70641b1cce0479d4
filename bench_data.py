"""Synthetic cell x PC embeddings for bench.py and the tests (simplified).

T cell types with centres mu_t ~ N(0, diag(s_j^2)), s_j = 10 j^-1/2; every covariate level b adds a shift
delta_b ~ N(0,(0.3 s_j)^2) and a type x level interaction eps_{t,b} ~ N(0,(0.1 s_j)^2); cell noise
N(0,(0.5 s_j)^2).  Level sizes ~ Dirichlet(5), cell-type mix per level of the first covariate ~ Dirichlet(1)
(batches differ in composition); nested=True makes each child level belong to one parent level.
Global structure comes from `seed`, the cells of a shard from (seed, shard) -- any shard can be generated
independently (Philox counter-based bit generator)."""
import numpy as np


def synth(N, d=50, n_types=30, levels=(10,), seed=0, nested=False, shard=0):
    g = np.random.Generator(np.random.Philox(key=seed))                 # global structure
    rng = np.random.Generator(np.random.Philox(key=[seed, shard + 1]))  # this shard's cells
    s = 10.0 / np.sqrt(np.arange(1, d + 1))
    mu = g.normal(size=(n_types, d)) * s
    weights, parents, deltas, epss = [], [], [], []
    n_par = None
    for L in levels:
        weights.append(g.dirichlet(5 * np.ones(L)))
        if nested and n_par is not None:
            par_of = np.sort(np.concatenate([np.arange(n_par), g.integers(0, n_par, size=L - n_par)]))
            parents.append(par_of)
        else:
            parents.append(None)
        n_par = L
        deltas.append(g.normal(size=(L, d)) * (0.3 * s))
        epss.append(g.normal(size=(n_types, L, d)) * (0.1 * s))
    mix = g.dirichlet(np.ones(n_types), size=levels[0])
    covs = []
    for ci, L in enumerate(levels):
        if parents[ci] is None or ci == 0:
            lab = rng.choice(L, size=N, p=weights[ci])
        else:
            lab = np.empty(N, dtype=np.int64)
            prev = covs[ci - 1]
            for p in range(len(weights[ci - 1])):
                idx = np.where(prev == p)[0]
                kids = np.where(parents[ci] == p)[0]
                w = weights[ci][kids] / weights[ci][kids].sum()
                lab[idx] = rng.choice(kids, size=idx.size, p=w)
        covs.append(lab)
    cum = np.cumsum(mix, axis=1)
    types = (rng.random(N)[:, None] > cum[covs[0]]).sum(axis=1).clip(0, n_types - 1)
    Z = mu[types] + rng.normal(size=(N, d)) * (0.5 * s)
    for ci in range(len(levels)):
        Z += deltas[ci][covs[ci]] + epss[ci][types, covs[ci]]
    meta = {"cov%d" % i: c for i, c in enumerate(covs)}
    return Z, meta, types


def pbmc30k(n=30000, seed=0):
    """BASELINE configs[1] at its stated size: pbmc_stim, ~30k cells x 50 PCs, stim / ctrl.  The full Kang et al. data set is a download
    (vignettes/Seurat.Rmd:54-75); the repository ships a 2 000-cell sample (data/pbmc_stim.RData -> tests/golden/pbmc_stim_pcs.npz: its
    own 50 PCs).  Stand-in of the stated size: the sample's cells resampled with replacement inside each condition (the conditions keep
    their 1:1 split) plus Gaussian jitter of 10 % of every PC's spread -- the sample's cluster and batch structure at 30k cells.
    Returns (Z [n x 50] float64, {"stim": labels})."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "pbmc_stim_pcs.npz"), allow_pickle=False)
    pcs, stim = fx["pcs"].astype(np.float64), fx["stim"]
    rng = np.random.Generator(np.random.Philox(key=[seed, 30000]))
    idx = []
    levels = np.unique(stim)
    for li, lv in enumerate(levels):
        pool = np.where(stim == lv)[0]
        cnt = n // len(levels) + (1 if li < n % len(levels) else 0)
        idx.append(rng.choice(pool, size=cnt, replace=True))
    idx = rng.permutation(np.concatenate(idx))
    Z = pcs[idx] + rng.normal(size=(n, pcs.shape[1])) * (0.1 * pcs.std(axis=0))
    return Z, {"stim": fx["stim_levels"][stim[idx]]}
