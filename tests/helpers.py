"""Shared test helpers: drive any backend (oracle or HIP) with the reference's driver logic."""
import numpy as np

from harmony_amd import harmony_options, prepare_setup_args
from harmony_amd.utils import harmonize


def run_backend(obj, data_mat, meta, vars_use, max_iter=10, Y0=None, **kw):
    """RunHarmony.default's call sequence (R/ui.R:269-283) on an already-constructed object."""
    skw, _ = prepare_setup_args(data_mat, meta, vars_use, **kw)
    obj.setup(**skw)
    obj.init_cluster_cpp(Y0) if Y0 is not None else obj.init_cluster_cpp()
    harmonize(obj, max_iter, verbose=False)
    return obj


def chi2(obj):
    O, E = obj.O, obj.E
    return float(np.sum((O - E) ** 2 / E))


def synth(N, d=50, n_types=30, levels=(10,), seed=0, nested=False):
    """Synthetic embedding in the spirit of SURVEY.md 8(d): cell types x covariate shifts."""
    rng = np.random.Generator(np.random.Philox(seed))
    s = 10.0 / np.sqrt(np.arange(1, d + 1))
    mu = rng.normal(size=(n_types, d)) * s
    covs, Z = [], None
    first = None
    parent = None
    for ci, L in enumerate(levels):
        if nested and parent is not None:
            par_of = np.sort(rng.integers(0, len(np.unique(parent)), size=L))
            par_of[: len(np.unique(parent))] = np.arange(len(np.unique(parent)))
            par_of = np.sort(par_of)
            lab = np.empty(N, dtype=np.int64)
            for p in np.unique(parent):
                idx = np.where(parent == p)[0]
                kids = np.where(par_of == p)[0]
                lab[idx] = rng.choice(kids, size=idx.size)
        else:
            w = rng.dirichlet(5 * np.ones(L))
            lab = rng.choice(L, size=N, p=w)
        covs.append(lab)
        parent = lab
        if first is None:
            first = lab
    mix = rng.dirichlet(np.ones(n_types), size=levels[0])
    types = np.array([rng.choice(n_types, p=mix[b]) for b in first])
    Z = mu[types] + rng.normal(size=(N, d)) * (0.5 * s)
    for ci, L in enumerate(levels):
        delta = rng.normal(size=(L, d)) * (0.3 * s)
        eps = rng.normal(size=(n_types, L, d)) * (0.1 * s)
        Z += delta[covs[ci]] + eps[types, covs[ci]]
    meta = {"cov%d" % i: c for i, c in enumerate(covs)}
    return Z, meta, types
