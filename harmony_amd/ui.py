"""RunHarmony(): the front door, mirroring RunHarmony.default (R/ui.R:91-309).

Host-side glue only: argument validation, defaults, the one-hot design Phi, theta / lambda /
sigma vectors -- then the same five calls the reference makes on its module object
(new -> setup -> init_cluster_cpp -> harmonize -> getZcorr).  All numerics happen in
libharmony_mi355x.so through `Harmony`.
"""
import numpy as np

from .harmony_obj import Harmony
from .options import HarmonyOptions, check_legacy_args, harmony_options
from .utils import _message, harmonize


def _columns(meta_data):
    """data.frame-like -> {name: 1-D array}; accepts pandas.DataFrame or a mapping."""
    if hasattr(meta_data, "columns") and hasattr(meta_data, "__getitem__"):
        return {str(c): np.asarray(meta_data[c]) for c in meta_data.columns}
    if isinstance(meta_data, dict):
        return {str(k): np.asarray(v) for k, v in meta_data.items()}
    return None


def as_factor(values, levels=None):
    """as.factor(): sorted unique levels, 0-based codes.  `levels` (sorted) fixes the level set, e.g. the GLOBAL
    levels when the cells are sharded over several processes."""
    values = np.asarray(values)
    if levels is None:
        levels, codes = np.unique(values, return_inverse=True)
        return codes.astype(np.int32), levels
    levels = np.asarray(levels)
    codes = np.searchsorted(levels, values)
    if np.any(codes >= len(levels)) or np.any(levels[np.minimum(codes, len(levels) - 1)] != values):
        raise ValueError("value outside the given factor levels")
    return codes.astype(np.int32), levels


def build_phi(factor_codes, n_levels):
    """rbind of t(sparse.model.matrix(~0 + as.factor(x))) per covariate (R/ui.R:210-213) as CSC."""
    C = len(factor_codes)
    N = len(factor_codes[0])
    offs = np.concatenate([[0], np.cumsum(n_levels)[:-1]]).astype(np.int32)
    phi_i = np.empty(C * N, dtype=np.int32)
    for c in range(C):
        phi_i[c::C] = factor_codes[c] + offs[c]
    phi_p = (np.arange(N + 1, dtype=np.int64) * C).astype(np.int32)
    return phi_i, phi_p, None, int(np.sum(n_levels))


def prepare_setup_args(data_mat, meta_data, vars_use, theta=None, sigma=0.1, lambda_=None, nclust=None,
                       early_stop=True, verbose=False, options=None, N_b=None, levels=None):
    """Everything RunHarmony.default computes before `new(harmony)` (R/ui.R:133-258).

    Returns (setup_kwargs, data_mat d x N).  Sharded callers pass the GLOBAL level sizes `N_b` (theta scaling)
    and the GLOBAL factor levels `levels` ({variable: sorted levels}); otherwise both come from meta_data.
    """
    if options is None:
        options = harmony_options()
    if not isinstance(options, HarmonyOptions):
        raise ValueError("Error: .options must be created from harmony_options()!")
    epsilon_harmony = options["epsilon_harmony"] if early_stop else -np.inf
    alpha, tau = options["alpha"], options["tau"]

    data_mat = np.asarray(data_mat)
    cols = _columns(meta_data)
    if cols is None:  # R/ui.R:158-166
        meta = np.asarray(meta_data)
        if meta.ndim == 1 and meta.shape[0] in data_mat.shape:
            cols = {"batch_variable": meta}
            vars_use = "batch_variable"
        else:
            raise ValueError("meta_data must be either a data.frame or a vector with batch values for each cell")
    if vars_use is None:
        raise ValueError("must provide variables names (e.g. vars_use='stim')")
    if isinstance(vars_use, str):
        vars_use = [vars_use]
    if any(v not in cols for v in vars_use):
        raise ValueError("must provide variables names (e.g. vars_use='stim')")
    N = len(next(iter(cols.values())))
    if data_mat.ndim != 2:
        raise ValueError("data_mat must be a matrix")
    if data_mat.shape[0] == N:  # R/ui.R:178-183
        if verbose:
            _message("Transposing data matrix")
        data_mat = data_mat.T
    if data_mat.shape[1] != N:
        raise ValueError("number of labels do not correspond to number of samples in data matrix")
    if nclust is None:
        nclust = int(min(round(N / 30.0), 100))  # R/ui.R:192-194
    if theta is None:
        theta = [2.0] * len(vars_use)
    theta = list(np.atleast_1d(theta).astype(float))
    if len(theta) != len(vars_use):
        raise ValueError("Please specify theta for each variable")
    sigma = np.atleast_1d(np.asarray(sigma, dtype=float))
    if sigma.size == 1 and nclust > 1:
        sigma = np.repeat(sigma, nclust)

    codes, n_levels = [], []
    for v in vars_use:
        c, lv = as_factor(cols[v], None if levels is None else levels.get(v))
        codes.append(c)
        n_levels.append(len(lv))
    phi = build_phi(codes, n_levels)
    B_vec = np.asarray(n_levels, dtype=np.int32)
    if N_b is None:
        N_b = np.concatenate([np.bincount(c, minlength=n) for c, n in zip(codes, n_levels)]).astype(float)

    if lambda_ is None:  # R/ui.R:224-249
        if verbose:
            _message("Using automatic lambda estimation")
        lambda_vec = np.array([-1.0])
    else:
        lam = np.atleast_1d(np.asarray(lambda_, dtype=float))
        if not np.all(lam > 0):
            raise ValueError("Provided lambdas must be positive")
        if lam.size == 1:
            lambda_vec = np.concatenate([[0.0], np.repeat(lam, int(B_vec.sum()))])
        else:
            if lam.size != len(vars_use):
                raise ValueError("You specified a lambda value for each covariate but the number of lambdas "
                                 "specified (%d) and the number of covariates (%d) mismatch." % (lam.size, len(vars_use)))
            lambda_vec = np.concatenate([[0.0]] + [np.repeat(lam[b], B_vec[b]) for b in range(len(B_vec))])

    theta_lv = np.concatenate([np.repeat(theta[b], B_vec[b]) for b in range(len(B_vec))])  # R/ui.R:254-255
    if tau > 0:  # R/ui.R:258; tau == 0 leaves theta unchanged (1 - exp(-Inf))
        theta_lv = theta_lv * (1 - np.exp(-(np.asarray(N_b, dtype=float) / (nclust * tau)) ** 2))
    if verbose:
        _message("Thetas: " + " ".join(str(t) for t in np.unique(theta_lv)))

    kwargs = dict(Z=np.asfortranarray(data_mat, dtype=np.float64), Phi=phi, sigma=sigma, theta=theta_lv,
                  lambda_vec=lambda_vec, alpha=alpha, max_iter_kmeans=options["max_iter_cluster"],
                  epsilon_kmeans=options["epsilon_cluster"], epsilon_harmony=epsilon_harmony, K=int(nclust),
                  block_size=options["block_size"], B_vec=B_vec, batch_proportion_cutoff=options["batch_prop_cutoff"],
                  verbose=verbose)
    return kwargs, data_mat


def RunHarmony(data_mat, meta_data, vars_use=None, theta=None, sigma=0.1, lambda_=None, nclust=None, max_iter=10,
               early_stop=True, ncores=1, plot_convergence=False, return_object=False, verbose=True,
               options=None, seed=None, device=None, rng=None, **kwargs):
    """RunHarmony.default (R/ui.R:91-309).

    `lambda` is a Python keyword, so the ridge penalty is `lambda_` (``**{"lambda": x}`` also works);
    `.options` is `options`.  `ncores` is accepted and ignored (the reference uses it for BLAS threads,
    R/ui.R:114-128).  Returns the corrected embedding as cells x PCs whatever the orientation of the input was -- the
    reference returns t(Z_corr) unconditionally (R/ui.R:292-295) --, or the Harmony object.
    `seed` + `rng="R"`: draw the centroid seeds and the per-round shuffles as `set.seed(seed); RunHarmony(...)` does in R
    (MT19937, RcppArmadillo's draw order); default: the library's counter-based generator keyed by `seed`.
    """
    if "lambda" in kwargs:
        lambda_ = kwargs.pop("lambda")
    check_legacy_args(**kwargs)
    skw, _ = prepare_setup_args(data_mat, meta_data, vars_use, theta=theta, sigma=sigma, lambda_=lambda_,
                                nclust=nclust, early_stop=early_stop, verbose=verbose, options=options)
    harmonyObj = Harmony(device=device, seed=seed, rng=rng)
    harmonyObj.setup(**skw)
    if verbose:
        _message("Initializing state using k-means centroids initialization")
    harmonyObj.init_cluster_cpp()
    harmonize(harmonyObj, max_iter, verbose)
    if plot_convergence:
        raise NotImplementedError("plot_convergence: HarmonyConvergencePlot is out of scope; read "
                                  "obj.objective_kmeans / obj.kmeans_rounds from return_object=True")
    if return_object:
        return harmonyObj
    return harmonyObj.getZcorr().T  # R/ui.R:292-295
