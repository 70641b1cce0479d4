"""ctypes wrapper of the CPU oracle (oracle/harmony_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the harmony_amd package.  How far parity is pinned (to the reference's own
sources over a stand-in for Armadillo: oracle/ref.py; Armadillo's kernels remain restated): header of harmony_oracle.cpp.

`OracleHarmony` exposes the same method/field names as the reference's module object (and as
harmony_amd.Harmony), so `harmony_amd.utils.harmonize` can drive either.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libharmony_oracle.so")
_lib = None
_blas = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("harmony_oracle.cpp", "lapack_inv.hpp")]
    if force or not os.path.exists(_SO) or max(os.path.getmtime(f) for f in src) > os.path.getmtime(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libharmony_oracle.so"])
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(_SO)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.c_int]
        lib.orc_create_mask.restype = C.c_void_p
        lib.orc_create_mask.argtypes = [C.c_uint]
        lib.orc_destroy.argtypes = [C.c_void_p]
        lib.orc_set_sgemm.argtypes = [C.c_void_p]
        lib.orc_setup.argtypes = [C.c_void_p, dp, C.c_int64, C.c_int, ip, ip, C.c_int, dp, dp, dp, C.c_int, C.c_double,
                                  C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, ip, C.c_int, C.c_double]
        lib.orc_init_cluster.argtypes = [C.c_void_p, dp, C.c_uint64]
        for f in ("orc_cluster", "orc_moe_correct_ridge"):
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.orc_compute_objective.argtypes = [C.c_void_p]
        lib.orc_compute_objective.restype = None
        lib.orc_check_convergence.argtypes = [C.c_void_p, C.c_int]
        lib.orc_get.restype = C.c_int64
        lib.orc_get.argtypes = [C.c_void_p, C.c_char_p, dp]
        lib.orc_push_update_order.argtypes = [C.c_void_p, lp]
        lib.orc_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.orc_last_error.restype = C.c_char_p
        lib.orc_last_error.argtypes = [C.c_void_p]
        lib.orc_feistel_pos.restype = C.c_uint64
        lib.orc_feistel_pos.argtypes = [C.c_uint64] * 4
        lib.orc_u01.restype = C.c_float
        lib.orc_u01.argtypes = [C.c_uint64] * 3
        lib.orc_r_runif.restype = None
        lib.orc_r_runif.argtypes = [C.c_uint32, C.c_int, dp]
        lib.orc_r_shuffle.restype = None
        lib.orc_r_shuffle.argtypes = [C.c_uint32, C.c_int64, lp]
        _lib = lib
    return _lib


def use_openblas(threads=1):
    """Route the oracle's K x d x N GEMM through scipy's bundled OpenBLAS (cblas_sgemm)."""
    global _blas
    import scipy
    cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    if not cands:
        return False
    _blas = C.CDLL(cands[0])
    try:
        _blas.scipy_openblas_set_num_threads(int(threads))
    except AttributeError:
        pass
    fn = C.cast(_blas.scipy_cblas_sgemm, C.c_void_p)
    load().orc_set_sgemm(fn)
    return True


def lapack_pointers():
    """(sgetrf_, sgetri_, spotrf_, spotri_) of scipy's bundled OpenBLAS as void pointers, or None (oracle/lapack_inv.hpp)"""
    global _blas
    if _blas is None:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        if not cands:
            return None
        _blas = C.CDLL(cands[0])
    try:
        return [C.cast(getattr(_blas, "scipy_" + f), C.c_void_p) for f in ("sgetrf_", "sgetri_", "spotrf_", "spotri_")]
    except AttributeError:
        return None


def use_lapack():
    """Make liberty bits 5 / 6 (arma::inv through a real LAPACK) available to the oracle; False if no LAPACK is found."""
    p = lapack_pointers()
    if p is None:
        return False
    lib = load()
    lib.orc_set_lapack.argtypes = [C.c_void_p] * 4
    lib.orc_set_lapack.restype = None
    lib.orc_set_lapack(*p)
    try:        # BLAS level 1 for liberty bit 7 (norms as Armadillo's op_norm forms them on a BLAS)
        b1 = [C.cast(getattr(_blas, "scipy_cblas_" + f), C.c_void_p) for f in ("sasum", "snrm2")]
        lib.orc_set_blas1.argtypes = [C.c_void_p] * 2
        lib.orc_set_blas1.restype = None
        lib.orc_set_blas1(*b1)
    except AttributeError:
        return False
    return True


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleHarmony(object):
    def __init__(self, accurate=True, seed=0, mask=None, rng=0, liberty=0):
        """mask: per-group arithmetic (bit 0 O/E tables, 1 objective sums, 2 ridge statistics, 3 ridge solve; bit set =
        fp64, clear = the reference's fp32).  accurate=True is mask 15, accurate=False (faithful) is mask 0.
        liberty: the places where Armadillo / BLAS -- not /root/reference -- fix the operation order, flipped one by one (bit 0 / 1 L1 sums
        with two / eight accumulators, 2 one rounded product per non-zero in the several-covariate apply, 3 fp32 Lloyd sums, 4 L2 norms in
        double, 5 / 6 arma::inv through LAPACK's sgetrf + sgetri / spotrf + spotri, 7 norms through Armadillo's op_norm on BLAS sasum / snrm2 -- after
        use_lapack(); harmony_oracle.cpp header)."""
        self._lib = load()
        self.mask = (15 if accurate else 0) if mask is None else int(mask)
        self._h = C.c_void_p(self._lib.orc_create_mask(self.mask))
        if rng:   # 1: R-compatible stream (MT19937 seeded like set.seed(seed), RcppArmadillo draw order)
            self._lib.orc_set_int(self._h, b"rng", int(rng))
        if liberty:
            self._lib.orc_set_int(self._h, b"liberty", int(liberty))
        self.seed = int(seed)
        self._dims = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.orc_destroy(h)

    def _get(self, name, shape=None):
        n = self._lib.orc_get(self._h, name.encode(), None)
        if n < 0:
            raise KeyError(name)
        out = np.empty(int(n), dtype=np.float64)
        if n:
            self._lib.orc_get(self._h, name.encode(), _dp(out))
        return out.reshape(shape, order="F") if shape is not None else out

    def setup(self, Z, Phi, sigma, theta, lambda_vec, alpha, max_iter_kmeans, epsilon_kmeans, epsilon_harmony,
              K, block_size, B_vec, batch_proportion_cutoff, verbose):
        Z = np.asfortranarray(Z, dtype=np.float64)
        d, N = Z.shape
        phi_i, phi_p, _x, B = Phi
        phi_i = np.ascontiguousarray(phi_i, dtype=np.int32)
        phi_p = np.ascontiguousarray(phi_p, dtype=np.int32)
        sigma = np.ascontiguousarray(np.atleast_1d(sigma), dtype=np.float64)
        theta = np.ascontiguousarray(np.atleast_1d(theta), dtype=np.float64)
        lam = np.ascontiguousarray(np.atleast_1d(lambda_vec), dtype=np.float64)
        B_vec = np.ascontiguousarray(np.atleast_1d(B_vec), dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        eps_h = float(epsilon_harmony)
        st = self._lib.orc_setup(self._h, _dp(Z), N, d, phi_i.ctypes.data_as(ip), phi_p.ctypes.data_as(ip), int(B),
                                 _dp(sigma), _dp(theta), _dp(lam), lam.size, float(alpha), int(max_iter_kmeans),
                                 float(epsilon_kmeans), eps_h, int(K), float(block_size), B_vec.ctypes.data_as(ip),
                                 B_vec.size, float(batch_proportion_cutoff))
        if st:
            raise RuntimeError("oracle setup failed (%d): %s" % (st, self._lib.orc_last_error(self._h).decode()))
        self._dims = (N, d, int(K), int(B))

    def init_cluster_cpp(self, Y0=None):
        if Y0 is None:
            st = self._lib.orc_init_cluster(self._h, None, self.seed)
        else:
            Y0 = np.asfortranarray(Y0, dtype=np.float64)
            st = self._lib.orc_init_cluster(self._h, _dp(Y0), self.seed)
        assert st == 0

    def push_update_order(self, order):
        order = np.ascontiguousarray(order, dtype=np.int64)
        self._lib.orc_push_update_order(self._h, order.ctypes.data_as(C.POINTER(C.c_int64)))

    def cluster_cpp(self):
        return self._lib.orc_cluster(self._h)

    def moe_correct_ridge_cpp(self):
        st = self._lib.orc_moe_correct_ridge(self._h)
        if st:
            raise RuntimeError("oracle moe failed: " + self._lib.orc_last_error(self._h).decode())

    def check_convergence(self, t):
        return bool(self._lib.orc_check_convergence(self._h, int(t)))

    def compute_objective(self):
        self._lib.orc_compute_objective(self._h)

    def timer(self, name):
        return self._get("timer:" + name)[0]

    N = property(lambda s: s._dims[0])
    d = property(lambda s: s._dims[1])
    K = property(lambda s: s._dims[2])
    B = property(lambda s: s._dims[3])
    R = property(lambda s: s._get("R", (s.K, s.N)))
    Y = property(lambda s: s._get("Y", (s.d, s.K)))
    O = property(lambda s: s._get("O", (s.K, s.B)))
    E = property(lambda s: s._get("E", (s.K, s.B)))
    dist_mat = property(lambda s: s._get("dist", (s.K, s.N)))
    Pr_b = property(lambda s: s._get("Pr_b"))
    kmeans_rounds = property(lambda s: s._get("kmeans_rounds").astype(int))
    objective_kmeans = property(lambda s: s._get("objective_kmeans"))
    objective_kmeans_dist = property(lambda s: s._get("objective_kmeans_dist"))
    objective_kmeans_entropy = property(lambda s: s._get("objective_kmeans_entropy"))
    objective_kmeans_cross = property(lambda s: s._get("objective_kmeans_cross"))
    objective_harmony = property(lambda s: s._get("objective_harmony"))
    subset_clusters = property(lambda s: int(s._get("subset_clusters")[0]))
    skipped_clusters = property(lambda s: int(s._get("skipped_clusters")[0]))

    @property
    def W(self):
        return self._get("W", (int(self._get("W_rows")[0]), self.d))

    @property
    def max_iter_kmeans(self):
        raise AttributeError

    @max_iter_kmeans.setter
    def max_iter_kmeans(self, v):
        self._lib.orc_set_int(self._h, b"max_iter_kmeans", int(v))

    def getZcorr(self):
        return self._get("Z_corr", (self.d, self.N))

    def getZorig(self):
        return self._get("Z_orig", (self.d, self.N))

    def getR(self):
        return self.R

    def getCentroids(self):
        return self.Y

    def getLambda(self):
        return self._get("Lambda", (self.K, self.B + 1))


def feistel_order(seed, rnd, N):
    """update_order of round `rnd` under the documented generator: order[pos(g)] = g."""
    lib = load()
    order = np.empty(N, dtype=np.int64)
    for g in range(N):
        order[lib.orc_feistel_pos(seed, rnd, N, g)] = g
    return order


def r_runif(seed, n):
    """n uniforms of R's default generator after set.seed(seed) (oracle's own MT19937 restatement)."""
    out = np.empty(n, dtype=np.float64)
    load().orc_r_runif(int(seed), int(n), _dp(out))
    return out


def r_shuffle(seed, N):
    """arma::shuffle(0..N-1) on R's stream after set.seed(seed)."""
    out = np.empty(N, dtype=np.int64)
    load().orc_r_shuffle(int(seed), int(N), out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out
