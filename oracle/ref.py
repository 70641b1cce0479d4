"""ctypes wrapper of oracle/_ref/libharmony_ref.so: the REFERENCE'S OWN engine sources (/root/reference/src/harmony.cpp, utils.cpp,
timer.cpp -- compiled where they lie, unmodified) over oracle/shim/ (a minimal stand-in for the Armadillo / Rcpp headers this image
lacks; oracle/shim/arma_min.hpp says exactly what it restates).

TEST INFRASTRUCTURE ONLY: the thing the restated oracle (oracle/harmony_oracle.cpp) is checked against, bit for bit
(tests/test_oracle_ref.py).  Built only where /root/reference exists (`make -C oracle _ref`); the built file travels with the tree.
`RefHarmony` has the method / field names of the reference's module object, like OracleHarmony and harmony_amd.Harmony.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("HARMONY_REF_SO", os.path.join(_HERE, "_ref", "libharmony_ref.so"))
_SO64 = os.path.join(os.path.dirname(_SO), "libharmony_ref_f64.so")    # the same sources with -DHARMONY_SCALAR_DOUBLE (src/types.h:5-9)
_REF_SRC = os.environ.get("HARMONY_REFERENCE_SRC", "/root/reference/src")
_lib = None
_lib64 = None


def available():
    """True when the library exists or can be built here (the reference's sources are on disk)."""
    return os.path.exists(_SO) or os.path.exists(os.path.join(_REF_SRC, "harmony.cpp"))


def build(force=False):
    """(Re)build where the reference's sources exist; elsewhere use the file that travelled with the tree."""
    if os.path.exists(os.path.join(_REF_SRC, "harmony.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "_ref", "REF=" + _REF_SRC])
    if not os.path.exists(_SO):
        raise RuntimeError("oracle/_ref/libharmony_ref.so absent and /root/reference is not here to build it from")
    return _SO


def load(double=False):
    """double=True: the double-precision build of the reference's sources"""
    global _lib, _lib64
    if double:
        if _lib64 is None:
            build()
            if not os.path.exists(_SO64):
                raise RuntimeError("oracle/_ref/libharmony_ref_f64.so absent and /root/reference is not here to build it from")
            _lib64 = _prototypes(C.CDLL(_SO64))
            assert _lib64.ref_scalar_bytes() == 8
        return _lib64
    if _lib is None:
        _lib = _prototypes(C.CDLL(build()))
        assert _lib.ref_scalar_bytes() == 4
    return _lib


def use_lapack(mode):
    """arma::inv of the single-precision build through a real LAPACK (scipy's OpenBLAS) instead of the stand-in's unblocked LU:
    mode 1 = sgetrf + sgetri, 2 = spotrf + spotri + mirror, 0 = back to the stand-in (oracle/lapack_inv.hpp).  False if there is no LAPACK."""
    from . import oracle as _orc
    lib = load()
    if mode:
        p = _orc.lapack_pointers()
        if p is None:
            return False
        lib.ref_set_lapack.argtypes = [C.c_void_p] * 4
        lib.ref_set_lapack.restype = None
        lib.ref_set_lapack(*p)
    lib.ref_set_inv_mode.argtypes = [C.c_int]
    lib.ref_set_inv_mode.restype = None
    lib.ref_set_inv_mode(int(mode))
    return True


def use_sgemm(on, threads=1):
    """the distance GEMM Y.t() * Z_corr of the single-precision build through scipy's OpenBLAS sgemm('T', 'N') -- as Armadillo's glue_times calls
    it, and as oracle.use_openblas() routes the oracle's; False: back to the stand-in's sequential dot products."""
    from . import oracle as _orc
    lib = load()
    lib.ref_set_sgemm.argtypes = [C.c_void_p]
    lib.ref_set_sgemm.restype = None
    if not on:
        lib.ref_set_sgemm(None)
        return True
    if _orc.lapack_pointers() is None:
        return False
    try:
        _orc._blas.scipy_openblas_set_num_threads(int(threads))
    except AttributeError:
        pass
    lib.ref_set_sgemm(C.cast(_orc._blas.scipy_cblas_sgemm, C.c_void_p))
    return True


def use_blas_norms(on):
    """norm(col, p) inside normalise() as Armadillo's op_norm forms it on a BLAS: two accumulators below 32 elements, scipy's OpenBLAS sasum /
    snrm2 from 32 on (oracle/lapack_inv.hpp, blas1); False: back to the stand-in's single accumulator."""
    from . import oracle as _orc
    lib = load()
    if on:
        if _orc.lapack_pointers() is None:
            return False
        b1 = [C.cast(getattr(_orc._blas, "scipy_cblas_" + f), C.c_void_p) for f in ("sasum", "snrm2")]
        lib.ref_set_blas1.argtypes = [C.c_void_p] * 2
        lib.ref_set_blas1.restype = None
        lib.ref_set_blas1(*b1)
    lib.ref_set_norm_mode.argtypes = [C.c_int]
    lib.ref_set_norm_mode.restype = None
    lib.ref_set_norm_mode(1 if on else 0)
    return True


def _prototypes(lib):
    if True:
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        lib.ref_create.restype = C.c_void_p
        lib.ref_destroy.argtypes = [C.c_void_p]
        lib.ref_last_error.restype = C.c_char_p
        lib.ref_last_error.argtypes = [C.c_void_p]
        lib.ref_setup.argtypes = [C.c_void_p, dp, C.c_int64, C.c_int, ip, ip, C.c_int, dp, dp, dp, C.c_int, C.c_double,
                                  C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, ip, C.c_int, C.c_double]
        lib.ref_init_cluster.argtypes = [C.c_void_p, C.c_uint64]
        for f in ("ref_cluster", "ref_moe_correct_ridge", "ref_compute_objective"):
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.ref_check_convergence.argtypes = [C.c_void_p, C.c_int]
        lib.ref_get.restype = C.c_int64
        lib.ref_get.argtypes = [C.c_void_p, C.c_char_p, dp]
        lib.ref_push_update_order.argtypes = [C.c_void_p, lp]
        lib.ref_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.ref_init_cluster_from.argtypes = [C.c_void_p, dp, C.c_uint64]
        lib.ref_clear_update_orders.restype = None
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class RefHarmony(object):
    def __init__(self, seed=0, double=False):
        """double=True: the reference's sources built with their own precision switch (SCALAR = double)"""
        self._lib = load(double)
        self._h = C.c_void_p(self._lib.ref_create())
        self.seed = int(seed)
        self._dims = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ref_destroy(h)

    def _err(self):
        return self._lib.ref_last_error(self._h).decode()

    def _get(self, name, shape=None):
        n = self._lib.ref_get(self._h, name.encode(), None)
        if n < 0:
            raise KeyError(name + ": " + self._err())
        out = np.empty(int(n), dtype=np.float64)
        if n:
            self._lib.ref_get(self._h, name.encode(), _dp(out))
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
        st = self._lib.ref_setup(self._h, _dp(Z), N, d, phi_i.ctypes.data_as(ip), phi_p.ctypes.data_as(ip), int(B),
                                 _dp(sigma), _dp(theta), _dp(lam), lam.size, float(alpha), int(max_iter_kmeans),
                                 float(epsilon_kmeans), float(epsilon_harmony), int(K), float(block_size),
                                 B_vec.ctypes.data_as(ip), B_vec.size, float(batch_proportion_cutoff))
        if st:
            raise RuntimeError("reference setup failed (%d): %s" % (st, self._err()))
        self._dims = (N, d, int(K), int(B))

    def _call(self, name, *a):
        st = getattr(self._lib, name)(self._h, *a)
        if st >= 100:
            raise RuntimeError("%s raised: %s" % (name, self._err()))
        return st

    def init_cluster_cpp(self, Y0=None):
        """Y0 given: init from these centroids (ref_driver.cpp: ref_init_cluster_from -- every number still from the reference's own lines)"""
        if Y0 is None:
            assert self._call("ref_init_cluster", self.seed) == 0
        else:
            Y0 = np.asfortranarray(Y0, dtype=np.float64)
            assert self._call("ref_init_cluster_from", _dp(Y0), self.seed) == 0

    def clear_update_orders(self):
        self._lib.ref_clear_update_orders()

    def push_update_order(self, order):
        order = np.ascontiguousarray(order, dtype=np.int64)
        self._lib.ref_push_update_order(self._h, order.ctypes.data_as(C.POINTER(C.c_int64)))

    def cluster_cpp(self):
        return self._call("ref_cluster")

    def moe_correct_ridge_cpp(self):
        self._call("ref_moe_correct_ridge")

    def check_convergence(self, t):
        return bool(self._call("ref_check_convergence", int(t)))

    def compute_objective(self):
        self._call("ref_compute_objective")

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
    block_size = property(lambda s: float(s._get("block_size")[0]))
    kmeans_rounds = property(lambda s: s._get("kmeans_rounds").astype(int))
    update_order = property(lambda s: s._get("update_order").astype(np.int64))
    objective_kmeans = property(lambda s: s._get("objective_kmeans"))
    objective_kmeans_dist = property(lambda s: s._get("objective_kmeans_dist"))
    objective_kmeans_entropy = property(lambda s: s._get("objective_kmeans_entropy"))
    objective_kmeans_cross = property(lambda s: s._get("objective_kmeans_cross"))
    objective_harmony = property(lambda s: s._get("objective_harmony"))

    @property
    def W(self):
        return self._get("W", (int(self._get("W_rows")[0]), self.d))

    @property
    def max_iter_kmeans(self):
        raise AttributeError

    @max_iter_kmeans.setter
    def max_iter_kmeans(self, v):
        self._lib.ref_set_int(self._h, b"max_iter_kmeans", int(v))

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
