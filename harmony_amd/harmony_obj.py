"""`Harmony`: the Python mirror of the reference's Rcpp-module object.

Same method and field names as ``class_<harmony>`` exposes to R
(/root/reference/src/harmony.cpp:672-709) -- ``setup``, ``init_cluster_cpp``, ``cluster_cpp``,
``moe_correct_ridge_cpp``, ``check_convergence``, ``compute_objective``, ``getZcorr``,
``getZorig``, ``getR``, ``getCentroids``, ``getLambda`` and the fields ``N B K d O E Y Pr_b
B_vec alpha W R theta sigma lambda kmeans_rounds objective_* max_iter_kmeans`` -- so that the
driver code (ui.py / utils.py, mirroring R/ui.R and R/utils.R) reads like the reference's.
Every method is a thin call through the C ABI; all numerics run in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _lib


class HarmonyError(RuntimeError):
    pass


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Harmony(object):
    """new(harmony)  (R/ui.R:269)."""

    def __init__(self, device=None, seed=None, rng=None, ridge_arith=None, oe_arith=None, obj_arith=None, solve_arith=None,
                 ref_arith=None, stale_dist=None):
        """rng: None/0 = the library's documented counter-based generator; 1 / "R" = R-compatible stream (MT19937 seeded like
        set.seed(seed), RcppArmadillo's randu / shuffle draw order).  ref_arith = 1: every accumulator group follows the reference's
        fp32 operation order (ridge statistics, O / E tables, objective sums, closed-form inverse; the groups can also be switched
        one by one: ridge_arith, oe_arith, obj_arith, solve_arith); ref_arith = 2: all groups but the O / E tables (4e-5 from the reference at 1M cells,
        a bit over half the time of ref_arith = 1); default: exact accumulators."""
        self._lib = _lib.load()
        self._h = C.c_void_p(self._lib.hmx_create())
        if not self._h:
            raise HarmonyError("hmx_create failed")
        self._keep = []  # keeps ctypes callbacks alive
        self.warnings = []
        if device is not None:
            self._set("device", int(device))
        if seed is not None:
            self._set("seed", int(seed))
        if rng:
            self._set("rng", 1)
        for name, v in (("ref_arith", ref_arith), ("ridge_arith", ridge_arith), ("oe_arith", oe_arith), ("obj_arith", obj_arith),
                        ("solve_arith", solve_arith), ("stale_dist", stale_dist)):
            if v:
                self._set(name, int(v))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.hmx_destroy(h)

    # ---- plumbing ----------------------------------------------------------------
    def _check(self, status, what):
        if status == 0:
            return 0
        if status == -1:
            return -1
        msg = self._lib.hmx_last_error(self._h).decode()
        raise HarmonyError("%s failed (status %d): %s" % (what, status, msg))

    def _set(self, field, value):
        self._check(self._lib.hmx_set_int(self._h, field.encode(), int(value)), "set " + field)

    def _get(self, field, shape=None):
        n = self._lib.hmx_get(self._h, field.encode(), None, 0)
        if n < 0:
            raise HarmonyError("unknown or unavailable field %r: %s" % (field, self._lib.hmx_last_error(self._h).decode()))
        out = np.empty(int(n), dtype=np.float64)
        if n:
            got = self._lib.hmx_get(self._h, field.encode(), _dptr(out), n)
            if got != n:
                raise HarmonyError("getter %r failed: %s" % (field, self._lib.hmx_last_error(self._h).decode()))
        if shape is not None:
            out = out.reshape(shape, order="F")
        return out

    def _scalar(self, field):
        return self._get(field)[0]

    # ---- distributed / runtime hooks (no reference counterpart) ---------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (call on rank 0, ship the bytes to the other ranks)."""
        buf = (C.c_uint8 * 128)()
        if _lib.load().hmx_comm_unique_id(buf) != 0:
            raise HarmonyError("hmx_comm_unique_id failed (is the system librccl loadable?)")
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.hmx_comm_init(self._h, int(rank), int(world), buf), "comm_init")

    # peer-to-peer block chain for hosts that bring their own all-reduce hook (comm_init does all of this by itself)
    def comm_allreduce_host(self, values, op="sum"):
        """all-reduce a few host doubles over the handle's communicator (also a barrier + device synchronisation): for hosts without a
        collective library of their own (bench.py --bootstrap file)"""
        a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._check(self._lib.hmx_comm_allreduce_host(self._h, _dptr(a), a.size, {"sum": 0, "max": 1, "min": 2}[op]), "comm_allreduce_host")
        return a

    def p2p_export(self):
        buf = (C.c_uint8 * 64)()
        self._check(self._lib.hmx_p2p_export(self._h, buf), "p2p_export")
        return bytes(buf)

    def p2p_connect(self, rank, world, handles):
        """handles: the ranks' exported handles in rank order (list of 64-byte strings)."""
        buf = (C.c_uint8 * (64 * int(world))).from_buffer_copy(b"".join(handles))
        self._check(self._lib.hmx_p2p_connect(self._h, int(rank), int(world), buf), "p2p_connect")

    def p2p_selftest(self):
        """collective: every rank at the same time; True if this rank heard every peer"""
        return self._lib.hmx_p2p_selftest(self._h) == 0

    def p2p_enable(self, on=True):
        self._check(self._lib.hmx_p2p_enable(self._h, int(bool(on))), "p2p_enable")

    @property
    def p2p_status(self):
        return self._lib.hmx_p2p_status(self._h).decode()

    def set_shard(self, rank, world, global_offset, N_global, allreduce_cb=None):
        """allreduce_cb None: use the built-in RCCL communicator (comm_init first)."""
        cb = None
        if allreduce_cb is not None:
            cb = _lib.ALLREDUCE_FN(allreduce_cb)
            self._keep.append(cb)
            cb = C.cast(cb, C.c_void_p)
        self._check(self._lib.hmx_set_shard(self._h, rank, world, int(global_offset), int(N_global), cb, None), "set_shard")

    def set_stream(self, hip_stream):
        self._check(self._lib.hmx_set_stream(self._h, C.c_void_p(hip_stream)), "set_stream")

    def set_abort_poll(self, fn):
        cb = _lib.POLL_FN(lambda _u: int(bool(fn())))
        self._keep.append(cb)
        self._check(self._lib.hmx_set_abort_poll(self._h, cb, None), "set_abort_poll")

    def push_update_order(self, order):
        order = np.ascontiguousarray(order, dtype=np.int64)
        self._check(self._lib.hmx_push_update_order(self._h, order.ctypes.data_as(C.POINTER(C.c_int64))), "push_update_order")

    def restart(self):
        self._check(self._lib.hmx_restart(self._h), "restart")

    def set_profile(self, on=True):
        """0 / False: off; 1 / True: the dominant kernel's launches carry start / stop events ("prof:*"); 2: every phase is bracketed by events
        as well ("gputimer:*" -- an event record is a packet of its own in the queue: ~200 of them per run cost 0.4-1 ms of a 15 ms run)"""
        self._set("profile", int(on))

    def timer(self, name):
        return self._scalar("timer:" + name)

    # ---- the reference's methods -----------------------------------------------------------
    def setup(self, Z, Phi, sigma, theta, lambda_vec, alpha, max_iter_kmeans, epsilon_kmeans, epsilon_harmony,
              K, block_size, B_vec, batch_proportion_cutoff, verbose):
        """harmony::setup (src/harmony.cpp:29-111).  Z: d x N; Phi: (i, p, x, B) CSC of the B x N design."""
        # Z: a numpy array (float64 = the R seam, or float32), or a device buffer (d, N, dtype, device_pointer) --
        # hmx_setup_ex ingests fp32 and/or HBM-resident embeddings without the fp64 host copy
        z_loc = 0
        if isinstance(Z, tuple):
            d, N, zdt, zptr = Z
            z_dtype = 1 if np.dtype(zdt) == np.float32 else 0
            z_loc, zarg = 1, C.c_void_p(int(zptr))
        else:
            z_dtype = 1 if getattr(Z, "dtype", None) == np.float32 else 0
            Z = np.asfortranarray(Z, dtype=np.float32 if z_dtype else np.float64)
            d, N = Z.shape
            zarg = C.c_void_p(Z.ctypes.data)
        phi_i, phi_p, phi_x, B = Phi
        phi_i = np.ascontiguousarray(phi_i, dtype=np.int32)
        phi_p = np.ascontiguousarray(phi_p, dtype=np.int32)
        phi_x = None if phi_x is None else np.ascontiguousarray(phi_x, dtype=np.float64)
        sigma = np.ascontiguousarray(np.atleast_1d(sigma), dtype=np.float64)
        theta = np.ascontiguousarray(np.atleast_1d(theta), dtype=np.float64)
        lam = np.ascontiguousarray(np.atleast_1d(lambda_vec), dtype=np.float64)
        B_vec = np.ascontiguousarray(np.atleast_1d(B_vec), dtype=np.int32)
        if sigma.size != K:
            raise HarmonyError("sigma must have one value per cluster")
        if theta.size != B:
            raise HarmonyError("theta must have one value per covariate level")
        st = self._lib.hmx_setup_ex(self._h, zarg, z_dtype, z_loc, N, d, _iptr(phi_i), _iptr(phi_p),
                                    None if phi_x is None else _dptr(phi_x), int(B), _dptr(sigma), _dptr(theta),
                                    _dptr(lam), lam.size, float(alpha), int(max_iter_kmeans), float(epsilon_kmeans),
                                    float(epsilon_harmony), int(K), float(block_size), _iptr(B_vec), B_vec.size,
                                    float(batch_proportion_cutoff), int(bool(verbose)))
        self._check(st, "setup")
        w = self._lib.hmx_last_warning(self._h).decode()
        if w:
            import warnings
            self.warnings.append(w)
            warnings.warn(w)

    def init_cluster_cpp(self, Y0=None):
        if Y0 is None:
            st = self._lib.hmx_init_cluster(self._h, None)
        else:
            Y0 = np.asfortranarray(Y0, dtype=np.float64)
            st = self._lib.hmx_init_cluster(self._h, _dptr(Y0))
        self._check(st, "init_cluster_cpp")

    def kmeans_centers(self):
        out = np.empty((int(self.d), int(self.K)), dtype=np.float64, order="F")
        self._check(self._lib.hmx_kmeans_centers(self._h, _dptr(out)), "kmeans_centers")
        return out

    def cluster_cpp(self):
        return self._check(self._lib.hmx_cluster(self._h), "cluster_cpp")

    def moe_correct_ridge_cpp(self):
        return self._check(self._lib.hmx_moe_correct_ridge(self._h), "moe_correct_ridge_cpp")

    def check_convergence(self, type_):
        r = self._lib.hmx_check_convergence(self._h, int(type_))
        if r < 0:
            raise HarmonyError("check_convergence: " + self._lib.hmx_last_error(self._h).decode())
        return bool(r)

    def compute_objective(self):
        self._check(self._lib.hmx_compute_objective(self._h), "compute_objective")

    def getZcorr(self):
        return self._get("Z_corr", (int(self.d), int(self._scalar("N_local"))))

    def get_matrix(self, field, dtype=np.float64, device_ptr=None):
        """Z_corr / Z_orig / R as float64 or float32, into a new numpy array or (device_ptr given) into the caller's HBM
        buffer -- hmx_get_matrix."""
        w = int(self.K) if field == "R" else int(self.d)
        n = int(self._scalar("N_local"))
        f32 = 1 if np.dtype(dtype) == np.float32 else 0
        if device_ptr is not None:
            got = self._lib.hmx_get_matrix(self._h, field.encode(), C.c_void_p(int(device_ptr)), f32, 1, n * w)
            if got != n * w:
                raise HarmonyError("get_matrix(%s) failed: %s" % (field, self._lib.hmx_last_error(self._h).decode()))
            return None
        out = np.empty((w, n), dtype=np.float32 if f32 else np.float64, order="F")
        got = self._lib.hmx_get_matrix(self._h, field.encode(), C.c_void_p(out.ctypes.data), f32, 0, n * w)
        if got != n * w:
            raise HarmonyError("get_matrix(%s) failed: %s" % (field, self._lib.hmx_last_error(self._h).decode()))
        return out

    def getZorig(self):
        return self._get("Z_orig", (int(self.d), int(self._scalar("N_local"))))

    def getR(self):
        return self._get("R", (int(self.K), int(self._scalar("N_local"))))

    def getCentroids(self):
        return self.Y

    def getLambda(self):
        return self._get("Lambda", (int(self.K), int(self.B) + 1))

    # ---- the reference's fields -----------------------------------------------------------------
    N = property(lambda s: int(s._scalar("N")))
    B = property(lambda s: int(s._scalar("B")))
    K = property(lambda s: int(s._scalar("K")))
    d = property(lambda s: int(s._scalar("d")))
    alpha = property(lambda s: float(s._scalar("alpha")))
    O = property(lambda s: s._get("O", (s.K, s.B)))
    E = property(lambda s: s._get("E", (s.K, s.B)))
    Y = property(lambda s: s._get("Y", (s.d, s.K)))
    R = property(lambda s: s.getR())
    Pr_b = property(lambda s: s._get("Pr_b"))
    B_vec = property(lambda s: s._get("B_vec").astype(int))
    theta = property(lambda s: s._get("theta"))
    sigma = property(lambda s: s._get("sigma"))
    kmeans_rounds = property(lambda s: s._get("kmeans_rounds").astype(int))
    objective_kmeans = property(lambda s: s._get("objective_kmeans"))
    objective_kmeans_dist = property(lambda s: s._get("objective_kmeans_dist"))
    objective_kmeans_entropy = property(lambda s: s._get("objective_kmeans_entropy"))
    objective_kmeans_cross = property(lambda s: s._get("objective_kmeans_cross"))
    objective_harmony = property(lambda s: s._get("objective_harmony"))

    @property
    def W(self):
        return self._get("W", (int(self._scalar("W_rows")), self.d))

    @property
    def max_iter_kmeans(self):
        return int(self._scalar("max_iter_kmeans"))

    @max_iter_kmeans.setter
    def max_iter_kmeans(self, v):  # vignettes/detailedWalkthrough.Rmd:364 writes this field
        self._set("max_iter_kmeans", v)


setattr(Harmony, "lambda", property(lambda s: s._get("lambda")))
