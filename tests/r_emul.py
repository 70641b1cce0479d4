"""TEST INFRASTRUCTURE: drives r/harmony_mi355x_glue.c -- the .Call glue a maintainer would build into the R package -- through an executable
emulation of the part of R's C API it uses (tests/stubs/r_emul.c; R itself is not installed here).  `GlueHarmony` is r/harmony_mi355x.R's
`new_harmony_mi355x()` line by line: the same .Call names (looked up in the table R_init_harmony registers), the same arguments in the same
order with the same R types (numeric / integer / logical / character, `dim` attributes), the same post-processing of the results."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "tests", "stubs", "_build", "libharmony_glue_emul.so")
_lib = None


def build():
    srcs = [os.path.join(ROOT, "r", "harmony_mi355x_glue.c"), os.path.join(ROOT, "tests", "stubs", "r_emul.c")]
    libdir = os.path.join(ROOT, "harmony_amd", "lib")
    deps = srcs + [os.path.join(ROOT, "tests", "stubs", "R.h"), os.path.join(ROOT, "include", "harmony_mi355x.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(f) for f in deps):
        return _SO
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call([gcc, "-std=c11", "-O1", "-fPIC", "-shared", "-Wall", "-Werror=implicit-function-declaration", "-Werror=int-conversion",
                           "-Werror=incompatible-pointer-types", "-Wno-cast-function-type", "-I" + os.path.join(ROOT, "tests", "stubs"),
                           "-I" + os.path.join(ROOT, "include")] + srcs + ["-L" + libdir, "-lharmony_mi355x", "-Wl,-rpath," + libdir, "-o", _SO])
    return _SO


def load():
    global _lib
    if _lib is None:
        from harmony_amd import _lib as product
        product.load()                                   # (the product library first: same search path as every other test)
        lib = C.CDLL(build())
        V = C.c_void_p
        lib.emul_real.restype = V
        lib.emul_real.argtypes = [C.POINTER(C.c_double), C.c_ssize_t, C.c_int, C.c_int]
        lib.emul_int.restype = V
        lib.emul_int.argtypes = [C.POINTER(C.c_int), C.c_ssize_t, C.c_int, C.c_int]
        lib.emul_logical.restype = V
        lib.emul_logical.argtypes = [C.c_int]
        lib.emul_string.restype = V
        lib.emul_string.argtypes = [C.c_char_p]
        lib.emul_type.argtypes = [V]
        lib.emul_length.restype = C.c_ssize_t
        lib.emul_length.argtypes = [V]
        lib.emul_data.restype = V
        lib.emul_data.argtypes = [V]
        lib.emul_dim.argtypes = [V, C.c_int]
        lib.emul_last_error.restype = C.c_char_p
        lib.emul_warnings.restype = C.c_char_p
        lib.emul_release.argtypes = [V]
        lib.emul_set_seed.argtypes = [C.c_uint32]
        lib.emul_set_interrupt.argtypes = [C.c_int]
        lib.emul_call.restype = V
        lib.emul_call.argtypes = [V, C.c_int, C.POINTER(V)]
        lib.emul_lookup.restype = V
        lib.emul_lookup.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        lib.R_init_harmony.argtypes = [V]
        lib.R_init_harmony(None)                         # what R does when the package's shared object is loaded
        _lib = lib
    return _lib


class RError(RuntimeError):
    pass


REALSXP, INTSXP, LGLSXP = 14, 13, 10


def numeric(x, dim=None):
    a = np.asfortranarray(x, dtype=np.float64)
    nr, nc = (dim if dim is not None else (a.shape if a.ndim == 2 else (0, 0)))
    return C.c_void_p(load().emul_real(a.ctypes.data_as(C.POINTER(C.c_double)), a.size, int(nr), int(nc)))


def integer(x, dim=None):
    a = np.asfortranarray(x, dtype=np.int32)
    nr, nc = (dim if dim is not None else (a.shape if a.ndim == 2 else (0, 0)))
    return C.c_void_p(load().emul_int(a.ctypes.data_as(C.POINTER(C.c_int)), a.size, int(nr), int(nc)))


def logical(v):
    return C.c_void_p(load().emul_logical(int(bool(v))))


def character(s):
    return C.c_void_p(load().emul_string(s.encode()))


def value(sexp):
    """an R vector / matrix back as numpy (column-major with its dim attribute), NULL as None"""
    lib = load()
    t, n = lib.emul_type(sexp), lib.emul_length(sexp)
    if t == 0:
        return None
    ct, dt = {REALSXP: (C.c_double, np.float64), INTSXP: (C.c_int, np.int32), LGLSXP: (C.c_int, np.int32)}[t]
    a = np.ctypeslib.as_array(C.cast(lib.emul_data(sexp), C.POINTER(ct)), shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)
    nr = lib.emul_dim(sexp, 0)
    if nr >= 0:
        a = a.reshape((nr, lib.emul_dim(sexp, 1)), order="F")
    return a.astype(bool) if t == LGLSXP else a


def dot_call(name, *args):
    """.Call(name, ...): the registered routine, its registered argument count, R errors as exceptions"""
    lib = load()
    n = C.c_int(-1)
    fn = lib.emul_lookup(name.encode(), C.byref(n))
    if not fn:
        raise RError('C symbol name "%s" not in the registration table' % name)
    if n.value != len(args):
        raise RError("Incorrect number of arguments (%d), expecting %d for '%s'" % (len(args), n.value, name))
    arr = (C.c_void_p * max(1, len(args)))(*[a.value if isinstance(a, C.c_void_p) else a for a in args])
    lib.emul_clear()
    out = lib.emul_call(fn, len(args), arr)
    if not out:
        raise RError(lib.emul_last_error().decode())
    return C.c_void_p(out)


class GlueHarmony(object):
    """new_harmony_mi355x() of r/harmony_mi355x.R"""

    def __init__(self, seed=None, r_rng=False, reference_arithmetic=False):
        self.ptr = dot_call("C_hmx_new")
        if reference_arithmetic:
            dot_call("C_hmx_set_int", self.ptr, character("ref_arith"), numeric([1]))
        if r_rng:
            dot_call("C_hmx_use_r_rng", self.ptr)
        else:
            dot_call("C_hmx_set_seed", self.ptr, numeric([float(seed if seed is not None else 1)]))

    def release(self):                   # the garbage collector running the finalizer
        load().emul_release(self.ptr)

    def warnings(self):
        return load().emul_warnings().decode()

    def get(self, field):
        return value(dot_call("C_hmx_get", self.ptr, character(field)))

    def mat(self, field, nr, nc):
        return self.get(field).reshape((int(nr), int(nc)), order="F")

    def setup(self, Z, Phi, sigma, theta, lambda_vec, alpha, max_iter_kmeans, epsilon_kmeans, epsilon_harmony, K, block_size, B_vec,
              batch_proportion_cutoff, verbose, single=False):
        phi_i, phi_p, phi_x, B = Phi
        if phi_x is None:                      # as.numeric(Phi@x): the design's unit values
            phi_x = np.ones(len(phi_i))
        if single:   # float::float32: @Data is an INTEGER matrix with the fp32 bits
            Zs = integer(np.asfortranarray(Z, dtype=np.float32).view(np.int32))
        else:
            Zs = numeric(np.asfortranarray(Z, dtype=np.float64))
        dot_call("C_hmx_setup_f32" if single else "C_hmx_setup", self.ptr, Zs, integer(phi_i), integer(phi_p), numeric(phi_x), integer([B]),
                 numeric(np.atleast_1d(sigma)), numeric(np.atleast_1d(theta)), numeric(np.atleast_1d(lambda_vec)), numeric([alpha]),
                 integer([max_iter_kmeans]), numeric([epsilon_kmeans]), numeric([epsilon_harmony]), integer([K]), numeric([block_size]),
                 integer(np.atleast_1d(B_vec)), numeric([batch_proportion_cutoff]), logical(verbose))

    def init_cluster_cpp(self):
        dot_call("C_hmx_init_cluster", self.ptr)

    def cluster_cpp(self):
        return int(value(dot_call("C_hmx_cluster", self.ptr))[0])

    def moe_correct_ridge_cpp(self):
        dot_call("C_hmx_moe_correct_ridge", self.ptr)

    def check_convergence(self, t):
        return bool(value(dot_call("C_hmx_check_convergence", self.ptr, integer([t])))[0])

    def compute_objective(self):
        dot_call("C_hmx_compute_objective", self.ptr)

    def getZcorr(self, single=False):
        d, N = int(self.get("d")[0]), int(self.get("N")[0])
        if single:
            bits = value(dot_call("C_hmx_get_matrix_f32", self.ptr, character("Z_corr"), integer([d]), integer([N])))
            return bits.view(np.float32)
        return self.mat("Z_corr", d, N)

    def set_max_iter_kmeans(self, v):
        dot_call("C_hmx_set_int", self.ptr, character("max_iter_kmeans"), numeric([v]))

    R = property(lambda s: s.mat("R", s.get("K")[0], s.get("N")[0]))
    O = property(lambda s: s.mat("O", s.get("K")[0], s.get("B")[0]))
    E = property(lambda s: s.mat("E", s.get("K")[0], s.get("B")[0]))
    Y = property(lambda s: s.mat("Y", s.get("d")[0], s.get("K")[0]))
    objective_kmeans = property(lambda s: s.get("objective_kmeans"))
    kmeans_rounds = property(lambda s: s.get("kmeans_rounds").astype(int))

    def getLambda(self):
        return self.mat("Lambda", self.get("K")[0], self.get("B")[0] + 1)
