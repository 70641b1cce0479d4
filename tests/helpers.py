"""Shared test helpers: drive any backend (oracle or HIP) with the reference's driver logic."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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


from bench_data import synth  # noqa: E402,F401
