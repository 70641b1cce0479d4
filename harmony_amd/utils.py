"""harmonize(): the outer loop of the reference, R/utils.R:15-46, verbatim in Python."""
import sys


def _message(msg):
    print(msg, file=sys.stderr)


def harmonize(harmonyObj, iter_harmony, verbose=True):
    """Works on any object exposing cluster_cpp / moe_correct_ridge_cpp / check_convergence."""
    if iter_harmony < 1:
        return 0
    for it in range(1, iter_harmony + 1):
        if verbose:
            _message("Harmony %d/%d" % (it, iter_harmony))
        # STEP 1: do clustering
        err_status = harmonyObj.cluster_cpp()
        if err_status == -1:
            raise KeyboardInterrupt("terminated by user")
        elif err_status != 0:
            raise RuntimeError("Harmony exited with non-zero exit status: %d" % err_status)
        # STEP 2: regress out covariates
        harmonyObj.moe_correct_ridge_cpp()
        # STEP 3: check for convergence
        if harmonyObj.check_convergence(1):
            if verbose:
                _message("Harmony converged after %d iterations" % it)
            return 0
    return None
