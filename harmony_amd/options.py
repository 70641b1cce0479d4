"""harmony_options() and legacy-argument handling, mirroring R/harmony_option.R."""


class HarmonyOptions(dict):
    """class 'harmony_options' (R/harmony_option.R:33-55)."""


def _validate_block_size(block_size):  # R/harmony_option.R:58-63
    if block_size <= 0 or block_size > 1:
        raise ValueError("Error: block.size should be set between 0 and 1 (0 < block.size <= 1)")
    return block_size


def harmony_options(alpha=0.2, tau=0, block_size=0.05, max_iter_cluster=4, epsilon_cluster=1e-3,
                    epsilon_harmony=1e-2, batch_prop_cutoff=1e-5):
    """Same defaults as the reference (R/harmony_option.R:33-40); dots became underscores."""
    block_size = _validate_block_size(block_size)
    return HarmonyOptions(alpha=alpha, tau=tau, block_size=block_size, max_iter_cluster=max_iter_cluster,
                          epsilon_cluster=epsilon_cluster, epsilon_harmony=epsilon_harmony,
                          batch_prop_cutoff=batch_prop_cutoff)


_LEGACY = {  # R/harmony_option.R:67-81: these arguments hard-error since v1.0
    "do_pca": "do_pca", "npcs": "npcs", "tau": "tau", "block.size": "block.size", "block_size": "block.size",
    "max.iter.harmony": "max.iter.harmony", "max_iter_harmony": "max.iter.harmony",
    "max.iter.cluster": "max.iter.cluster", "max_iter_cluster": "max.iter.cluster",
    "epsilon.cluster": "epsilon.cluster", "epsilon_cluster": "epsilon.cluster",
    "epsilon.harmony": "epsilon.harmony", "epsilon_harmony": "epsilon.harmony",
}


def check_legacy_args(**kwargs):
    for k in kwargs:
        if k in _LEGACY:
            raise TypeError("Error: Argument %s is deprecated and moved to harmony_options(); "
                            "pass it through options=harmony_options(...)" % _LEGACY[k])
        raise TypeError("RunHarmony got an unexpected argument %r" % k)
