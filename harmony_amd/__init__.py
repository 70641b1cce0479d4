"""harmony_amd -- MI355X-native implementation of Harmony's clustering + correction loop.

Host-side mirror of the reference's R API (RunHarmony / harmony_options / the `harmony` module
object) over the C ABI in include/harmony_mi355x.h.  All numerics run in hand-written HIP
(harmony_amd/csrc); there is no CPU fallback.
"""
from .harmony_obj import Harmony, HarmonyError
from .options import harmony_options
from .ui import RunHarmony, prepare_setup_args
from .utils import harmonize

__all__ = ["RunHarmony", "harmony_options", "Harmony", "HarmonyError", "harmonize", "prepare_setup_args"]
