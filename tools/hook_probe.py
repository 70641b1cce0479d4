"""torch-first process: torch.distributed (nccl == RCCL) all-reduce hook on the library's device buffers, with a
1-rank process group and forced collectives; must reproduce the plain single-handle run.  Prints HOOK_PROBE_OK."""
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(150, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

from bench_data import synth  # noqa: E402
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402  (loaded AFTER torch: shares torch's HIP runtime)
from harmony_amd.dist import TorchAllReduce  # noqa: E402

Z, meta, _ = synth(20000, d=50, levels=(10,), seed=33)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
outs = []
for use in (False, True):
    g = Harmony(device=0, seed=5)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    hook = None
    if use:
        hook = TorchAllReduce(device=dev)
        g.set_shard(0, 1, 0, Z.shape[0], hook)
        g._set("comm_force", 1)
    g.setup(**skw)
    g.init_cluster_cpp()
    assert g.cluster_cpp() == 0
    g.moe_correct_ridge_cpp()
    torch.cuda.synchronize()
    outs.append((g.getZcorr(), g.O, g.objective_kmeans, hook.calls if hook else 0))
assert outs[1][3] > 80, outs[1][3]
assert np.array_equal(outs[0][1], outs[1][1])
assert np.allclose(outs[0][2], outs[1][2], rtol=1e-6)
assert np.linalg.norm(outs[0][0] - outs[1][0]) / np.linalg.norm(outs[0][0]) < 1e-6
dist.destroy_process_group()
print("HOOK_PROBE_OK calls=%d" % outs[1][3], flush=True)
