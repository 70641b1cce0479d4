"""torch-first process: the library's built-in communicator bound to the RCCL that torch already loaded."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(70, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
x = torch.ones(4, device="cuda"); torch.cuda.synchronize()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.all_reduce(x); torch.cuda.synchronize()          # makes torch load and initialise its librccl
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
t = time.time(); uid = Harmony.comm_unique_id(); print("unique id ok %.2fs" % (time.time() - t), flush=True)
Z, meta, _ = synth(20000, d=50, levels=(10,), seed=33)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
outs = []
for use in (False, True):
    g = Harmony(device=0, seed=5)
    if use:
        t = time.time(); g.comm_init(0, 1, uid); print("comm_init ok %.2fs" % (time.time() - t), flush=True)
        g.set_shard(0, 1, 0, Z.shape[0], None); g._set("comm_force", 1)
    g.setup(**skw); g.init_cluster_cpp(); assert g.cluster_cpp() == 0; g.moe_correct_ridge_cpp()
    outs.append((g.getZcorr(), g.O, g._scalar("comm:calls")))
assert outs[1][2] > 80 and np.array_equal(outs[0][1], outs[1][1])
assert np.linalg.norm(outs[0][0] - outs[1][0]) / np.linalg.norm(outs[0][0]) < 1e-6
print("RCCL_PROBE2_OK calls=%d" % outs[1][2], flush=True)
