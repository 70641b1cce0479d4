import os, sys, time, faulthandler
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harmony_amd import Harmony
t = time.time(); uid = Harmony.comm_unique_id(); print("unique id ok %.2fs" % (time.time() - t), flush=True)
g = Harmony(seed=1)
t = time.time(); g.comm_init(0, 1, uid); print("comm_init ok %.2fs" % (time.time() - t), flush=True)
import numpy as np
from bench_data import synth
from harmony_amd import prepare_setup_args
Z, meta, _ = synth(5000, d=20, levels=(3,), seed=1)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=10)
g.set_shard(0, 1, 0, 5000, None); g._set("comm_force", 1)
t = time.time(); g.setup(**skw); print("setup ok %.2fs" % (time.time() - t), flush=True)
t = time.time(); g.init_cluster_cpp(); print("init ok %.2fs" % (time.time() - t), flush=True)
t = time.time(); g.cluster_cpp(); print("cluster ok %.2fs calls=%d" % (time.time() - t, g._scalar("comm:calls")), flush=True)
