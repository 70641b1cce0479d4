"""All-reduce hook for multi-GPU runs: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for CPU tests).  The C library calls the hook between kernel launches with a raw
buffer pointer; the hook wraps it zero-copy and issues ONE collective.  Cells are sharded contiguously
(shard_bounds); every cross-cell quantity of the algorithm is a small sum, so these all-reduces are the
only communication (DESIGN.md "Multi-GPU").
"""
import ctypes

import numpy as np

_NP = {0: np.int64, 1: np.float64, 2: np.int64}


def shard_bounds(N, world):
    """Contiguous cell ranges [lo, hi) per rank, sizes differing by at most one."""
    base, rem = divmod(int(N), int(world))
    lo = [r * base + min(r, rem) for r in range(world)]
    return [(lo[r], lo[r] + base + (1 if r < rem else 0)) for r in range(world)]


class _DevPtr(object):
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}


class TorchAllReduce(object):
    """Callable matching hmx_allreduce_fn(user, buf, count, dtype, stream) -> int."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device = device  # None => host buffers (gloo); else a torch.device("cuda", i)
        self.calls = 0
        self.bytes = 0
        self._streams = {}

    def wrap(self, buf, count, dtype):
        if self.device is None:
            ct = ctypes.c_int64 if dtype in (0, 2) else ctypes.c_double
            arr = np.ctypeslib.as_array((ct * int(count)).from_address(int(buf)))
            return self.torch.from_numpy(arr)
        return self.torch.as_tensor(_DevPtr(buf, count, "<i8" if dtype in (0, 2) else "<f8"), device=self.device)

    def __call__(self, user, buf, count, dtype, stream):
        try:
            t = self.wrap(buf, count, dtype)
            op = self.dist.ReduceOp.MIN if dtype == 2 else self.dist.ReduceOp.SUM
            if self.device is not None and stream:
                # order the collective on the LIBRARY's stream: RCCL waits for the kernels queued there and the
                # library's next kernels wait for the collective -- no host synchronisation
                ext = self._streams.get(stream)
                if ext is None:
                    ext = self._streams[stream] = self.torch.cuda.ExternalStream(int(stream), device=self.device)
                with self.torch.cuda.stream(ext):
                    self.dist.all_reduce(t, op=op, group=self.group)
            else:
                self.dist.all_reduce(t, op=op, group=self.group)
            self.calls += 1
            self.bytes += int(count) * 8
            return 0
        except Exception as e:  # never let an exception cross the C ABI
            import sys
            print("harmony_amd.dist: all-reduce failed: %r" % (e,), file=sys.stderr)
            return 1
