#!/usr/bin/env python
"""How fast do the restarted sequential sums converge when the fp32 accumulator SATURATES (terms below half an ulp of the running sum, the
regime of my_accu at 10M cells)?  Three chains of 2e7 terms; passes 3 .. 48 against numpy's one-after-the-other float32 accumulate."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harmony_amd import _lib  # noqa: E402

rng = np.random.default_rng(3)
n = 20_000_000
a = (rng.random(n, dtype=np.float32) * 0.9).astype(np.float32)                 # sum ~9e6 > 2^23: the accumulator's ulp reaches 1 -> most terms are dropped
b = (rng.random(n, dtype=np.float32) ** 4).astype(np.float32)                  # heavy share of tiny terms: dropped long before
c = (rng.random(n, dtype=np.float32) * 0.05).astype(np.float32)                # sum ~5e5: mild regime (ulp 0.03 against terms <= 0.05)
T = np.ascontiguousarray(np.stack([a, b, c]), dtype=np.float32)
want = np.array([np.add.accumulate(T[i], dtype=np.float32)[-1] for i in range(3)], np.float32)
exact = T.astype(np.float64).sum(axis=1)
print("sequential fp32:", want, " exact:", exact, " bias:", (want - exact) / exact)
for passes in (3, 4, 6, 8, 12, 16, 24, 32, 48):
    tot = np.empty(3, np.float32)
    mm, res = C.c_int64(-1), C.c_double(-1)
    st = _lib.load().hmx_debug_seq_arr(T.ctypes.data_as(C.POINTER(C.c_float)), n, 3, 4096, passes, tot.ctypes.data_as(C.POINTER(C.c_float)),
                                       C.byref(mm), C.byref(res))
    assert st == 0
    print("passes %2d: rel. error vs sequential %s, segment starts still moving %d, largest relative move %.2e"
          % (passes, np.abs(tot.astype(np.float64) - want) / np.abs(want), mm.value, res.value))
