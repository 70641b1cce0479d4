"""Dump the reference's bundled fixtures to small .npz files under tests/golden/.

Run in the build container only (it reads /root/reference/data, which does not
exist on the GPU box):   python tools/make_golden.py

Outputs (all inputs of the path, not outputs -- the reference ships no golden
numeric outputs):
  cell_lines_small.npz  300 cells x 20 PCs, `dataset` (3 levels), `cell_type` (2)
  cell_lines.npz        2370 cells x 20 PCs, same covariates
  pbmc_stim_pcs.npz     the 2 x 1000 shipped PBMC cells -> log-normalise, scale,
                        truncated SVD to 50 PCs (PCA is upstream of the path);
                        covariate `stim` (ctrl/stim)
Factor levels follow R's as.factor(): sorted unique strings (R/ui.R:210-213).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rdata_reader import data_frame_columns, read_rdata  # noqa: E402

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def factor(strings):
    levels = sorted(set(strings))
    lut = {s: i for i, s in enumerate(levels)}
    return np.array([lut[s] for s in strings], dtype=np.int32), np.array(levels)


def cell_lines(fname, key, out):
    top = read_rdata(os.path.join(REF, fname))[key]
    parts = dict(zip(top["attr"]["names"]["value"], top["value"]))
    meta = data_frame_columns(parts["meta_data"])
    pcs = data_frame_columns(parts["scaled_pcs"])
    Z = np.stack([pcs["X%d" % (j + 1)] for j in range(len(pcs))], axis=1)  # N x d
    ds, ds_lv = factor(meta["dataset"])
    ct, ct_lv = factor(meta["cell_type"])
    np.savez_compressed(os.path.join(OUT, out), pcs=Z, dataset=ds, dataset_levels=ds_lv,
                        cell_type=ct, cell_type_levels=ct_lv)
    print(out, Z.shape, np.bincount(ds), np.bincount(ct))


def pbmc(out):
    import scipy.sparse as sp
    top = read_rdata(os.path.join(REF, "pbmc_stim.RData"))
    mats = []
    for name in ("pbmc.ctrl", "pbmc.stim"):
        a = top[name]["attr"]
        dim = a["Dim"]["value"]
        m = sp.csc_matrix((a["x"]["value"], a["i"]["value"], a["p"]["value"]), shape=tuple(dim))
        mats.append(m)
    X = sp.hstack(mats).tocsc().astype(np.float64)  # genes x cells
    stim = np.repeat(np.array([0, 1], dtype=np.int32), [mats[0].shape[1], mats[1].shape[1]])
    tot = np.asarray(X.sum(axis=0)).ravel()
    X = X @ sp.diags(1e4 / tot)
    X.data = np.log1p(X.data)
    Xd = X.toarray()
    var = Xd.var(axis=1)
    top_genes = np.argsort(-var, kind="stable")[:2000]
    Xs = Xd[top_genes]
    Xs = (Xs - Xs.mean(axis=1, keepdims=True)) / (Xs.std(axis=1, ddof=1, keepdims=True) + 1e-12)
    Xs = np.clip(Xs, -10, 10)
    u, s, vt = np.linalg.svd(Xs, full_matrices=False)
    pcs = (vt[:50].T * s[:50])  # cells x 50
    np.savez_compressed(os.path.join(OUT, out), pcs=pcs.astype(np.float32), stim=stim,
                        stim_levels=np.array(["ctrl", "stim"]))
    print(out, pcs.shape, np.bincount(stim))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    cell_lines("cell_lines_small.RData", "cell_lines_small", "cell_lines_small.npz")
    cell_lines("cell_lines.rda", "cell_lines", "cell_lines.npz")
    pbmc("pbmc_stim_pcs.npz")
