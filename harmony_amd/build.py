"""Build libharmony_mi355x.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m harmony_amd.build            # or: from harmony_amd.build import build; build()

The library is written to harmony_amd/lib/ so that it travels with the repo snapshot to
the GPU box (a JIT cache under ~/.cache would not).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("hmx_kernels.hip", "hmx_tile_bf.hip", "hmx_seq.hip", "hmx_api.cpp")]
# (hmx_tile_bf.hip is hmx_kernels.hip's tile kernel built a second time with the split-bf16 distance GEMM: it includes that file)
_INC = [os.path.join(HERE, "csrc", f) for f in ("hmx_k_stream.inc", "hmx_k_tile.inc", "hmx_k_correct.inc", "hmx_k_launch.inc")]      # the kernels, by section (included by hmx_kernels.hip)
_API_INC = [os.path.join(HERE, "csrc", "hmx_api_%s.inc" % f) for f in ("seam", "kmeans", "refarith", "update", "ridge", "p2p", "setup", "diag")]    # the host orchestration, by section (included by hmx_api.cpp)
EXTRA_DEP = {"hmx_kernels.hip": _INC, "hmx_tile_bf.hip": [os.path.join(HERE, "csrc", "hmx_kernels.hip")] + _INC, "hmx_api.cpp": _API_INC}
HDR = [os.path.join(HERE, "csrc", "hmx_internal.h"), os.path.join(HERE, "csrc", "hmx_rrng.h"), os.path.join(HERE, "..", "include", "harmony_mi355x.h"), os.path.join(HERE, "..", "include", "harmony_mi355x_lab.h")]
OUT = os.path.join(HERE, "lib", "libharmony_mi355x.so")


def _torch_lib():
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
            if os.path.exists(os.path.join(d, "libamdhip64.so")):
                return d
    except Exception:
        pass
    return None


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SRC + HDR + _INC + _API_INC)


def build(force=False, verbose=True, trace=False):
    """trace=True builds the diagnostics variant (-DHMX_TRACE: per-wave phase stamps in the block-update kernel,
    see tools/trace_update.py) next to the product library; it is never loaded unless HMX_LIB_PATH points at it."""
    if trace:
        out = OUT.replace(".so", "_trace.so")
    else:
        out = OUT
    if not trace and not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # one object per source (kept under lib/obj, rebuilt only when the source or a header is newer), compiled concurrently, then linked
    objdir = os.path.join(os.path.dirname(OUT), "obj_trace" if trace else "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result"] + (["-DHMX_TRACE"] if trace else [])
    hdr_t = max(os.path.getmtime(p) for p in HDR)
    procs, objs = [], []
    for src in SRC:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        dep_t = max([os.path.getmtime(src), hdr_t] + [os.path.getmtime(q) for q in EXTRA_DEP.get(os.path.basename(src), [])])
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < dep_t:
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs + ["-lpthread", "-ldl"]
    # Share ONE HIP runtime with PyTorch when both live in a process: torch wheels bundle their own
    # libamdhip64.so (no SONAME).  Linking against that file records DT_NEEDED "libamdhip64.so": if torch is
    # already imported the loader reuses torch's runtime (device pointers, streams and RCCL then interoperate),
    # otherwise the name resolves through RUNPATH to /opt/rocm's runtime (R / C hosts without torch).
    tl = _torch_lib()
    if tl:
        cmd += ["-L" + tl]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, trace="--trace" in sys.argv)
