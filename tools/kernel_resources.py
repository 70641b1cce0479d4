"""Static resource table of every gfx950 kernel in the build: VGPRs, spilled VGPRs, scratch, SGPRs, static LDS, workgroup size -- read
from the code objects' own metadata (llvm-readelf --notes) -- plus the MFMA instruction counts of each object's disassembly.  Needs no GPU.

    python tools/kernel_resources.py                 # table on stdout
    python tools/kernel_resources.py --write          # also profiles/r6_kernel_resources.txt

tests/test_kernel_resources.py pins the figures DESIGN.md quotes (registers / spills of the chain kernels, MFMA flavour of the distance
GEMM) to the objects of the build, so that the document cannot drift from the code again.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDIR = os.path.join(ROOT, "harmony_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
OBJECTS = ("hmx_kernels", "hmx_tile_bf", "hmx_seq")
FIELDS = (".vgpr_count", ".vgpr_spill_count", ".agpr_count", ".sgpr_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".max_flat_workgroup_size")


def available():
    return all(os.path.exists(os.path.join(OBJDIR, o + ".o")) for o in OBJECTS) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))


def code_object(obj, workdir):
    """the gfx950 code object bundled in harmony_amd/lib/obj/<obj>.o"""
    fat, co = os.path.join(workdir, obj + ".fat"), os.path.join(workdir, obj + ".co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(OBJDIR, obj + ".o"), fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("hmx::", "").replace("(hmx::Dev, int)", "").replace("void ", "") for o in out[:len(names)]]


def kernels(obj, workdir=None):
    """{demangled kernel name: {field: int}} of one object"""
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj, workdir or tmp)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    res, cur = [], None
    for line in notes.split("\n"):
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == ".agpr_count":                # first key of a kernel's record (keys are sorted)
            cur = {}
            res.append(cur)
        if cur is None:
            continue
        if k == ".name":
            cur["name"] = v.strip("'\"")
        elif k in FIELDS:
            cur[k] = int(v)
    res = [r for r in res if "name" in r]
    for r, n in zip(res, demangle([r["name"] for r in res])):
        r["kernel"] = re.sub(r"\(.*$", "", n).strip()
    return {r["kernel"]: r for r in res}


def mfma_counts(obj):
    """{mfma mnemonic: occurrences} in the object's gfx950 disassembly"""
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(obj, tmp)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
    cnt = {}
    for m in re.finditer(r"\b(v_mfma_[a-z0-9_]+)", dis):
        cnt[m.group(1)] = cnt.get(m.group(1), 0) + 1
    return cnt


def table():
    lines = ["# static resources of every gfx950 kernel of the build (llvm-readelf --notes of harmony_amd/lib/obj/*.o; tools/kernel_resources.py)",
             "# object            vgpr spill agpr sgpr scratchB  ldsB  wgmax  kernel"]
    for o in OBJECTS:
        ks = kernels(o)
        for name in sorted(ks):
            r = ks[name]
            lines.append("%-18s %5d %5d %4d %4d %8d %5d %6d  %s" % (o, r.get(".vgpr_count", -1), r.get(".vgpr_spill_count", -1), r.get(".agpr_count", -1),
                                                                    r.get(".sgpr_count", -1), r.get(".private_segment_fixed_size", -1),
                                                                    r.get(".group_segment_fixed_size", -1), r.get(".max_flat_workgroup_size", -1), name))
    lines.append("# MFMA instructions per object (llvm-objdump -d)")
    for o in OBJECTS:
        c = mfma_counts(o)
        lines.append("%-18s %s" % (o, ", ".join("%s x %d" % kv for kv in sorted(c.items())) or "none"))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    if not available():
        sys.exit("harmony_amd/lib/obj/*.o absent: run `python -m harmony_amd.build --force` first")
    t = table()
    sys.stdout.write(t)
    if "--write" in sys.argv:
        open(os.path.join(ROOT, "profiles", "r6_kernel_resources.txt"), "w").write(t)
