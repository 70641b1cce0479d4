#!/usr/bin/env python
"""DESIGN.md = docs/design_parts/*.md in file order (and README.md = docs/readme_template.md), with the @@KEY@@ placeholders filled from the committed evidence under profiles/<tag>_*
(python tools/assemble_design.py [tag], default r6; no GPU needed).  A key whose source file is missing is left as `n/a (file)` and reported."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
missing = []


def line(fn):
    try:
        return json.loads(open(os.path.join(P, fn)).read().strip().splitlines()[-1])
    except Exception:
        missing.append(fn)
        return None


def js(fn):
    try:
        return json.load(open(os.path.join(P, fn)))
    except Exception:
        missing.append(fn)
        return None


def fmt(x, nd=1):
    return ("%." + str(nd) + "f") % x


vals = {}
d = line(TAG + "_bench_default.json")
if d:
    also = d.get("also") or {}
    vals["DEF_MS"] = fmt(d["ms_per_step"], 2)
    vals["DEF_CPS"] = fmt(d["value"] / 1e6, 1)
    vals["STEP_US"] = fmt(d["roofline"]["avg_block_step_us"], 1)
    vals["FRAC"] = fmt(d["roofline"]["frac"], 3)
    vals["FRAC_NOM"] = fmt(d["roofline"]["nominal"]["frac"], 3)
    ph = d["config"]["gpu_phase_ms_per_step"]
    vals["PHASES_DEF"] = ", ".join("%s %s" % (k, fmt(v, 2)) for k, v in ph.items() if isinstance(v, (int, float)))
    ra = also.get("reference_arith")
    if ra:
        vals["REF_MS"] = fmt(ra["ms_per_step"], 1)
        vals["REF_CPS"] = fmt(ra["cells_per_s"] / 1e6, 1)
        vals["PHASES_REF"] = ", ".join("%s %s" % (k, fmt(v, 1)) for k, v in ra["gpu_phase_ms_per_step"].items())
    r2 = also.get("reference_arith_2")
    if r2 and "ms_per_step" in r2:
        vals["REF2_MS"] = fmt(r2["ms_per_step"], 1)
        vals["REF2_CPS"] = fmt(r2["cells_per_s"] / 1e6, 1)
    for key, leg in (("10M", "10M_one_gpu"), ("SHARE", "configs3_share_1p25M"), ("C5_1M", "c5_shape_1M"), ("PBMC", "pbmc30k")):
        if leg in also and "ms_per_step" in also[leg]:
            vals[key + "_MS"] = fmt(also[leg]["ms_per_step"], 1)
            if key == "10M":
                vals["10M_CPS"] = fmt(also[leg]["cells_per_s"] / 1e6, 0)
c5 = line(TAG + "_bench_c5_5M.json")
if c5:
    vals["C5_5M_MS"] = fmt(c5["ms_per_step"], 1)
for key, fn in (("C5_5M", TAG + "_parity_c5_5M.json"), ("C4_10M", TAG + "_parity_c4_10M.json")):
    t = js(fn)
    if t:
        pr = t["pairs"]
        ga, rf = pr["gpu_vs_oracle_accurate"], pr["gpu_ref_arith_vs_oracle_faithful"]
        row = "%.1e / %d clear flips · %.1e / %d" % (ga["Z_rel"], ga["argmax_diff_margin_ge_1e-5"], rf["Z_rel"], rf["argmax_diff_margin_ge_1e-5"])
        if "oracle_faithful_liberty1_vs_oracle_faithful" in pr:
            lb = pr["oracle_faithful_liberty1_vs_oracle_faithful"]
            row += " · %.1e / %d" % (lb["Z_rel"], lb["argmax_diff_margin_ge_1e-5"])
        row += " (GPU %.2f s / %.2f s, oracle %.0f s)" % (t["seconds"]["gpu"], t["seconds"]["gpu_ref_arith"], t["seconds"]["oracle_faithful"])
        vals[key + "_ROWS"] = row
        if key == "C5_5M":
            vals["C5_5M_REF"] = fmt(t["seconds"]["gpu_ref_arith"], 2)
        else:
            vals["10M_REF_S"] = fmt(t["seconds"]["gpu_ref_arith"], 2)
            vals["10M_Z"] = "%.1e" % ga["Z_rel"]
            vals["10M_REF_Z"] = "%.1e" % rf["Z_rel"]
try:
    tab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_all_kernels.py"), TAG], capture_output=True, text=True, timeout=120)
    if tab.returncode == 0:
        vals["ROOFLINE_TABLE"] = tab.stdout.strip()
    else:
        missing.append("roofline_all_kernels.py: " + tab.stderr.strip()[-200:])
except Exception as e:       # noqa: BLE001
    missing.append("roofline_all_kernels.py: %s" % e)

try:
    tab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_prediction.py"), TAG], capture_output=True, text=True, timeout=120)
    if tab.returncode == 0:
        vals["SCALING_TABLE"] = tab.stdout.strip()
    else:
        missing.append("scaling_prediction.py: " + (tab.stderr.strip() or tab.stdout.strip())[-200:])
except Exception as e:       # noqa: BLE001
    missing.append("scaling_prediction.py: %s" % e)

parts = sorted(f for f in os.listdir(os.path.join(ROOT, "docs", "design_parts")) if f.endswith(".md"))
out = io.StringIO()
for f in parts:
    out.write(open(os.path.join(ROOT, "docs", "design_parts", f)).read().rstrip("\n") + "\n\n")
text = out.getvalue()
import re
for k in sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", text))):
    if k in vals:
        text = text.replace("@@%s@@" % k, vals[k])
    else:
        text = text.replace("@@%s@@" % k, "n/a")
        missing.append("placeholder " + k)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text.rstrip("\n") + "\n")
# README.md from docs/readme_template.md: the same placeholders
rt = open(os.path.join(ROOT, "docs", "readme_template.md")).read()
for k in sorted(set(re.findall(r"@@([A-Z0-9_]+)@@", rt))):
    rt = rt.replace("@@%s@@" % k, vals.get(k, "n/a"))
    if k not in vals:
        missing.append("README placeholder " + k)
open(os.path.join(ROOT, "README.md"), "w").write(rt)
print("DESIGN.md: %d bytes from %d parts; %d placeholders filled" % (len(text), len(parts), len(vals)))
if missing:
    print("MISSING:", "; ".join(missing))
