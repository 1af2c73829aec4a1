#!/usr/bin/env python3
"""tools/summarize_profiles.py <tag> -- turn gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed
profiles/<tag>_* files: bench lines, rocprofv3 kernel stats, and the per-launch HBM traffic of every kernel from the
FETCH_SIZE / WRITE_SIZE PMC passes (gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts
64 B per 128-B request -> doubled; WRITE_SIZE is exact)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.environ.get("CSPLAT_PROFILES_DST") or os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    for p in ("void ", "(anonymous namespace)::"):
        name = name.replace(p, "")
    return name.split("(")[0].strip()


for f in ("bench.json", "bench_2rank_gloo.json", "bench_2rank_scenes.json", "valu_rate.txt", "bench_serial.json", "bench_gnn.json", "bench_train.json", "linear128.txt", "linear128_train.txt", "dw128.txt", "mfma_rate.txt", "gnn_train.txt", "edge_mlp3.txt"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        if f.endswith(".json"):      # keep the JSON line only (the gloo transport prints its own lines on stdout)
            lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(dst, f"{tag}_{f}"), "w").write(lines[-1] + "\n")
        else:
            shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))
for sub, out in (("trace", "kernel_stats.csv"), ("trace_serial", "kernel_stats_serial.csv"), ("trace_gnn", "gnn_kernel_stats.csv"),
                 ("trace_train", "train_kernel_stats.csv")):
    hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        rows = list(csv.reader(open(hits[0])))
        with open(os.path.join(dst, f"{tag}_{out}"), "w", newline="") as fh:
            csv.writer(fh).writerows(rows[:41])          # header + top 40 kernels


def pmc(sub_prefix):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(src, f"{sub_prefix}{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    per[short(r["Kernel_Name"])][c].append(float(r["Counter_Value"]))
    out = {}
    for k, d in per.items():
        if not (k.startswith("k_") or "k_" in k):
            continue
        fs, ws = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        fkb = sum(fs) / len(fs) if fs else 0.0
        wkb = sum(ws) / len(ws) if ws else 0.0
        out[k] = {"FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1), "launches": max(len(fs), len(ws)),
                  "hbm_bytes_per_launch": round((2.0 * fkb + wkb) * 1024.0)}
    return out


raster = pmc("pmc_")
raster = {k: v for k, v in raster.items() if "linear128" not in k}
serial = {k: v for k, v in pmc("pmcserial_").items() if "linear128" not in k}
sha = None
if os.path.exists(os.path.join(src, "raster_src_sha1.txt")):
    sha = open(os.path.join(src, "raster_src_sha1.txt")).read().strip()
if raster:
    k7 = next((v for k, v in raster.items() if k.startswith("k_composite_bwd_views")), None) or \
        next((v for k, v in raster.items() if k.startswith("k_composite_bwd") or k.startswith("k_render_bwd")), None)
    k6 = next((v for k, v in raster.items() if k.startswith("k_composite_fwd_views")), None)
    doc = {"kernel": "k_composite_bwd_views (K7, all views of a step in ONE launch -- the launch bench.py times)",
           "raster_src_sha1": sha,
           "k6_hbm_bytes_per_launch": k6["hbm_bytes_per_launch"] if k6 else None,
           "one_view_per_launch": serial or None,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python3 bench.py "
                     "--steps 2 --warmup 1 --no-cpu-baseline --no-train-step` (the default command; `one_view_per_launch`: the same "
                     "with --no-view-streams) (tools/collect_profiles.sh)",
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE tallies 64 B per 128-B request on gfx950 -> doubled; "
                         "WRITE_SIZE exact for float atomics. K7's reads are 4-16 B/lane gathers (width not calibrated), "
                         "so the doubled figure is an upper estimate",
           "FETCH_SIZE_KB_per_launch": k7["FETCH_SIZE_KB"] if k7 else None,
           "WRITE_SIZE_KB_per_launch": k7["WRITE_SIZE_KB"] if k7 else None,
           "launches": k7["launches"] if k7 else 0,
           "hbm_bytes_per_launch": k7["hbm_bytes_per_launch"] if k7 else None,
           "all_kernels": raster}
    json.dump(doc, open(os.path.join(dst, f"{tag}_k7_pmc_traffic.json"), "w"), indent=1)
l128 = {k: v for k, v in pmc("pmc_l128_").items() if "linear128" in k}
if l128:
    json.dump({"source": "rocprofv3 --pmc passes of tools/bench_linear128.py 300000 2 (M = 300,000 rows: 153.6 MB in, "
                         "153.6 MB out algorithmic)", "kernels": l128},
              open(os.path.join(dst, f"{tag}_linear128_pmc_traffic.json"), "w"), indent=1)
import subprocess
subprocess.call([sys.executable, os.path.join(root, "tools", "summarize_issue_counters.py"), tag])
print("profiles/ now holds:", sorted(os.listdir(dst)))
