#!/usr/bin/env python3
"""tools/summarize_issue_counters.py <tag> -- gpurun_out/<tag>/issue_* (tools/collect_issue_counters.sh) ->
profiles/<tag>_k67_issue.json: per rasterizer kernel the per-launch averages of the SQ issue counters, the kernel duration
of the same dispatches (from the --kernel-trace rows of the PMC runs: slower than an unprofiled run, quoted for the ratio
only) and two derived figures:
  valu_issue_frac = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES-equivalent SIMD cycles)  -- share of SIMD issue slots spent on VALU
  lanes_per_valu  = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 4 ... reported raw as thread_cycles / active cycles
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")


def short(name):
    for p in ("void ", "(anonymous namespace)::"):
        name = name.replace(p, "")
    return name.split("(")[0].split("<")[0].strip()


per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(src, "issue_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_"):
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_"):
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {}
for k, d in sorted(per.items()):
    row = {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}
    row["launches"] = max(len(v) for v in d.values())
    if dur.get(k):
        row["duration_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]), 2)
    wc, av, aa = row.get("SQ_WAVE_CYCLES"), row.get("SQ_ACTIVE_INST_VALU"), row.get("SQ_ACTIVE_INST_ANY")
    if wc and av is not None:
        row["valu_share_of_wave_cycles"] = round(av / wc, 4)
    if wc and row.get("SQ_WAIT_ANY") is not None:
        row["wait_any_share_of_wave_cycles"] = round(row["SQ_WAIT_ANY"] / wc, 4)
    if wc and row.get("SQ_WAIT_INST_ANY") is not None:
        row["wait_inst_share_of_wave_cycles"] = round(row["SQ_WAIT_INST_ANY"] / wc, 4)
    iv, tv = row.get("SQ_INSTS_VALU"), row.get("SQ_THREAD_CYCLES_VALU")
    if iv and av:
        row["quad_cycles_per_valu_inst"] = round(av / iv, 3)
    if tv and av:
        row["active_lanes_per_valu_cycle"] = round(tv / av / 4.0, 2)       # thread-cycles / (quad-cycles * 4) -> lanes of 64... see doc
    # issue bound: a SIMD issues one VALU wave-instruction per 4 cycles; 1024 SIMDs; clock from GRBM_GUI_ACTIVE if present
    if iv and row.get("duration_us_under_pmc"):
        row["valu_issue_frac_at_2p4GHz"] = round(iv * 4.0 / (1024 * 2.4e3 * row["duration_us_under_pmc"]), 4)
    out[k] = row
json.dump({"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                     "--no-train-step --no-view-streams` (tools/collect_issue_counters.sh); per-launch averages",
           "note": "valu_issue_frac_at_2p4GHz = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * duration * 2.4 GHz): the share of the chip's VALU "
                   "issue slots the kernel used (duration measured under the profiler)",
           "kernels": out}, open(os.path.join(dst, f"{tag}_k67_issue.json"), "w"), indent=1)
for k in ("k_composite_fwd", "k_composite_bwd", "k_block_masks", "k_tile_sort", "k_render_fwd", "k_render_bwd"):
    if k in out:
        print(k, json.dumps(out[k]))
