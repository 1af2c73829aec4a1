#!/usr/bin/env python3
"""tools/summarize_issue_counters.py <tag> -- gpurun_out/<tag>/issue_* (tools/collect_issue_counters.sh) ->
profiles/<tag>_k67_issue.json: per rasterizer kernel the per-launch averages of the SQ issue counters, the kernel duration
of the same dispatches (from the --kernel-trace rows of the PMC runs: slower than an unprofiled run, quoted for the ratio
only) and derived figures.  `kernels` = the DEFAULT bench command (k_composite_bwd_views: all views of a step in one launch),
`kernels_one_view_per_launch` = the same workload with --no-view-streams.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, rocprofv3 PMC slots).

VALU pricing (tools/valu_rate.hip, profiles/<tag>_valu_rate.txt; MI355X_MICROARCH.md cycle-constants table): a wave64 VALU
instruction occupies a SIMD's vector pipe for VALU_CYC = 2 cycles once >= 2 waves are resident on the SIMD (4 cycles is what ONE
wave alone can issue)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.environ.get("CSPLAT_PROFILES_DST") or os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
VALU_CYC = 2.0
SIMDS, GHZ = 1024, 2.4


def short(name):
    for p in ("void ", "(anonymous namespace)::"):
        name = name.replace(p, "")
    return name.split("(")[0].split("<")[0].strip()


def collect(prefix, exclude=None):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(src, prefix + "*"))):
        if not os.path.isdir(d) or (exclude and os.path.basename(d).startswith(exclude)):
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
        wc, av = row.get("SQ_WAVE_CYCLES"), row.get("SQ_ACTIVE_INST_VALU")
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
            row["active_lanes_per_valu_cycle"] = round(tv / av / 4.0, 2)
        if wc and row.get("SQ_BUSY_CYCLES"):
            # average waves resident per SIMD while the kernel ran: wave quad-cycles * 4 / (SIMDs * duration in cycles)
            if row.get("duration_us_under_pmc"):
                row["avg_waves_per_simd"] = round(wc * 4.0 / (SIMDS * GHZ * 1e3 * row["duration_us_under_pmc"]), 2)
        if iv and row.get("duration_us_under_pmc"):
            row["valu_issue_frac_at_2p4GHz"] = round(iv * VALU_CYC / (SIMDS * GHZ * 1e3 * row["duration_us_under_pmc"]), 4)
        out[k] = row
    return out


batched = collect("issue_", exclude="issue_serial_")
serial = collect("issue_serial_")
sha = None
if os.path.exists(os.path.join(src, "raster_src_sha1.txt")):
    sha = open(os.path.join(src, "raster_src_sha1.txt")).read().strip()
json.dump({"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                     "--no-train-step` (`kernels`: the default command, one launch per stage for all views) and of the same with "
                     "--no-view-streams (`kernels_one_view_per_launch`) (tools/collect_issue_counters.sh); per-launch averages",
           "raster_src_sha1": sha,
           "note": f"valu_issue_frac_at_2p4GHz = SQ_INSTS_VALU * {VALU_CYC:g} cycles / (1024 SIMDs * duration * 2.4 GHz): the share of the "
                   "chip's vector-pipe cycles the kernel's VALU instructions occupied (duration measured under the profiler; "
                   "cycles per instruction: tools/valu_rate.hip)",
           "kernels": batched, "kernels_one_view_per_launch": serial},
          open(os.path.join(dst, f"{tag}_k67_issue.json"), "w"), indent=1)
for name, table in (("batched", batched), ("serial", serial)):
    for k in ("k_composite_fwd_views", "k_composite_bwd_views", "k_composite_fwd", "k_composite_bwd", "k_block_masks", "k_tile_sort"):
        if k in table:
            print(name, k, json.dumps(table[k]))
