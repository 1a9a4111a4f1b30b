#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_summary.json + .md.
HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced read, so the read side is given both raw and doubled (upper bound)."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
kernel_filter = sys.argv[2] if len(sys.argv) > 2 else "k_stream"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
out = {"tag": tag, "kernel_filter": kernel_filter}


def newest(pattern):
    """one file per pass directory: the newest (gpurun MERGES a call's output into gpurun_out/, so a pass that was run twice has the
    files of both runs there — averaging them would mix two builds)"""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows]
for f in newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if kernel_filter in r["Kernel_Name"]:
            out["dispatch"] = {k: r[k] for k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                  "Workgroup_Size_X", "Grid_Size_X")}
            break

counters = {}
for f in newest(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kernel_filter not in r["Kernel_Name"]:
            continue
        counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out["pmc_per_launch_mean"] = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
out["pmc_launches"] = {k: len(v) for k, v in counters.items()}
# the bench's launches differ in size (calibration frames, then multi-frame launches): "full" = the launches whose value is at
# least half the largest one, i.e. the multi-frame launches of the warm-up and of the timed region
full = {k: [x for x in v if x >= 0.5 * max(v)] for k, v in counters.items() if v}
out["pmc_per_full_launch_mean"] = {k: sum(v) / len(v) for k, v in sorted(full.items())}
out["pmc_full_launches"] = {k: len(v) for k, v in full.items()}
c = out["pmc_per_full_launch_mean"]
d = {}
if "FETCH_SIZE" in c:
    d["hbm_read_bytes_raw"] = c["FETCH_SIZE"] * 1024
    d["hbm_read_bytes_x2_gfx950_correction"] = c["FETCH_SIZE"] * 2048
if "WRITE_SIZE" in c:
    d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in c:
            d[k + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"] * 64, 1)
if "GRBM_GUI_ACTIVE" in c:
    # SQ_* / TA_* sums are per shader engine (32 per chip on MI355X), GRBM_GUI_ACTIVE is summed over the 8 XCDs: /32 gives the
    # per-SIMD (resp. per-TA) busy fraction of the launch, as in profiles/hbm_traffic.json
    if "SQ_ACTIVE_INST_VALU" in c:
        # SQ_ACTIVE_INST_VALU sums, over the WAVES of a SIMD, the (quad-)cycles in which the wave has a VALU instruction in flight: with
        # several waves per SIMD it exceeds the SIMD's own cycles (up to waves-per-SIMD times) — it is not a busy fraction
        d["valu_active_wave_cycles_per_simd_cycle"] = c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] * 32)
    if "SQ_INSTS_VALU" in c:
        # VALU wave-instructions issued per SIMD cycle (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs): spec peak 0.5
        d["valu_wave_instr_per_simd_cycle"] = c["SQ_INSTS_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_INSTS_VMEM_RD" in c:
        # vector-memory read instructions per CU cycle x 16.2 cycles each (tools/ubench/vmem_rate.hip): share of the CU's vector-memory issue
        d["vmem_issue_frac_per_cu"] = c["SQ_INSTS_VMEM_RD"] * 16.2 / (c["GRBM_GUI_ACTIVE"] / 8 * 256)
    if "TA_TA_BUSY_sum" in c:
        d["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (c["GRBM_GUI_ACTIVE"] * 32)
    if "TD_TD_BUSY_sum" in c:
        d["td_busy_frac"] = c["TD_TD_BUSY_sum"] / (c["GRBM_GUI_ACTIVE"] * 32)
if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
    d["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1)
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    d["hbm_bytes_per_full_launch"] = c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024
log = os.path.join(src, "stats.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            b = json.loads(line)
            out["bench_line_under_rocprof"] = {k: b[k] for k in ("value", "ms_per_step", "steps", "config")}
            d["rays_per_frame"] = b["config"]["rays_per_frame"]
out["derived"] = d
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` (tools/profile.sh; kernel filter `{kernel_filter}`)\n\n## --kernel-trace --stats\n\n")
    f.write("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
    for r in out.get("kernel_stats", []):
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.3f} |\n")
    f.write(f"\ndispatch: `{json.dumps(out.get('dispatch', {}))}`\n\n## --pmc (separate passes; mean per launch of the filtered kernel)\n\n| counter | value |\n|---|---|\n")
    for k, v in c.items():
        f.write(f"| {k} | {v:.6g} |\n")
    f.write("\n## derived\n\n| quantity | value |\n|---|---|\n")
    for k, v in d.items():
        f.write(f"| {k} | {v:.6g} |\n")
print(json.dumps(out["derived"], indent=1)); print(json.dumps(out["pmc_per_launch_mean"], indent=1)); print(out.get("dispatch"))
