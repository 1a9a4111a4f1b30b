#!/usr/bin/env python3
"""profiles/<tag>_summary.json (tools/summarize_profile.py) -> one entry of profiles/pmc_table.json, normalised PER FRAME so that
bench.py can scale it to any --steps:   tools/make_pmc_table.py <tag> <config> <W> <H> <rays> <frames per profiled launch>
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half the bytes of
wide reads, so the read side is doubled (an upper bound for this divergent 16-B pattern)."""
import json
import os
import sys

tag, config, W, H, rays, fpl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
suffix = sys.argv[7] if len(sys.argv) > 7 else ""            # "_philox" for the counter-based mode
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_sha16
s = json.load(open(os.path.join(root, "profiles", f"{tag}_summary.json")))
c, d = s["pmc_per_full_launch_mean"], s["derived"]
full = [k for k in s["kernel_stats"] if s["kernel_filter"] in k["Name"]]
ent = {
    "kernel": s.get("dispatch", {}).get("Kernel_Name", "").replace("void rtk::", "").split("(")[0],
    "frames_per_profiled_launch": fpl,
    "launch_ms_under_rocprof_max": round(max(float(k["MaxNs"]) for k in full) / 1e6, 3) if full else None,
    "valu_wave_instructions_per_frame": c["SQ_INSTS_VALU"] / fpl,
    "salu_wave_instructions_per_frame": c["SQ_INSTS_SALU"] / fpl,
    "vmem_read_wave_instructions_per_frame": c["SQ_INSTS_VMEM_RD"] / fpl,
    "lds_wave_instructions_per_frame": c["SQ_INSTS_LDS"] / fpl,
    "hbm_read_bytes_per_frame_FETCH_SIZE_x2": c["FETCH_SIZE"] * 2048 / fpl,
    "hbm_write_bytes_per_frame_WRITE_SIZE": c["WRITE_SIZE"] * 1024 / fpl,
    "hbm_bytes_per_frame": (c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024) / fpl,
    "l2_hit_rate": d.get("l2_hit_rate"), "l1_hit_rate": d.get("l1_hit_rate"),
    "valu_lane_utilisation": d.get("valu_lane_utilisation"), "ta_busy_frac": d.get("ta_busy_frac"), "td_busy_frac": d.get("td_busy_frac"),
    "wait_any_frac_of_wave_cycles": d.get("SQ_WAIT_ANY_frac_of_wave_cycles"),
    "wait_inst_any_frac_of_wave_cycles": d.get("SQ_WAIT_INST_ANY_frac_of_wave_cycles"),
    "valu_wave_instr_per_simd_cycle": d.get("valu_wave_instr_per_simd_cycle"), "vmem_issue_frac_per_cu": d.get("vmem_issue_frac_per_cu"),
    "csrc_sha16": csrc_sha16(),        # the kernels this pass ran: bench.py flags the entry as stale when the sources have changed since
    "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc, one counter group per pass; mean over the {fpl}-frame launches of each pass)",
}
# rays per frame of this workload (the bench line printed under the stats pass): lets bench.py scale the entry to another resolution or part of the image
if d.get("rays_per_frame"):
    ent["rays_per_frame"] = d["rays_per_frame"]
path = os.path.join(root, "profiles", "pmc_table.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table[f"config{config}_{W}x{H}_{rays}{suffix}"] = ent
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(ent, indent=1))
