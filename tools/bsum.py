#!/usr/bin/env python3
"""one line per bench JSON: value, ms/step, latencies, lane utilisation"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d.get("roofline") or {}
    u = r.get("phase_lane_utilisation", {})
    e = r.get("phase_wave_execs_per_ray", {})
    print(f"{f.split('/')[-1]:42s} {d['value']:9.1f} Mrays/s {d['ms_per_step']:7.3f} ms  lat {d.get('latency_ms_single_frame')} / {d.get('latency_ms_single_frame_moving_camera')}  "
          f"util n{u.get('node')} t{u.get('triangle')} s{u.get('shade')} e{u.get('environment')} c{u.get('camera')}  execs n{e.get('node')} t{e.get('triangle')}")
