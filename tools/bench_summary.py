#!/usr/bin/env python
"""One-screen summary of a bench.py JSON line (tuning tool)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(f"value {d['value']} {d['unit']}  ms/step {d['ms_per_step']}  engine-only {d.get('engine_only', {}).get('value')}"
      f" ({d.get('engine_only', {}).get('ms_per_step')} ms)")
if r:
    print(f"3x3: {r['achieved']} TFLOP/s frac {r['frac']} ({r['kernel_ms_per_step']} ms, {r['launches']} launches)"
          f"  1x1: {r['conv1x1']['achieved']} ({r['conv1x1']['ms_per_step']} ms)  all kernels {r['all_kernels_ms_per_step']} ms"
          f"  other {r['other_ms_per_step']}")
for k in ("host_frames", "reference_default", "cpu_baseline"):
    if k in d:
        print(k, {kk: vv for kk, vv in d[k].items() if kk not in ("what", "trackers", "sample")})
p = d.get("parity")
if p:
    print("parity", {k: p[k] for k in ("linf_px_vs_fp32_oracle", "linf_px_vs_fp64", "oracle_floor_px", "classes_equal", "detection_sets_equal", "detections")})
    if "low_noise_heads" in p:
        print("low-noise heads", {k: v for k, v in p["low_noise_heads"].items() if k not in ("what", "per_tracker")})
