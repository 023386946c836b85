#!/bin/bash
# kernel + memory-copy trace of a short engine-only c3 run: how long the D2H result copies take and how much of a step the GPU idles
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; OUT=${1:-$R/gpurun_out/copy_trace}; WL=${2:-c3}
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o t -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --quick --engine-only --no-roofline --traffic none > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc=$?"
python - "$OUT" "$WL" <<'PY'
import csv, sys, glob
root = sys.argv[1]
k = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0]
c = glob.glob(f"{root}/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:40]))
if c:
    for r in csv.DictReader(open(c[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "?")) + " " + r.get("Bytes", r.get("Size", "?")) if True else ""))
ev.sort()
# the last two steps of the engine-only loop: from the end of the (2 T + 1)-th last nms_kernel to the end of the last one
# (T = trackers per step)
T = {"c3": 3, "c2": 2, "c4": 3}.get(sys.argv[2] if len(sys.argv) > 2 else "c3", 3)
nms = [e for e in ev if e[2] == "K" and "nms_kernel" in e[3]]
lo, hi = nms[-(2 * T + 1)][1], nms[-1][1]
sel = [e for e in ev if e[0] >= lo and e[1] <= hi]
busy = 0; cur_end = sel[0][0]; gaps = []
for s, e, kind, name in sel:
    if s > cur_end:
        gaps.append((s - cur_end, name, kind))
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
span = cur_end - sel[0][0]
print(f"two steps: window {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {(span-busy)/1e6:.2f} ms ({100*(span-busy)/span:.1f} %)")
gaps.sort(reverse=True)
for g, name, kind in gaps[:14]:
    print(f"  gap {g/1e3:8.1f} us before {kind} {name}")
cp = [e for e in sel if e[2] == "C"]
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for s, e, kind, name in cp:
    agg[name][0] += 1; agg[name][1] += e - s
for name, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:10]:
    print(f"  copy {name}: {n} x, {t/1e3/n:.1f} us each")
PY
find "$OUT" -name '*.csv' -delete; find "$OUT" -name '*.db' -delete 2>/dev/null
