"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench_traffic.sh into profiles/r3_traffic.json.

Per launch of the dominant kernels (every 3x3 conv: conv_bx3p_kernel + conv_bx3_kernel, or conv_tap_kernel for --impl tap): HBM bytes read = FETCH_SIZE (KiB) x 1024 x 2
— MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B / lane) coalesced
reads at 64 B, `buffer_load ... lds` included, which is how this kernel reads everything; WRITE_SIZE (KiB) x 1024 is
taken as is (uncalibrated in the guide; it is 1.0-1.3x the algorithmic output here).  The algorithmic bytes of a
launch are input pixels x cin x 4 + M x cout x 4 + weights, from the per-op profile of the same workload."""
import csv
import glob
import json
import sys
from pathlib import Path


KERNEL = ("conv_tap_kernel",)


def per_kernel(pmc_dir, counter):
    files = glob.glob(f"{pmc_dir}/**/*counter_collection.csv", recursive=True)
    tot, disp = 0.0, set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k + "<" in r["Kernel_Name"] for k in KERNEL):
                tot += float(r["Counter_Value"])
                disp.add((f, r["Dispatch_Id"]))
    return tot, len(disp)


def main():
    global KERNEL
    wl, ops_csv, d_fetch, d_write = sys.argv[1:5]
    impl = sys.argv[5] if len(sys.argv) > 5 else "tap"
    # bx3: stride-1 layers run the patch kernel (conv_patch_bx3.hip), stride-2 layers the tap kernel
    KERNEL = {"tap": ("conv_tap_kernel",), "bx3": ("conv_bx3p_kernel", "conv_bx3_kernel"), "h2": ("conv_h2p_kernel", "conv_h2q_kernel", "conv_h2w_kernel", "conv_h2_kernel")}[impl]
    fetch_kib, n_f = per_kernel(d_fetch, "FETCH_SIZE")
    write_kib, n_w = per_kernel(d_write, "WRITE_SIZE")
    alg, n_ops = 0.0, 0
    for r in csv.DictReader(open(ops_csv)):
        if r["kind"] == "2" and r["ksize"] == "3":
            M, cout, cin, s = int(r["M"]), int(r["cout"]), int(r["cin"]), int(r["stride"])
            alg += M * s * s * cin * 4 + M * cout * 4 + 9 * cin * cout * (6 if impl == "bx3" else 4)      # weights: 3 x bf16 (bx3), 2 x fp16 (h2) or fp32
            n_ops += 1
    assert n_f and n_w and n_ops, (n_f, n_w, n_ops)
    fetch = fetch_kib * 1024 * 2 / n_f
    write = write_kib * 1024 / n_w
    out_path = Path("profiles/r3_traffic.json")
    doc = json.loads(out_path.read_text()) if out_path.exists() else {}
    doc[f"{wl}-{impl}"] = {
        "kernel": f"{' + '.join(KERNEL)} (all 3x3 convs of the workload)",
        "launches_counted": {"FETCH_SIZE": n_f, "WRITE_SIZE": n_w, "ops_per_step": n_ops},
        "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
        "bytes_per_launch": round(fetch + write), "algorithmic_bytes_per_launch": round(alg / n_ops),
        "ratio_to_algorithmic": round((fetch + write) / (alg / n_ops), 3),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over `bench.py --engine-only`; "
                  "FETCH_SIZE KiB x 1024 x 2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE KiB x 1024; "
                  "tools/pmc_bench_traffic.sh",
    }
    out_path.write_text(json.dumps(doc, indent=1))
    print(json.dumps(doc[f"{wl}-{impl}"], indent=1))


if __name__ == "__main__":
    main()
