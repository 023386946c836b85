#!/usr/bin/env python
"""Per-wave s_memtime timeline of an instrumented conv kernel (tuning tool, GPU only).

--kernel h2q: the quad patch kernel of h2 graphs (conv_patch_h2q.hip, tile 323; needs a library built with
-DPADEL_H2P_PROBES); --kernel tap: the fp32-MFMA tap kernel, 64x96 tile.


Runs the yolov8m P4 bottleneck conv (192->192 3x3, batch 64) through the DIAG-16 instantiation of
conv_lds_kernel (every wave stamps 5 points of every k-step: loop top / fragments in registers / last MFMA issued /
prefetch landed / after store+barrier), reads the dump (PADEL_CONV_DBG) and prints where a k-step's cycles go,
chip-wide and as an ASCII timeline of the waves that shared one SIMD.

    python tools/timeline_probe.py [--out gpurun_out/timeline.txt]
"""
import argparse, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

STEPS, WORDS = 64, 8 + 4 * 64 * 5


def run_conv(dump, kernel="lds", B=64, H=48, W=80, cin=192, cout=192, tune=1):
    w16 = kernel == "h2r"
    if kernel == "h2r":
        kernel = "h2q"
    if kernel not in ("tap", "h2q"):
        raise SystemExit("the LDS kernel's DIAG instantiations were retired with round 2 (tools/legacy_conv/); use --kernel tap")
    from padel_analytics_amd import engine as E, graph as G
    eng = E.default_engine(0)
    eng.set_profiling(True)
    h2 = kernel == "h2q"
    eng.set_tuning(impl=0, variant=(324 if w16 else 323) if h2 else 7, timeline=1, tune=tune)
    eng.lib.pa_engine_set_timeline_path(eng.handle, dump.encode())
    rng = np.random.default_rng(0)
    g = G.Graph(task=G.TASK_TRACKNET, dtype=G.DTYPE_H2 if h2 else G.DTYPE_F32)
    b0 = g.buf(0, cin)
    b1 = g.buf(0, G.pad16(cout))
    w = rng.normal(0, (2.0 / (cin * 9)) ** 0.5, (cout, cin, 3, 3)).astype(np.float32)
    if w16:
        w = w.astype(np.float16).astype(np.float32)          # fp16 numbers: PA_CONV_W_SINGLE, the two-product kernels
    g.conv((b0, 0, cin), (b1, 0), w, np.zeros(cout, np.float32), 3, 1, 1, out_width=G.pad16(cout) if h2 else None)
    if h2:                             # the measured conv writes pairs; a tiny fp32 head keeps pa_tracknet_infer's contract
        hd = g.buf(0, 16)
        g.conv((b1, 0, G.pad16(cout)), (hd, 0), np.zeros((1, G.pad16(cout), 1, 1), np.float32), np.zeros(1, np.float32), 1, 1, 0)
        g.head_buf = (hd, -1, -1)
    else:
        g.head_buf = (b1, -1, -1)
    m = E.Model(eng, g)
    m.set_max_batch(B)
    x = rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)
    for _ in range(2):
        m.tracknet_infer(x)
    ms = [p for p in m.profile_rows() if p["kind"] == 2][0]["ms"]
    m.close()
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/timeline.txt")
    ap.add_argument("--dump", default="/tmp/padel_conv_timeline.bin")
    ap.add_argument("--kernel", default="tap", choices=["tap", "h2q", "h2r"],
                    help="tap = conv_tap_kernel (v5) timeline instantiation; h2q = conv_h2q_kernel (h2 quad patch kernel)")
    ap.add_argument("--cin", type=int, default=192)
    ap.add_argument("--cout", type=int, default=192)
    ap.add_argument("--hw", default="48x80")
    ap.add_argument("--tune", type=int, default=1, help="tuning word of the launch (h2r: 9 = no phase offset)")
    a = ap.parse_args()
    global STEPS, WORDS
    h2r = a.kernel == "h2r"
    h2q = a.kernel in ("h2q", "h2r")
    if h2q:
        STEPS, WORDS = 32, 8 + 4 * 32 * 5            # conv_patch_h2q.hip:kQDbgSteps
    H_, W_ = (int(v) for v in a.hw.split("x"))
    ms = run_conv(a.dump, a.kernel, H=H_, W=W_, cin=a.cin, cout=a.cout, tune=a.tune)
    raw = np.fromfile(a.dump, dtype=np.uint64)
    nblk = raw.size // WORDS
    raw = raw[: nblk * WORDS].reshape(nblk, WORDS)
    tile_log = None
    if h2r:                                             # persistent workgroups: records [G, 2G) hold the tile log (tile start, epilogue start)
        G = min(512, nblk // 2)                         # 2 workgroups per CU (launch_conv_h2r)
        tile_log = raw[G:2 * G].astype(np.int64)
        raw = raw[:G]
    raw = raw[raw[:, 5] != 0]                           # grid padding: workgroups that returned at once wrote nothing
    nblk = raw.shape[0]
    hdr = raw[:, :8]
    st = raw[:, 8:].reshape(nblk, 4, STEPS, 5).astype(np.int64)
    nks = int(hdr[0, 6])
    out = []
    P = out.append
    P(f"kernel: {'conv_h2r_kernel<3,2> (register-weights quad, two products)' if h2r else 'conv_h2q_kernel<3> (h2 quad patch)' if h2q else 'conv_tap_kernel<2,2,2,3> (v5)' if a.kernel == 'tap' else 'conv_lds_kernel<2,2,2,3,3,1> (v2)'}")
    P(f"conv {a.cin}->{a.cout} 3x3 on 64 x {a.hw}, {nblk} workgroups, {nks} k-steps, instrumented kernel time {ms:.3f} ms")
    life = (hdr[:, 5].astype(np.int64) - hdr[:, 4].astype(np.int64))
    P(f"workgroup lifetime (s_memtime ticks): mean {life.mean():.0f}  p10 {np.percentile(life, 10):.0f}  p90 {np.percentile(life, 90):.0f}"
      f"  -> {life.mean() / nks:.0f} ticks per k-step incl. prologue/epilogue")
    # steps whose 5 stamps are all from the same pass of the ring: nks-64 .. nks-3
    steps = np.arange(max(nks - STEPS + 2, 0), nks - 2)
    sl = steps & (STEPS - 1)
    t = st[:, :, sl, :]                                     # (blk, wave, step, 5)
    nxt = st[:, :, (steps + 1) & (STEPS - 1), 0]
    if h2q:
        seg = {
            "t0->t1 row reads issued + wait own requests (vmcnt 0)": t[..., 1] - t[..., 0],
            "t1->t2 barrier": t[..., 2] - t[..., 1],
            "t2->t3 weight reads + DMA requests + all operands in registers": t[..., 3] - t[..., 2],
            "t3->t4 36 MFMAs issued (576 if alone)": t[..., 4] - t[..., 3],
            "t4->t0' to the next step (chunk flush after tap 8)": nxt - t[..., 4],
            "whole tap step": nxt - t[..., 0],
        } if not h2r else {
            "t0->t1 weight requests of tap + 2 issued + counted wait for this tap's": t[..., 1] - t[..., 0],
            "t1->t2 (tap 8 only: first row's 6 MFMAs + chunk barrier)": t[..., 2] - t[..., 1],
            "t2->t3 first row's operands + its 6 MFMAs issued (96 if alone; tap 8: 0)": t[..., 3] - t[..., 2],
            "t3->t4 the other 18 MFMAs + row prefetch reads (288 if alone)": t[..., 4] - t[..., 3],
            "t4->t0' to the next step (chunk flush after tap 8)": nxt - t[..., 4],
            "whole tap step (384 if alone)": nxt - t[..., 0],
        }
        for wv in range(4):
            v = (nxt - t[..., 0])[:, wv].reshape(-1)
            m_ = (t[..., 4] - t[..., 3])[:, wv].reshape(-1)
            r_ = (t[..., 3] - t[..., 2])[:, wv].reshape(-1)
            P(f"  wave {wv}: whole step mean {v.mean():.0f}  MFMA issue {m_.mean():.0f}  reads+requests {r_.mean():.0f}")
        names = ("wait", "barrier", "reads", "mfma", "tail")
        parts = (t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3], nxt - t[..., 4])
        P("  per wave and tap (mean ticks): " + " | ".join(names) + " | whole")
        for tp in range(9):
            sel = (steps % 9) == tp
            if sel.any():
                for wv in range(4):
                    P(f"    tap {tp} wave {wv}: " + " ".join(f"{pp[:, wv][:, sel].mean():6.0f}" for pp in parts)
                      + f" | {(nxt - t[..., 0])[:, wv][:, sel].mean():6.0f}")
        for tp in range(9):
            sel = (steps % 9) == tp
            if sel.any():
                P(f"  tap {tp}: whole step mean {(nxt - t[..., 0])[:, :, sel].mean():.0f}  wait {(t[..., 1] - t[..., 0])[:, :, sel].mean():.0f}"
                  f"  barrier {(t[..., 2] - t[..., 1])[:, :, sel].mean():.0f}  reads {(t[..., 3] - t[..., 2])[:, :, sel].mean():.0f}  mfma {(t[..., 4] - t[..., 3])[:, :, sel].mean():.0f}")
    elif a.kernel == "tap":
        seg = {
            "t0->t1 wait own requests of this step (vmcnt n)": t[..., 1] - t[..., 0],
            "t1->t2 barrier": t[..., 2] - t[..., 1],
            "t2->t3 issue 2-3 DMA requests + wait fragments (ds_read)": t[..., 3] - t[..., 2],
            "t3->t4 24 MFMAs issued (768 if alone)": t[..., 4] - t[..., 3],
            "t4->t0' to the next step": nxt - t[..., 4],
            "whole k-step": nxt - t[..., 0],
        }
    else:
        seg = {
            "t0->t1 issue prefetch + wait fragments (ds_read)": t[..., 1] - t[..., 0],
            "t1->t2 24 MFMAs issued (768 if alone)": t[..., 2] - t[..., 1],
            "t2->t3 wait prefetch (vmcnt 0)": t[..., 3] - t[..., 2],
            "t3->t4 ds_write + barrier": t[..., 4] - t[..., 3],
            "t4->t0' loop back": nxt - t[..., 4],
            "whole k-step": nxt - t[..., 0],
        }
    P("\nper k-step segment, ticks (all workgroups, waves, steps %d..%d):" % (steps[0], steps[-1]))
    for k, v in seg.items():
        v = v.reshape(-1)
        P(f"  {k:52s} mean {v.mean():7.0f}  p10 {np.percentile(v, 10):6.0f}  p50 {np.percentile(v, 50):6.0f}  p90 {np.percentile(v, 90):6.0f}")
    # prologue / epilogue (ring keeps steps nks-64 ..): epilogue = last MFMA issued -> end stamp
    first = max(nks - STEPS, 0)
    t_first = st[:, 0, first & (STEPS - 1), 0]
    t_lastm = st[:, 0, (nks - 1) & (STEPS - 1), 4 if a.kernel in ("tap", "h2q", "h2r") else 2]
    tb, te = hdr[:, 4].astype(np.int64), hdr[:, 5].astype(np.int64)
    epi = te - t_lastm
    mainloop = (t_lastm - t_first) / (nks - first)
    pro = (t_first - tb) - mainloop * first
    P(f"\nwave 0: epilogue (last MFMA issued -> all stores issued) mean {epi.mean():.0f} p10 {np.percentile(epi, 10):.0f} p90 {np.percentile(epi, 90):.0f} ticks;"
      f" prologue (estimated) mean {pro.mean():.0f}; main loop {mainloop.mean():.0f} ticks/k-step")
    # --- one SIMD's view
    hw = (hdr[:, 0] & 0xFFFFFFFF).astype(np.int64)         # wave 0 of every workgroup
    xcc = (hdr[:, 0] >> 32).astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    P(f"\ndistinct (xcc, se, sh, cu) seen: {uniq.size}; workgroups per CU: min {cnt.min()} max {cnt.max()}")
    pick = uniq[np.argmax(cnt)]
    blks = np.where(key == pick)[0]
    blks = blks[np.argsort(hdr[blks, 4])]
    # take workgroups from the middle of the kernel
    ref = blks[len(blks) // 2]
    t_lo = int(st[ref, 0, steps[5] & (STEPS - 1), 0])
    t_hi = t_lo + 160 * 64
    rec_lo = st[blks, 0, steps[0] & (STEPS - 1), 0]
    rec_hi = st[blks, 0, steps[-1] & (STEPS - 1), 0]
    mid = blks[(rec_lo < t_hi) & (rec_hi > t_lo)]
    P(f"CU key {pick}: {len(blks)} workgroups; timeline of wave 0 of the workgroups alive around tick {t_lo} (their SIMD ids: "
      + ", ".join(str(int((hdr[b, 0] >> 4) & 3)) for b in mid) + ")")
    width, res = 160, 64                                    # 160 columns x 64 ticks
    for wv in range(4):
        P(f"  -- wave {wv} of each workgroup (one SIMD per wave index if the 4 waves spread over the 4 SIMDs):")
        for b in mid:
            simd = int((hdr[b, wv] >> 4) & 3)
            row = [" "] * width
            for s in steps:
                e = st[b, wv, s & (STEPS - 1)]
                n0 = st[b, wv, (s + 1) & (STEPS - 1), 0]
                segs = (((e[0], e[1], "w"), (e[1], e[2], "b"), (e[2], e[3], "l"), (e[3], e[4], "M"), (e[4], n0, ".")) if a.kernel in ("tap", "h2q", "h2r")
                        else ((e[0], e[1], "l"), (e[1], e[2], "M"), (e[2], e[3], "w"), (e[3], e[4], "b"), (e[4], n0, ".")))
                for (lo, hi, ch) in segs:
                    c0, c1 = int((lo - t_lo) // res), int((hi - t_lo) // res)
                    for c in range(max(c0, 0), min(c1 + 1, width)):
                        row[c] = ch
            P(f"     wg {int(hdr[b, 7]):5d} simd {simd}: " + "".join(row))
    P("  legend: " + ("w = counted vmcnt wait, b = barrier, l = DMA requests + ds_read wait, M = MFMA burst being issued"
                     if a.kernel in ("tap", "h2q", "h2r") else
                     "l = prefetch issue + ds_read wait, M = MFMA burst being issued, w = vmcnt wait, b = ds_write + barrier")
      + f"; 1 column = {res} ticks")
    if tile_log is not None:
        # phase relation of the two workgroups of a CU over the launch: tile starts (S) and epilogue starts (E) of both, relative to the first
        P("\nphase of the two workgroups of some CUs (ticks since the launch's first stamp; S tile start, E epilogue start):")
        t00 = tile_log[tile_log > 0].min()
        shown = 0
        for kk in uniq:
            bl = np.where(key == kk)[0]
            if len(bl) != 2 or shown >= 4:
                continue
            shown += 1
            for b in bl:
                bid_ = int(hdr[b, 7])
                row = tile_log[bid_]
                row = row[row > 0] - t00
                P(f"  CU {int(kk):4d} wg {bid_:4d} tg {(int(hdr[b, 0]) >> 16) & 15}: " + " ".join(f"{'S' if i % 2 == 0 else 'E'}{v}" for i, v in enumerate(row[:20])))
        # overlap statistic: fraction of each workgroup's epilogue+prologue span (E_i .. S_{i+1} + first tap) that falls inside the partner's main loops
        tot = ov = 0
        for kk in uniq:
            bl = np.where(key == kk)[0]
            if len(bl) != 2:
                continue
            logs = []
            for b in bl:
                row = tile_log[int(hdr[b, 7])]
                row = row[row > 0]
                logs.append(row)
            for me, other in ((0, 1), (1, 0)):
                a_, o_ = logs[me], logs[other]
                nt_ = len(a_) // 2
                for i in range(nt_ - 1):
                    e0, s1 = a_[2 * i + 1], a_[2 * i + 2]             # my epilogue .. my next tile start
                    tot += s1 - e0
                    for j in range(len(o_) // 2):
                        m0, m1 = o_[2 * j], o_[2 * j + 1]              # partner's tile start .. its epilogue start (~ its main loop)
                        ov += max(0, min(s1, m1) - max(e0, m0))
        if tot:
            P(f"  epilogue spans that ran beside the partner's main loop: {ov / tot:.2f} of their time (0 = in phase, 1 = anti-phase)")
    # compact copy of 16 CUs for offline analysis
    keep = np.where(np.isin(key, uniq[:16]))[0]
    np.savez_compressed(str(Path(a.out).with_suffix(".npz")), hdr=hdr[keep], st=(st[keep] - tb[keep, None, None, None]).astype(np.int32),
                        key=key[keep], tb=tb[keep], te=te[keep])
    txt = "\n".join(out)
    print(txt)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(txt + "\n")


if __name__ == "__main__":
    main()
