"""Fuzz two builds of the native ByteTrack against each other (ids must be identical): random streams with ties — boxes on a
grid, quarter-rounded scores, exact duplicates, crowds, empty frames, zone masks, batches split at random places.

    g++ -O3 -ffp-contract=off -std=c++17 -fPIC -shared -o /tmp/bt/libnew.so padel_analytics_amd/csrc/bytetrack.cpp
    git show <rev>:padel_analytics_amd/csrc/bytetrack.cpp > /tmp/bt/old.cpp   # (fix its #include path)
    g++ -O2 -std=c++17 -fPIC -shared -o /tmp/bt/libold.so /tmp/bt/old.cpp
    python tools/bytetrack_fuzz.py 600
"""
import ctypes as C, numpy as np, sys
def load(p):
    l = C.CDLL(p)
    l.pa_bytetrack_create.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_void_p)]
    l.pa_bytetrack_update_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    l.pa_bytetrack_destroy.argtypes = [C.c_void_p]
    l.pa_bytetrack_reset.argtypes = [C.c_void_p]
    return l
old, new = load('/tmp/bt/libold.so'), load('/tmp/bt/libnew.so')
def run(l, boxes, counts, keep, params, split):
    h = C.c_void_p()
    l.pa_bytetrack_create(*params, C.byref(h))
    nf, stride = counts.shape[0], boxes.shape[1]
    ids = np.zeros((nf, stride), np.int32)
    lo = 0
    for hi in split + [nf]:
        if hi > lo:
            k = None if keep is None else np.ascontiguousarray(keep[lo:hi])
            b = np.ascontiguousarray(boxes[lo:hi]); c = np.ascontiguousarray(counts[lo:hi]); o = np.zeros((hi - lo, stride), np.int32)
            assert l.pa_bytetrack_update_batch(h, b.ctypes.data, c.ctypes.data, None if k is None else k.ctypes.data, hi - lo, stride, o.ctypes.data) == 0
            ids[lo:hi] = o
        lo = hi
    l.pa_bytetrack_destroy(h)
    return ids
nseeds = int(sys.argv[1])
bad = 0
for seed in range(nseeds):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 140)); nf = int(rng.integers(5, 90)); stride = 160
    mode = seed % 6
    pos = rng.uniform([50, 50], [1230, 670], (n, 2))
    if mode == 1: pos = np.round(pos / 40) * 40                 # grid: many identical / touching boxes
    if mode == 2: pos = rng.uniform([500, 300], [700, 420], (n, 2))     # crowded: heavy overlaps
    vel = rng.uniform(-4, 4, (n, 2)) * (0 if mode == 1 else 1)
    boxes = np.zeros((nf, stride, 6), np.float32); counts = np.zeros(nf, np.int32)
    keep = np.zeros((nf, stride), np.uint8) if seed % 2 else None
    pa = float(rng.choice([0.05, 0.1, 0.3]))
    for f in range(nf):
        pos = pos + vel + (0 if mode == 1 else rng.normal(0, 0.7, pos.shape))
        alive = rng.random(n) > pa
        if mode == 3 and f % 7 == 3: alive[:] = False                # empty frames
        k = int(alive.sum())
        wh = np.stack([30 + np.arange(n) % 40, 80 + np.arange(n) % 60], 1)[alive]
        if mode == 4: wh = wh * 0 + [40, 90]
        boxes[f, :k, :2] = pos[alive] - wh / 2; boxes[f, :k, 2:4] = pos[alive] + wh / 2
        sc = rng.uniform(0.05, 0.99, k)
        if mode in (1, 5): sc = np.round(sc * 4) / 4                 # tied scores
        boxes[f, :k, 4] = sc
        if mode == 5 and k > 3:                                      # exact duplicates of boxes
            boxes[f, 1, :4] = boxes[f, 0, :4]; boxes[f, 3, :5] = boxes[f, 2, :5]
        counts[f] = k
        if keep is not None: keep[f, :k] = rng.random(k) > 0.3
    params = (float(rng.choice([0.25, 0.5])), int(rng.choice([2, 5, 30])), float(rng.choice([0.8, 0.6, 0.95])), int(rng.choice([30, 60])))
    split = sorted(set(int(x) for x in rng.integers(0, nf, 3)))
    a = run(old, boxes, counts, keep, params, split); b = run(new, boxes, counts, keep, params, split)
    if not (a == b).all():
        bad += 1
        f = int(np.argmax((a != b).any(1)))
        print("MISMATCH seed", seed, "mode", mode, "frame", f, "n", n)
print("seeds", nseeds, "mismatches", bad)
