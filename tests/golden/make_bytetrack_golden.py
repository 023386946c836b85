"""Independent scalar ByteTrack (per-track Kalman filter objects, no stacked numpy) used ONLY to generate
``tests/golden/bytetrack_golden.json``: the known answers ``tests/test_bytetrack_golden.py`` pins the product's
vectorised ``padel_analytics_amd/bytetrack.py`` to.  Same published algorithm (ByteTrack, in the form the
supervision 0.21-0.23 line ships it: reference call sites ``players_tracker.py:311,367-369``), same chosen id
semantics (public ids assigned at confirmation).

Scenarios are scripted so the hand-checkable events of VERDICT item 8 occur at known frames:
  * "two_players": two boxes moving at constant velocity; the track born on frame 1 is confirmed at once
    (ids 1, 2), a third box appearing on frame 3 is reported only from frame 4 (confirmed at its first matched
    update) with id 3;
  * "occlusion": a track disappears for 5 frames and comes back inside ``lost_track_buffer`` -> same id;
  * "expiry": with ``frame_rate=30, lost_track_buffer=3`` a track missing for 4 frames comes back with a NEW id;
  * "low_score": a detection whose score drops to 0.2 (< track_activation_threshold .25, > .1) keeps its track
    through the second association;
  * "spurious": one-frame detections never get a public id, so later ids stay dense;
  * "random": 40 frames of 12 jittering boxes with random drop-outs and false positives (seeded).

    python tests/golden/make_bytetrack_golden.py      # rewrites tests/golden/bytetrack_golden.json
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment


NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3


class KalmanFilter:
    def __init__(self):
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)
        self.wp, self.wv = 1.0 / 20, 1.0 / 160

    def initiate(self, m):
        mean = np.r_[m, np.zeros(4)]
        h = m[3]
        std = [2 * self.wp * h, 2 * self.wp * h, 1e-2, 2 * self.wp * h,
               10 * self.wv * h, 10 * self.wv * h, 1e-5, 10 * self.wv * h]
        return mean, np.diag(np.square(std))

    def predict(self, mean, cov):
        h = mean[3]
        std = [self.wp * h, self.wp * h, 1e-2, self.wp * h, self.wv * h, self.wv * h, 1e-5, self.wv * h]
        mean = self.F @ mean
        cov = self.F @ cov @ self.F.T + np.diag(np.square(std))
        return mean, cov

    def project(self, mean, cov):
        h = mean[3]
        std = [self.wp * h, self.wp * h, 1e-1, self.wp * h]
        return self.H @ mean, self.H @ cov @ self.H.T + np.diag(np.square(std))

    def update(self, mean, cov, m):
        pm, pc = self.project(mean, cov)
        k = np.linalg.solve(pc, (cov @ self.H.T).T).T
        return mean + (m - pm) @ k.T, cov - k @ pc @ k.T


class STrack:
    def __init__(self, tlwh, score):
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.score = float(score)
        self.mean = self.cov = None
        self.is_activated = False
        self.state = NEW
        self.track_id = -1           # public id (assigned at confirmation)
        self.internal_id = 0
        self.frame_id = self.start_frame = 0
        self.tracklet_len = 0

    @property
    def tlwh(self):
        if self.mean is None:
            return self._tlwh.copy()
        r = self.mean[:4].copy()
        r[2] *= r[3]
        r[:2] -= r[2:] / 2
        return r

    @property
    def tlbr(self):
        r = self.tlwh
        r[2:] += r[:2]
        return r

    @staticmethod
    def to_xyah(tlwh):
        r = np.asarray(tlwh, dtype=np.float64).copy()
        r[:2] += r[2:] / 2
        r[2] /= r[3]
        return r

    def predict(self, kf):
        mean = self.mean.copy()
        if self.state != TRACKED:
            mean[7] = 0
        self.mean, self.cov = kf.predict(mean, self.cov)

    def activate(self, kf, frame_id, internal_id, ids):
        self.internal_id = internal_id
        self.mean, self.cov = kf.initiate(self.to_xyah(self._tlwh))
        self.tracklet_len = 0
        self.state = TRACKED
        self.frame_id = self.start_frame = frame_id
        if frame_id == 1:
            self.confirm(ids)

    def confirm(self, ids):
        self.is_activated = True
        if self.track_id == -1:
            self.track_id = ids()

    def re_activate(self, kf, new, frame_id, ids):
        self.mean, self.cov = kf.update(self.mean, self.cov, self.to_xyah(new.tlwh))
        self.tracklet_len = 0
        self.state = TRACKED
        self.confirm(ids)
        self.frame_id = frame_id
        self.score = new.score

    def update(self, kf, new, frame_id, ids):
        self.frame_id = frame_id
        self.tracklet_len += 1
        self.mean, self.cov = kf.update(self.mean, self.cov, self.to_xyah(new.tlwh))
        self.state = TRACKED
        self.confirm(ids)
        self.score = new.score


def box_iou_batch(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    tl = np.maximum(a[:, None, :2], b[None, :, :2])
    br = np.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = np.prod(np.clip(br - tl, 0, None), 2)
    return inter / (area_a[:, None] + area_b[None] - inter)


def iou_distance(ta, tb):
    return 1 - box_iou_batch(np.array([t.tlbr for t in ta]).reshape(-1, 4), np.array([t.tlbr for t in tb]).reshape(-1, 4))


def fuse_score(cost, dets):
    if cost.size == 0:
        return cost
    sim = (1 - cost) * np.array([d.score for d in dets])[None]
    return 1 - sim


def linear_assignment(cost, thresh):
    if cost.size == 0:
        return [], list(range(cost.shape[0])), list(range(cost.shape[1]))
    c = cost.copy()
    c[c > thresh] = thresh + 1e-4
    rows, cols = linear_sum_assignment(c)
    matches = [(int(r), int(k)) for r, k in zip(rows, cols) if c[r, k] <= thresh]
    mr, mc = {m[0] for m in matches}, {m[1] for m in matches}
    return matches, [i for i in range(cost.shape[0]) if i not in mr], [j for j in range(cost.shape[1]) if j not in mc]


def _joint(a, b):
    seen = {t.internal_id for t in a}
    return a + [t for t in b if t.internal_id not in seen]


def _sub(a, b):
    ids = {t.internal_id for t in b}
    return [t for t in a if t.internal_id not in ids]


def _remove_duplicates(a, b):
    d = iou_distance(a, b)
    pa, pb = np.where(d < 0.15) if d.size else ([], [])
    da, db = set(), set()
    for p, q in zip(pa, pb):
        if a[p].frame_id - a[p].start_frame > b[q].frame_id - b[q].start_frame:
            db.add(q)
        else:
            da.add(p)
    return [t for i, t in enumerate(a) if i not in da], [t for i, t in enumerate(b) if i not in db]


class ByteTrack:
    def __init__(self, track_activation_threshold: float = 0.25, lost_track_buffer: int = 30,
                 minimum_matching_threshold: float = 0.8, frame_rate: int = 30):
        self.track_thresh = track_activation_threshold
        self.match_thresh = minimum_matching_threshold
        self.det_thresh = track_activation_threshold + 0.1
        self.max_time_lost = int(frame_rate / 30.0 * lost_track_buffer)
        self.kf = KalmanFilter()
        self.reset()

    def reset(self) -> None:
        self.frame_id = 0
        self.tracked, self.lost, self.removed = [], [], []
        self._next_id = 0
        self._next_internal = 0

    def _new_id(self) -> int:
        self._next_id += 1
        return self._next_id

    def update_with_detections(self, xyxy, conf):
        """-> (kept detection indices, their public track ids)."""
        tensors = np.hstack((np.asarray(xyxy, np.float64).reshape(-1, 4), np.asarray(conf, np.float64).reshape(-1, 1)))
        tracks = self.update_with_tensors(tensors)
        if len(tracks) and len(tensors):
            cost = 1 - box_iou_batch(tensors[:, :4], np.array([t.tlbr for t in tracks]))
            matches, _, _ = linear_assignment(cost, 0.5)
            tid = np.full(len(tensors), -1, dtype=int)
            for i_det, i_trk in matches:
                tid[i_det] = tracks[i_trk].track_id
            keep = np.nonzero(tid != -1)[0]
            return keep.tolist(), tid[keep].tolist()
        return [], []

    def update_with_tensors(self, tensors: np.ndarray) -> list:
        self.frame_id += 1
        activated, refind, lost, removed = [], [], [], []
        scores, boxes = tensors[:, 4], tensors[:, :4]
        keep = scores > self.track_thresh
        second = (scores > 0.1) & (scores < self.track_thresh)
        mk = lambda bb, ss: [STrack(np.r_[b[:2], b[2:] - b[:2]], s) for b, s in zip(bb, ss)]
        dets, dets2 = mk(boxes[keep], scores[keep]), mk(boxes[second], scores[second])
        unconfirmed = [t for t in self.tracked if not t.is_activated]
        tracked = [t for t in self.tracked if t.is_activated]
        pool = _joint(tracked, self.lost)
        for t in pool:
            t.predict(self.kf)
        d = fuse_score(iou_distance(pool, dets), dets)
        matches, u_trk, u_det = linear_assignment(d, self.match_thresh)
        for it, idet in matches:
            t = pool[it]
            if t.state == TRACKED:
                t.update(self.kf, dets[idet], self.frame_id, self._new_id); activated.append(t)
            else:
                t.re_activate(self.kf, dets[idet], self.frame_id, self._new_id); refind.append(t)
        r_tracked = [pool[i] for i in u_trk if pool[i].state == TRACKED]
        matches, u_trk2, _ = linear_assignment(iou_distance(r_tracked, dets2), 0.5)
        for it, idet in matches:
            t = r_tracked[it]
            if t.state == TRACKED:
                t.update(self.kf, dets2[idet], self.frame_id, self._new_id); activated.append(t)
            else:
                t.re_activate(self.kf, dets2[idet], self.frame_id, self._new_id); refind.append(t)
        for it in u_trk2:
            t = r_tracked[it]
            if t.state != LOST:
                t.state = LOST; lost.append(t)
        rest = [dets[i] for i in u_det]
        d = fuse_score(iou_distance(unconfirmed, rest), rest)
        matches, u_unc, u_det = linear_assignment(d, 0.7)
        for it, idet in matches:
            unconfirmed[it].update(self.kf, rest[idet], self.frame_id, self._new_id); activated.append(unconfirmed[it])
        for it in u_unc:
            unconfirmed[it].state = REMOVED; removed.append(unconfirmed[it])
        for i in u_det:
            t = rest[i]
            if t.score < self.det_thresh:
                continue
            self._next_internal += 1
            t.activate(self.kf, self.frame_id, self._next_internal, self._new_id); activated.append(t)
        for t in self.lost:
            if self.frame_id - t.frame_id > self.max_time_lost:
                t.state = REMOVED; removed.append(t)
        self.tracked = [t for t in self.tracked if t.state == TRACKED]
        self.tracked = _joint(_joint(self.tracked, activated), refind)
        self.lost = _sub(_sub(self.lost, self.tracked) + lost, self.removed + removed)
        self.removed += removed
        self.tracked, self.lost = _remove_duplicates(self.tracked, self.lost)
        return [t for t in self.tracked if t.is_activated]


def scenarios():
    out = {}

    def box(cx, cy, w=40.0, h=100.0):
        return [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]

    fr = []
    for f in range(10):
        b = [box(100 + 5 * f, 200), box(400 - 4 * f, 220 + 2 * f)]
        s = [0.9, 0.8]
        if f >= 2:
            b.append(box(250, 300 + 3 * f)); s.append(0.7)
        fr.append((b, s))
    out["two_players"] = dict(params=dict(), frames=fr)

    fr = []
    for f in range(16):
        b, s = [box(100 + 3 * f, 200)], [0.9]
        if not (5 <= f < 10):
            b.append(box(300 + 2 * f, 250)); s.append(0.85)
        fr.append((b, s))
    out["occlusion"] = dict(params=dict(), frames=fr)

    fr = []
    for f in range(14):
        b, s = [box(100 + 3 * f, 200)], [0.9]
        if not (4 <= f < 8):
            b.append(box(300 + 2 * f, 250)); s.append(0.85)
        fr.append((b, s))
    out["expiry"] = dict(params=dict(lost_track_buffer=3), frames=fr)

    fr = []
    for f in range(10):
        fr.append(([box(100 + 3 * f, 200), box(300, 250 + f)], [0.9, 0.2 if 4 <= f < 7 else 0.8]))
    out["low_score"] = dict(params=dict(), frames=fr)

    fr = []
    for f in range(12):
        b, s = [box(100 + 3 * f, 200)], [0.9]
        if f in (2, 5, 8):
            b.append(box(500 + 30 * f, 400)); s.append(0.6)      # one-frame false positives
        if f >= 6:
            b.append(box(300, 100 + 4 * f)); s.append(0.75)
        fr.append((b, s))
    out["spurious"] = dict(params=dict(), frames=fr)

    rng = np.random.default_rng(7)
    pos = rng.uniform([100, 100], [1100, 600], (12, 2))
    vel = rng.uniform(-6, 6, (12, 2))
    fr = []
    for f in range(40):
        pos = pos + vel + rng.normal(0, 1.0, pos.shape)
        b, s = [], []
        for i in range(12):
            if rng.random() < 0.12:
                continue
            b.append(box(pos[i, 0], pos[i, 1], 40 + i, 90 + 2 * i)); s.append(float(rng.uniform(0.15, 0.95)))
        for _ in range(int(rng.integers(0, 3))):
            b.append(box(*rng.uniform([50, 50], [1200, 650]))); s.append(float(rng.uniform(0.3, 0.7)))
        fr.append((b, s))
    out["random"] = dict(params=dict(), frames=fr)
    return out


def main():
    import json
    from pathlib import Path
    res = {}
    for name, sc in scenarios().items():
        bt = ByteTrack(frame_rate=30, **sc["params"])
        frames = []
        for b, s in sc["frames"]:
            keep, ids = bt.update_with_detections(np.array(b, np.float32).reshape(-1, 4), np.array(s, np.float32))
            frames.append(dict(xyxy=[[float(np.float32(v)) for v in bb] for bb in b], conf=[float(np.float32(v)) for v in s],
                               keep=keep, ids=ids))
        res[name] = dict(params=sc["params"], frames=frames)
    p = Path(__file__).with_name("bytetrack_golden.json")
    p.write_text(json.dumps(res))
    print("wrote", p, {k: sum(len(f["keep"]) for f in v["frames"]) for k, v in res.items()})


if __name__ == "__main__":
    main()
