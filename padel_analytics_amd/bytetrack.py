"""ByteTrack on the host — stand-in for ``sv.ByteTrack`` as used at ``players_tracker.py:311,367-369``
(SURVEY.md §8 a5 / (f)#1).  Stateful and sequential in global frame order, so it stays on the CPU (rank 0)
after the per-GPU detections are gathered.

Restated from the published ByteTrack algorithm in the form the ``supervision`` 0.21-0.23 line ships it
(``main.py:118`` passes ``frame_resolution_wh=`` to ``PolygonZone``, which that line still accepts):
8-state constant-velocity Kalman filter on (cx, cy, aspect, h); first association of confirmed + lost
tracks with high-score detections on fused (IoU x score) cost, threshold ``minimum_matching_threshold``;
second association of the remaining tracked tracks with low-score detections (IoU, 0.5); unconfirmed
tracks against what is left (0.7); new tracks from detections above ``track_activation_threshold + 0.1``;
lost tracks expire after ``frame_rate / 30 * lost_track_buffer`` frames.  ``update_with_detections``
returns only detections matched (IoU) to an active track, carrying the detector's own boxes and the track
id — unmatched / unconfirmed detections are dropped, like upstream.

Id semantics (chosen, documented): tracks carry an *internal* id from birth and receive their public
(*external*) id only when they are first confirmed — on frame 1 at activation, otherwise at the first
matched update — so spurious one-frame detections do not consume public ids (supervision >= 0.21).

The Kalman predict / update of all tracks of a frame run as stacked (n, 8) / (n, 8, 8) numpy operations: the
runner hands this class ~10^2 detections per frame on synthetic weights, and per-track Python would cost more
than the GPU forward.  ``tests/golden/make_bytetrack_golden.py`` is an independent scalar implementation of the
same published algorithm; ``tests/test_bytetrack_golden.py`` pins this file to its outputs.

*Parity unpinned* against supervision itself: not installable offline and un-pinned in the reference
(``requirements.txt:8``); no reference test pins ids.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

from .detections import Detections

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3

_WP, _WV = 1.0 / 20, 1.0 / 160
_F = np.eye(8)
for _i in range(4):
    _F[_i, 4 + _i] = 1.0
_H = np.eye(4, 8)


def kf_initiate(m: np.ndarray):
    mean = np.r_[m, np.zeros(4)]
    h = m[3]
    std = np.array([2 * _WP * h, 2 * _WP * h, 1e-2, 2 * _WP * h, 10 * _WV * h, 10 * _WV * h, 1e-5, 10 * _WV * h])
    return mean, np.diag(np.square(std))


def kf_multi_predict(means: np.ndarray, covs: np.ndarray):
    """means (n, 8), covs (n, 8, 8) -> predicted (one constant-velocity step)."""
    h = means[:, 3]
    std = np.stack([_WP * h, _WP * h, np.full_like(h, 1e-2), _WP * h, _WV * h, _WV * h, np.full_like(h, 1e-5), _WV * h], 1)
    q = np.zeros_like(covs)
    idx = np.arange(8)
    q[:, idx, idx] = np.square(std)
    means = means @ _F.T
    covs = _F @ covs @ _F.T + q
    return means, covs


def kf_multi_update(means: np.ndarray, covs: np.ndarray, meas: np.ndarray):
    """Stacked Kalman correction with measurements (n, 4) = (cx, cy, aspect, h)."""
    h = means[:, 3]
    std = np.stack([_WP * h, _WP * h, np.full_like(h, 1e-1), _WP * h], 1)
    r = np.zeros((len(means), 4, 4))
    idx = np.arange(4)
    r[:, idx, idx] = np.square(std)
    pm = means @ _H.T                                  # (n, 4)
    pc = _H @ covs @ _H.T + r                          # (n, 4, 4)
    b = covs @ _H.T                                    # (n, 8, 4)
    k = np.linalg.solve(pc, b.transpose(0, 2, 1)).transpose(0, 2, 1)     # (n, 8, 4) (pc symmetric)
    innov = meas - pm
    new_means = means + np.einsum("nij,nj->ni", k, innov)
    new_covs = covs - k @ pc @ k.transpose(0, 2, 1)
    return new_means, new_covs


class STrack:
    __slots__ = ("_tlwh", "score", "mean", "cov", "is_activated", "state", "internal_id", "track_id", "frame_id",
                 "start_frame", "tracklet_len")

    def __init__(self, tlwh, score):
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.score = float(score)
        self.mean = self.cov = None
        self.is_activated = False
        self.state = NEW
        self.internal_id = 0
        self.track_id = -1          # public id, assigned at confirmation
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


def _tlbr_of(tracks) -> np.ndarray:
    """(n, 4) x1,y1,x2,y2 of a list of STracks (stacked: state mean if activated, else the detection box)."""
    if not tracks:
        return np.zeros((0, 4))
    out = np.empty((len(tracks), 4))
    for i, t in enumerate(tracks):
        out[i] = t.tlbr
    return out


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
    return 1 - box_iou_batch(_tlbr_of(ta), _tlbr_of(tb))


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
    ok = c[rows, cols] <= thresh
    matches = [(int(r), int(k)) for r, k in zip(rows[ok], cols[ok])]
    mr = np.zeros(cost.shape[0], bool)
    mc = np.zeros(cost.shape[1], bool)
    mr[rows[ok]] = True
    mc[cols[ok]] = True
    return matches, np.nonzero(~mr)[0].tolist(), np.nonzero(~mc)[0].tolist()


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
        self.reset()

    def reset(self) -> None:
        self.frame_id = 0
        self.tracked, self.lost, self.removed = [], [], []
        self._next_internal = 0
        self._next_id = 0

    def _new_id(self) -> int:
        self._next_id += 1
        return self._next_id

    def _confirm(self, t: STrack) -> None:
        t.is_activated = True
        if t.track_id == -1:
            t.track_id = self._new_id()

    # ---- stacked Kalman steps over lists of tracks
    def _predict(self, pool) -> None:
        if not pool:
            return
        means = np.stack([t.mean for t in pool])
        covs = np.stack([t.cov for t in pool])
        for i, t in enumerate(pool):
            if t.state != TRACKED:
                means[i, 7] = 0
        means, covs = kf_multi_predict(means, covs)
        for i, t in enumerate(pool):
            t.mean, t.cov = means[i], covs[i]

    def _correct(self, pairs) -> None:
        """pairs: [(track, detection STrack)] -> Kalman update of each track with its detection."""
        if not pairs:
            return
        means = np.stack([t.mean for t, _ in pairs])
        covs = np.stack([t.cov for t, _ in pairs])
        meas = np.stack([STrack.to_xyah(d._tlwh) for _, d in pairs])
        means, covs = kf_multi_update(means, covs, meas)
        for i, (t, d) in enumerate(pairs):
            t.mean, t.cov = means[i], covs[i]
            t.score = d.score

    def update_with_detections(self, detections: Detections) -> Detections:
        conf = detections.confidence if detections.confidence is not None else np.ones(len(detections))
        tensors = np.hstack((detections.xyxy.astype(np.float64), np.asarray(conf, np.float64)[:, None])) \
            if len(detections) else np.zeros((0, 5))
        tracks = self.update_with_tensors(tensors)
        if len(tracks) and len(detections):
            cost = 1 - box_iou_batch(tensors[:, :4], _tlbr_of(tracks))
            matches, _, _ = linear_assignment(cost, 0.5)
            tid = np.full(len(detections), -1, dtype=int)
            for i_det, i_trk in matches:
                tid[i_det] = tracks[i_trk].track_id
            detections.tracker_id = tid
            return detections[tid != -1]
        out = Detections.empty()
        out.tracker_id = np.array([], dtype=int)
        return out

    def update_with_tensors(self, tensors: np.ndarray) -> list:
        self.frame_id += 1
        fid = self.frame_id
        activated, refind, lost, removed = [], [], [], []
        scores, boxes = tensors[:, 4], tensors[:, :4]
        keep = scores > self.track_thresh
        second = (scores > 0.1) & (scores < self.track_thresh)

        def mk(bb, ss):
            tlwh = np.concatenate([bb[:, :2], bb[:, 2:] - bb[:, :2]], 1) if len(bb) else np.zeros((0, 4))
            return [STrack(tlwh[i], ss[i]) for i in range(len(bb))]

        dets, dets2 = mk(boxes[keep], scores[keep]), mk(boxes[second], scores[second])
        unconfirmed = [t for t in self.tracked if not t.is_activated]
        tracked = [t for t in self.tracked if t.is_activated]
        pool = _joint(tracked, self.lost)
        self._predict(pool)

        def apply(matches, trk, det):
            """Kalman-correct the matched tracks, then the per-track bookkeeping of update / re_activate."""
            self._correct([(trk[it], det[idet]) for it, idet in matches])
            for it, _ in matches:
                t = trk[it]
                if t.state == TRACKED:
                    t.tracklet_len += 1
                    activated.append(t)
                else:
                    t.tracklet_len = 0
                    refind.append(t)
                t.state = TRACKED
                t.frame_id = fid
                self._confirm(t)

        d = fuse_score(iou_distance(pool, dets), dets)
        matches, u_trk, u_det = linear_assignment(d, self.match_thresh)
        apply(matches, pool, dets)
        r_tracked = [pool[i] for i in u_trk if pool[i].state == TRACKED]
        matches, u_trk2, _ = linear_assignment(iou_distance(r_tracked, dets2), 0.5)
        apply(matches, r_tracked, dets2)
        for it in u_trk2:
            t = r_tracked[it]
            if t.state != LOST:
                t.state = LOST
                lost.append(t)
        rest = [dets[i] for i in u_det]
        d = fuse_score(iou_distance(unconfirmed, rest), rest)
        matches, u_unc, u_det = linear_assignment(d, 0.7)
        apply(matches, unconfirmed, rest)
        for it in u_unc:
            unconfirmed[it].state = REMOVED
            removed.append(unconfirmed[it])
        for i in u_det:
            t = rest[i]
            if t.score < self.det_thresh:
                continue
            self._next_internal += 1
            t.internal_id = self._next_internal
            t.mean, t.cov = kf_initiate(STrack.to_xyah(t._tlwh))
            t.tracklet_len = 0
            t.state = TRACKED
            t.frame_id = t.start_frame = fid
            if fid == 1:
                self._confirm(t)
            activated.append(t)
        for t in self.lost:
            if fid - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        self.tracked = [t for t in self.tracked if t.state == TRACKED]
        self.tracked = _joint(_joint(self.tracked, activated), refind)
        self.lost = _sub(_sub(self.lost, self.tracked) + lost, removed)
        self.lost = [t for t in self.lost if t.state == LOST]
        self.removed = removed                    # only this frame's (supervision keeps no history either)
        self.tracked, self.lost = _remove_duplicates(self.tracked, self.lost)
        return [t for t in self.tracked if t.is_activated]
