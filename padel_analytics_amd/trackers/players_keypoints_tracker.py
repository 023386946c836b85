"""Players keypoints tracker — drop-in for the reference's
``trackers/players_keypoints_tracker/players_keypoints_tracker.py`` (``PlayerKeypoint`` :14-42,
``PlayerKeypoints`` :59-135, ``PlayersKeypoints`` :165-197, ``PlayerKeypointsTracker`` :207-325).

Hot path (``predict_sample`` :271-322): BGR frame -> RGB -> Pillow bicubic stretch to SxS (S in {640,1280})
-> YOLOv8-pose (conf .25, iou .7, ``classes=[0]``) -> ``keypoints.xy`` scaled back by (w/S, h/S).  The
resize, the network, NMS and the keypoint decode all run on the GPU (``pil_stretch=True``).

Divergence from the reference, on purpose: the reference's
``xy.squeeze(0)`` logic (:299-301) raises for exactly 1 or 2 detections; here every n yields n x 13 keypoints.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Iterable, Optional, Type

import numpy as np

from ..yolo import YOLO
from .tracker import NoPredictFrames, Object, Tracker


@dataclass
class PlayerKeypoint:
    id: int
    name: str
    xy: tuple

    def asint(self) -> tuple: return tuple(int(v) for v in self.xy)

    @classmethod
    def from_json(cls, x: dict): return cls(**x)

    def serialize(self) -> dict: return {"id": self.id, "name": self.name, "xy": self.xy}


class PlayerKeypoints:
    """The 13 keypoints of one person (reference :59-162): ``player_keypoints`` (list of ``PlayerKeypoint``),
    ``keypoints_by_name``, ``[name]``.  Built from that list (reference signature) or — ``from_xy``, what the tracker
    uses — from the person's (K, 2) row of the detector's keypoint array plus the frame's ratio: the object then costs
    ~0.5 us, and its 13 ``PlayerKeypoint`` records (same values: ``float32 -> float`` times the ratio, :303-316) come into
    being when ``player_keypoints`` / ``keypoints_by_name`` / ``[name]`` / iteration first asks for them."""
    __slots__ = ("_kps", "_by_name", "_xy", "_ratio")

    KEYPOINTS_NAMES = ["left_foot", "right_foot", "torso", "right_shoulder", "left_shoulder", "head", "neck",
                       "left_hand", "right_hand", "right_knee", "left_knee", "right_elbow", "left_elbow"]
    CONNECTIONS = [("left_foot", "left_knee"), ("left_knee", "torso"), ("right_foot", "right_knee"),
                   ("right_knee", "torso"), ("torso", "left_shoulder"), ("torso", "right_shoulder"),
                   ("left_hand", "left_elbow"), ("left_elbow", "left_shoulder"), ("left_shoulder", "neck"),
                   ("neck", "head"), ("right_hand", "right_elbow"), ("right_elbow", "right_shoulder"),
                   ("right_shoulder", "neck")]

    def __init__(self, player_keypoints: list):
        self._kps = player_keypoints
        self._by_name = {k.name: k for k in player_keypoints}
        self._xy, self._ratio = None, (1.0, 1.0)

    @classmethod
    def from_xy(cls, xy: np.ndarray, ratio: tuple = (1.0, 1.0)) -> "PlayerKeypoints":
        """``xy``: (K, 2) float32 network-input coordinates (a view of the tracker's array), ``ratio`` = (w / S, h / S)."""
        p = object.__new__(cls)
        p._xy, p._ratio = xy, ratio
        p._kps = p._by_name = None
        return p

    @property
    def player_keypoints(self) -> list:
        if self._kps is None:
            names = self.KEYPOINTS_NAMES
            rx, ry = self._ratio
            # reference :303-316: keypoint[0].item() * ratio_x — the float32 value widened to a Python float first
            self._kps = [PlayerKeypoint(id=i, name=names[i] if i < len(names) else str(i), xy=(x * rx, y * ry))
                         for i, (x, y) in enumerate(self._xy.tolist())]
        return self._kps

    @property
    def keypoints_by_name(self) -> dict:
        if self._by_name is None:
            self._by_name = {k.name: k for k in self.player_keypoints}
        return self._by_name

    @classmethod
    def from_json(cls, x: dict):
        return cls([PlayerKeypoint.from_json(k) for k in x["player_keypoints"]])

    def serialize(self) -> dict:
        return {"player_keypoints": [k.serialize() for k in self.player_keypoints]}

    def __len__(self) -> int: return len(self._xy) if self._kps is None else len(self._kps)

    def __iter__(self): return iter(self.player_keypoints)

    def __getitem__(self, name: str) -> PlayerKeypoint:
        assert name in self.KEYPOINTS_NAMES
        return self.keypoints_by_name[name]

    def draw(self, frame: np.ndarray) -> np.ndarray: return frame


class PlayersKeypoints(Object):
    """All players' keypoints of one frame (reference :165-197).  Built from ``PlayerKeypoints`` objects (reference
    signature) or from the tracker's (n, K, 2) array of network-input coordinates and the frame's ratio; the per-person
    ``PlayerKeypoints`` objects are then created in the constructor (``EAGER``: where the reference creates them, :303-320)
    or on first access."""

    #: 0 / False (default): the ``PlayerKeypoints`` objects on first access; 1 / True: in the constructor, inside
    #: ``predict_sample`` like the reference (:303-320) — their 13 ``PlayerKeypoint`` records each on first access;
    #: 2: those too (everything the reference allocates, allocated where it allocates it)
    EAGER = False

    def __init__(self, players_keypoints: Optional[list] = None, *, xy: Optional[np.ndarray] = None,
                 ratio: tuple = (1.0, 1.0)) -> None:
        super().__init__()
        self._items = players_keypoints
        self._xy, self._ratio = xy, ratio
        if players_keypoints is None and xy is None:
            self._items = []
        if self.EAGER:
            items = self.players_keypoints
            if self.EAGER == 2:
                for p in items:
                    p.player_keypoints, p.keypoints_by_name

    @property
    def players_keypoints(self) -> list:
        if self._items is None:
            new, ratio = PlayerKeypoints.from_xy, (self._ratio[0], self._ratio[1])
            self._items = [new(person, ratio) for person in self._xy]
        return self._items

    @classmethod
    def from_json(cls, x) -> "PlayersKeypoints":
        return cls([PlayerKeypoints.from_json(p) for p in x])

    def serialize(self) -> list:
        return [p.serialize() for p in self.players_keypoints]

    def __len__(self) -> int: return len(self._xy) if self._items is None else len(self._items)

    def __iter__(self): return iter(self.players_keypoints)

    def __getitem__(self, i: int) -> PlayerKeypoints: return self.players_keypoints[i]


class PlayerKeypointsTracker(Tracker):
    CONF = 0.25
    IOU = 0.7
    streams = False

    def __init__(self, model_path: str, train_image_size: int, batch_size: int,
                 load_path: Optional[str | Path] = None, save_path: Optional[str | Path] = None, half: bool = False):
        super().__init__(load_path=load_path, save_path=save_path)
        self.model = YOLO(model_path, half=half)
        assert train_image_size in (640, 1280)
        self.train_image_size = train_image_size
        self.batch_size = batch_size

    def video_info_post_init(self, video_info) -> "PlayerKeypointsTracker": return self

    def object(self) -> Type[Object]: return PlayersKeypoints

    def draw_kwargs(self) -> dict: return {}

    def __str__(self) -> str: return "players_keypoints_tracker"

    def restart(self) -> None: self.results.restart()

    def processor(self, frame: np.ndarray):
        """BGR2RGB + ``Image.resize((S, S))`` (Pillow default bicubic), reference :260-266; API parity only —
        the hot path does this on the device."""
        from PIL import Image
        return Image.fromarray(np.ascontiguousarray(frame[..., ::-1])).resize((self.train_image_size,) * 2)

    def to(self, device: str) -> None:
        self.model.to(device)

    def infer_sample(self, sample, **kwargs):
        sample = sample if isinstance(sample, (list, np.ndarray)) else list(sample)
        h_frame, w_frame = sample[0].shape[:2]
        _, kpts, counts, _, _, _ = self.model.infer_frames(sample, self.CONF, self.IOU, self.train_image_size, classes=[0],
                                                          channel_reverse=True, pil_stretch=True, reuse_outputs=self._reuse_outputs)
        return kpts, counts, (h_frame, w_frame)

    def submit_sample(self, sample, **kwargs):
        sample = sample if isinstance(sample, (list, np.ndarray)) else list(sample)
        if not hasattr(self.model, "submit_frames"):
            return None
        token = self.model.submit_frames(sample, self.CONF, self.IOU, self.train_image_size, classes=[0], channel_reverse=True,
                                         pil_stretch=True)
        if token is not None:
            token["frame_hw"] = sample[0].shape[:2]
        return token

    def collect_sample(self, token):
        _, kpts, counts, _, _, _ = self.model.collect_frames(token)
        return kpts, counts, tuple(token["frame_hw"])

    def post_sample(self, raw, **kwargs) -> list:
        kpts, counts, (h_frame, w_frame) = raw
        ratio = (w_frame / self.train_image_size, h_frame / self.train_image_size)     # reference :276-278
        K, ndim = self.model.kpt_shape
        predictions = []
        for i in range(len(counts)):
            k = kpts[i, :counts[i]].reshape(counts[i], K, ndim)
            xy = k[..., :2].copy()
            if ndim == 3:
                xy[k[..., 2] < 0.5] = 0          # [upstream] Keypoints.xy zeroes points predicted invisible
            predictions.append(PlayersKeypoints(xy=xy, ratio=ratio))
        return predictions

    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs) -> list:
        return self.post_sample(self.infer_sample(sample, **kwargs), **kwargs)

    # ---- sharded: a frame's partial is its (n, K, 2) array of network-input coordinates (invisible points zeroed) with the
    # frame's ratio in a trailing row — arrays on the wire (Tracker.pack_partials), the containers are built on rank 0
    def predict_partial(self, frame_generator, *, first_frame: int = 0, head_context: int = 0, tail_context: int = 0,
                        **kwargs) -> list:
        assert head_context == 0 and tail_context == 0
        K, ndim = self.model.kpt_shape
        out = []
        for kpts, counts, (h_frame, w_frame) in self._raw_batches(frame_generator):
            ratio = np.array([w_frame / self.train_image_size, h_frame / self.train_image_size], np.float64)
            for i in range(len(counts)):
                k = kpts[i, :counts[i]].reshape(counts[i], K, ndim)
                xy = k[..., :2].astype(np.float64)                 # (float32 values, exactly: the ratio row needs doubles)
                if ndim == 3:
                    xy[k[..., 2] < 0.5] = 0
                out.append(np.concatenate([xy.reshape(counts[i], K * 2), np.tile(ratio, K)[None, :]], axis=0))
        return out

    def merge_partials(self, partials: list, **kwargs) -> list:
        """Partials: ``predict_partial``'s (n + 1, 2 K) float64 arrays, or — after the wire — (float32 (n, 2 K) view, (rx, ry))."""
        K, _ = self.model.kpt_shape
        return [PlayersKeypoints(xy=p[0].reshape(-1, K, 2), ratio=p[1]) if isinstance(p, tuple) else
                PlayersKeypoints(xy=p[:-1].reshape(-1, K, 2).astype(np.float32), ratio=(float(p[-1, 0]), float(p[-1, 1])))
                for p in partials]

    # on the wire (round 6): the keypoint coordinates are float32 VALUES carried in float64 rows (the ratio row needs doubles) — they
    # travel as float32 (lossless: half the bytes of the default ragged pack, 1.9 MB instead of 3.8 MB per rank and 64-frame batch at
    # the bench's 283 persons per frame), the ratio rows as one (frames, 2) float64 array beside them
    def pack_partials(self, partials: list):
        K, _ = self.model.kpt_shape
        counts = np.array([len(p) - 1 for p in partials], np.int64)
        rows = np.empty((int(counts.sum()), 2 * K), np.float32)
        pos = 0
        for p, c in zip(partials, counts):             # one conversion per frame straight into the send buffer
            rows[pos:pos + c] = p[:-1]
            pos += int(c)
        ratios = np.array([p[-1, :2] for p in partials], np.float64).reshape(len(partials), 2)
        return [counts, rows, ratios]

    def unpack_partials(self, arrays: list) -> list:
        counts, rows, ratios = arrays                  # views into the received buffer: no per-frame copies on rank 0
        ends = np.cumsum(counts)
        return [(rows[int(e) - int(c):int(e)], (float(ratios[i, 0]), float(ratios[i, 1]))) for i, (c, e) in enumerate(zip(counts, ends))]

    def predict_frames(self, frame_generator, **kwargs):
        raise NoPredictFrames()
