"""Players tracker — drop-in for the reference's ``trackers/players_tracker/players_tracker.py``
(``Player`` :14-98, ``Players`` :199-231, ``PlayerTracker`` :266-383).

Hot path (``predict_sample`` :341-380): a batch of BGR frames -> YOLOv8 detect (conf .5, iou .7, imgsz
640, ``classes=[0]``) -> polygon-zone filter -> ByteTrack ids -> ``Players``.  The network, decode, NMS and
box rescale run in the HIP engine; the reference's host ``processor`` (BGR2RGB, :335-336) followed by
upstream's own channel flip is the identity on channel order, so raw frames go to the device with
``channel_reverse=False``.  Drawing is out of scope.

Host side: the polygon-zone test runs once per batch on the flattened box array, ByteTrack is the host-native
``pa_bytetrack_*`` (one call per batch of frames), and ``Players`` materialises its ``Player`` objects on first
access — a 64-frame batch carries thousands of boxes on the synthetic bench weights and must not cost more host
time than its ~25 ms of GPU time.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional, Type

import numpy as np

from ..detections import Detections
from ..yolo import YOLO
from .tracker import NoPredictFrames, Object, Tracker


class Player:
    """One tracked person of one frame (reference :14-98): ``xyxy``, ``id``, ``class_id``, ``confidence``, ``projection``,
    ``detection`` and the geometry properties.  ``__slots__`` + ``Player.from_row``: the tracker builds these straight from
    the detector's arrays (a view of the box row, Python scalars for the rest: ~1 us each) — the one-row ``Detections`` the
    reference's constructor takes is created only when ``.detection`` is read."""
    __slots__ = ("xyxy", "id", "class_id", "confidence", "projection", "_detection", "_tid")

    def __init__(self, detection: Detections, projection: Optional[tuple] = None):
        self._detection = detection
        self.projection = projection
        self.xyxy = detection.xyxy[0]
        tid = detection.tracker_id
        # reference :32-36 uses array truthiness: a single id 0 reads as None
        self.id = int(tid[0]) if (tid is not None and len(tid) and bool(np.asarray(tid).any())) else None
        self._tid = None                                   # (only from_row objects rebuild their Detections from it)
        self.class_id = int(detection.class_id[0])
        self.confidence = float(detection.confidence[0])

    @classmethod
    def from_row(cls, xyxy: np.ndarray, confidence: float, class_id: int, tracker_id: Optional[int]) -> "Player":
        """The same object as ``Player(Detections(one row))`` from the row's pieces (``xyxy``: a (4,) float32 array)."""
        p = object.__new__(cls)
        p.xyxy = xyxy
        p.confidence = confidence
        p.class_id = class_id
        p._tid = tracker_id
        p.id = tracker_id if tracker_id else None          # the reference's truthiness rule: id 0 reads as None
        p.projection = None
        p._detection = None
        return p

    @property
    def detection(self) -> Detections:
        if self._detection is None:
            self._detection = Detections(xyxy=self.xyxy[None, :], confidence=np.array([self.confidence], np.float32),
                                         class_id=np.array([self.class_id]),
                                         tracker_id=None if self._tid is None else np.array([self._tid]))
        return self._detection

    @property
    def top_left(self) -> tuple: return tuple(int(p) for p in self.xyxy[:2])

    @property
    def bottom_right(self) -> tuple: return tuple(int(p) for p in self.xyxy[2:])

    @property
    def height(self): return self.bottom_right[1] - self.top_left[1]

    @property
    def width(self): return self.bottom_right[0] - self.top_left[0]

    @property
    def midpoint(self) -> tuple:
        return int(self.top_left[0] + self.width / 2), int(self.top_left[1] + self.height / 2)

    @property
    def feet(self) -> tuple:
        return int(self.top_left[0] + self.width / 2), int(self.bottom_right[1])

    @classmethod
    def from_json(cls, x: dict) -> "Player":
        det = Detections(xyxy=np.array([x["xyxy"]]), confidence=np.array([x["confidence"]]),
                         class_id=np.array([x["class_id"]]), tracker_id=np.array([x["id"]]))
        return cls(detection=det, projection=x.get("projection"))

    def serialize(self) -> dict:
        return {"id": self.id, "xyxy": [float(p) for p in self.xyxy], "projection": self.projection,
                "class_id": self.class_id, "confidence": self.confidence}


class Players(Object):
    """List of ``Player`` (reference :199-231).  Built either from ``Player`` objects (reference signature) or from
    the tracker's arrays (``rows`` (k, 6) x1,y1,x2,y2,conf,cls + ``ids`` (k,)); the ``Player`` objects are then created
    in the constructor (``EAGER``: where the reference creates them) or on first access."""

    #: True: build the ``Player`` objects in the constructor, i.e. inside ``predict_sample`` like the reference does
    #: (:371-378); False (default): on first access.  ``trackers.set_eager_objects`` flips it for every container class.
    EAGER = False

    def __init__(self, players: Optional[list] = None, *, rows: Optional[np.ndarray] = None,
                 ids: Optional[np.ndarray] = None):
        super().__init__()
        self._players = players
        self._rows, self._ids = rows, ids
        if players is None and rows is None:
            self._players = []
        if self.EAGER:
            self.players

    @property
    def players(self) -> list:
        if self._players is None:
            r, t = self._rows, self._ids
            n = len(r)
            # one C call per column instead of one numpy scalar conversion per field and object
            conf = r[:, 4].tolist()
            cls_ = r[:, 5].astype(int).tolist()
            tids = [None] * n if t is None else np.asarray(t).astype(int).tolist()
            new = Player.from_row
            self._players = [new(r[i, :4], conf[i], cls_[i], tids[i]) for i in range(n)]
        return self._players

    @classmethod
    def from_json(cls, x: list) -> "Players":
        return cls([Player.from_json(p) for p in x])

    def serialize(self) -> list:
        if self._players is None:                  # straight from the arrays (same values as Player.serialize)
            r, t = self._rows, self._ids
            return [{"id": (int(t[i]) if (t is not None and t[i] != 0) else None), "xyxy": [float(v) for v in r[i, :4]],
                     "projection": None, "class_id": int(r[i, 5]), "confidence": float(r[i, 4])} for i in range(len(r))]
        return [p.serialize() for p in self._players]

    def __len__(self) -> int: return len(self._rows) if self._players is None else len(self._players)

    def __iter__(self): return iter(self.players)

    def __getitem__(self, i: int) -> Player: return self.players[i]


class PlayerTracker(Tracker):
    CONF = 0.5
    IOU = 0.7
    IMGSZ = 640
    streams = False

    def __init__(self, model_path: str, polygon_zone, batch_size: int, annotator: str = "rectangle_bounding_box",
                 show_confidence: bool = True, load_path: Optional[str | Path] = None,
                 save_path: Optional[str | Path] = None, half: bool = False):
        super().__init__(load_path=load_path, save_path=save_path)
        self.model = YOLO(model_path, half=half)
        self.polygon_zone = polygon_zone
        self.batch_size = batch_size
        self.annotator = annotator
        self.show_confidence = show_confidence
        self.video_info = None
        self.byte_track = None

    def video_info_post_init(self, video_info) -> "PlayerTracker":
        from ..engine import NativeByteTrack
        self.video_info = video_info
        self.byte_track = NativeByteTrack(frame_rate=video_info.fps)     # reference :311
        return self

    def object(self) -> Type[Object]: return Players

    def draw_kwargs(self) -> dict:
        return {"video_info": self.video_info, "annotator": self.annotator, "show_confidence": self.show_confidence}

    def __str__(self) -> str: return "players_tracker"

    def restart(self) -> None:
        self.results.restart()
        print(f"{self.__str__()}: Byte tracker reset")
        if self.byte_track is not None:
            self.byte_track.reset()

    def processor(self, frame: np.ndarray) -> np.ndarray:
        """cv2.cvtColor(frame, COLOR_BGR2RGB) (reference :335-336); kept for API parity — the hot path
        folds it into the device preprocessing."""
        return frame[..., ::-1]

    def to(self, device: str) -> None:
        self.model.to(device)

    # ---- device stage: detector over one batch of raw BGR frames (host arrays or HBM-resident DeviceFrames)
    def infer_sample(self, sample, **kwargs):
        boxes, _, counts, _, _, _ = self.model.infer_frames(sample, self.CONF, self.IOU, self.IMGSZ, classes=[0],
                                                            channel_reverse=False, reuse_outputs=self._reuse_outputs)
        return boxes, counts

    def submit_sample(self, sample, **kwargs):
        if not hasattr(self.model, "submit_frames"):
            return None
        return self.model.submit_frames(sample, self.CONF, self.IOU, self.IMGSZ, classes=[0], channel_reverse=False)

    def collect_sample(self, token):
        boxes, _, counts, _, _, _ = self.model.collect_frames(token)
        return boxes, counts

    # ---- host stage, split in the stateless part (zone) and the sequential part (ByteTrack)
    def _zone_keep(self, boxes: np.ndarray, counts: np.ndarray) -> np.ndarray:
        """(n, max_det) bool: rows that exist and (with a zone) whose bottom-centre anchor lies inside it
        (reference :364-365, once per batch instead of once per frame)."""
        valid = np.arange(boxes.shape[1])[None, :] < counts[:, None]
        if self.polygon_zone is None:
            return valid
        keep = np.zeros_like(valid)
        keep[valid] = self.polygon_zone.trigger_boxes(boxes[valid][:, :4])
        return keep

    def _track(self, boxes: np.ndarray, counts: np.ndarray, keep: Optional[np.ndarray]) -> list:
        """Frame-sequential ids (reference :367-369) + result objects (:371-378) for a batch of frames."""
        n = len(counts)
        valid = np.arange(boxes.shape[1])[None, :] < counts[:, None]
        sel = valid if keep is None else (valid & keep.astype(bool))
        if self.byte_track is not None:
            ids = self.byte_track.update_batch(boxes, counts, None if keep is None else keep)
            sel = sel & (ids >= 0)
        else:
            ids = None
        # one gather for the whole batch, then per-frame views (row-major boolean indexing keeps frame and row order)
        rows = boxes[sel]
        cuts = np.cumsum(sel.sum(axis=1))[:-1]
        rows_f = np.split(rows, cuts)
        ids_f = [None] * n if ids is None else np.split(ids[sel].astype(int), cuts)
        return [Players(rows=rows_f[i], ids=ids_f[i]) for i in range(n)]

    def post_sample(self, raw, **kwargs) -> list:
        boxes, counts = raw
        return self._track(boxes, counts, self._zone_keep(boxes, counts))

    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs) -> list:
        return self.post_sample(self.infer_sample(sample, **kwargs), **kwargs)

    # ---- sharded: per-frame kept rows travel to rank 0, ByteTrack runs there in global frame order
    def predict_partial(self, frame_generator, *, first_frame: int = 0, head_context: int = 0, tail_context: int = 0,
                        **kwargs) -> list:
        assert head_context == 0 and tail_context == 0
        out = []
        for boxes, counts in self._raw_batches(frame_generator):     # (boolean indexing copies: nothing of a batch's arrays is kept)
            keep = self._zone_keep(boxes, counts)
            out += [boxes[i, keep[i]] for i in range(len(counts))]
        return out

    def merge_partials(self, partials: list, **kwargs) -> list:
        out = []
        for lo in range(0, len(partials), max(1, self.batch_size)):
            part = partials[lo:lo + max(1, self.batch_size)]
            stride = max(1, max(len(r) for r in part))
            boxes = np.zeros((len(part), stride, 6), np.float32)
            counts = np.array([len(r) for r in part], np.int32)
            for i, r in enumerate(part):
                boxes[i, :len(r)] = r
            out += self._track(boxes, counts, None)
        return out

    def predict_frames(self, frame_generator, **kwargs):
        raise NoPredictFrames()
