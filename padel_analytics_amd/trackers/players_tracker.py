"""Players tracker — drop-in for the reference's ``trackers/players_tracker/players_tracker.py``
(``Player`` :14-98, ``Players`` :199-231, ``PlayerTracker`` :266-383).

Hot path (``predict_sample`` :341-380): a batch of BGR frames -> YOLOv8 detect (conf .5, iou .7, imgsz
640, ``classes=[0]``) -> polygon-zone filter -> ByteTrack ids -> ``Players``.  The network, decode, NMS and
box rescale run in the HIP engine; the reference's host ``processor`` (BGR2RGB, :335-336) followed by
upstream's own channel flip is the identity on channel order, so raw frames go to the device with
``channel_reverse=False``.  Drawing is out of scope.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional, Type

import numpy as np

from ..detections import Detections
from ..yolo import YOLO
from .tracker import NoPredictFrames, Object, Tracker


class Player:
    def __init__(self, detection: Detections, projection: Optional[tuple] = None):
        self.detection = detection
        self.projection = projection
        self.xyxy = detection.xyxy[0]
        tid = detection.tracker_id
        # reference :32-36 uses array truthiness: a single id 0 reads as None
        self.id = int(tid[0]) if (tid is not None and len(tid) and bool(np.asarray(tid).any())) else None
        self.class_id = int(detection.class_id[0])
        self.confidence = float(detection.confidence[0])

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
    def __init__(self, players: list):
        super().__init__()
        self.players = players

    @classmethod
    def from_json(cls, x: list) -> "Players":
        return cls([Player.from_json(p) for p in x])

    def serialize(self) -> list:
        return [p.serialize() for p in self.players]

    def __len__(self) -> int: return len(self.players)

    def __iter__(self): return iter(self.players)

    def __getitem__(self, i: int) -> Player: return self.players[i]


class PlayerTracker(Tracker):
    CONF = 0.5
    IOU = 0.7
    IMGSZ = 640

    def __init__(self, model_path: str, polygon_zone, batch_size: int, annotator: str = "rectangle_bounding_box",
                 show_confidence: bool = True, load_path: Optional[str | Path] = None,
                 save_path: Optional[str | Path] = None):
        super().__init__(load_path=load_path, save_path=save_path)
        self.model = YOLO(model_path)
        self.polygon_zone = polygon_zone
        self.batch_size = batch_size
        self.annotator = annotator
        self.show_confidence = show_confidence
        self.video_info = None
        self.byte_track = None

    def video_info_post_init(self, video_info) -> "PlayerTracker":
        from ..bytetrack import ByteTrack
        self.video_info = video_info
        self.byte_track = ByteTrack(frame_rate=video_info.fps)
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

    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs) -> list:
        results = self.model.predict_frames(sample, self.CONF, self.IOU, self.IMGSZ, classes=[0], channel_reverse=False)
        predictions = []
        for result in results:            # sequential in frame order: ByteTrack is stateful
            det = Detections.from_ultralytics(result)
            if self.polygon_zone is not None:
                det = det[self.polygon_zone.trigger(det)]
            if self.byte_track is not None:
                det = self.byte_track.update_with_detections(detections=det)
            predictions.append(Players([Player(detection=det[i]) for i in range(len(det))]))
        return predictions

    def predict_frames(self, frame_generator, **kwargs):
        raise NoPredictFrames()
