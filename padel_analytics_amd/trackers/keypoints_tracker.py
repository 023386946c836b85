"""Court keypoints tracker — drop-in for the reference's ``trackers/keypoints_tracker/keypoints_tracker.py``
(``Keypoint`` :17-66, ``Keypoints`` :69-117, ``KeypointsTracker`` :122-312), SURVEY.md §8(f)#4.

In the shipped configuration the reference never runs a model here: ``main.py:81-104,154-161`` always passes
``fixed_keypoints_detection`` and both predict methods short-circuit (:204-209, :266-271).  The "yolo" model
type (:199-262) is just another YOLOv8-pose graph (K = 12 keypoints, ``max_det = 12``, conf .5, Pillow stretch
to 640x640) and runs on the same HIP engine as the players keypoints tracker.  The "resnet" type (a
torchvision ResNet-50 regressor, :158-168, :273-312) is outside the engine's op set and raises.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional, Type

import numpy as np

from .tracker import NoPredictFrames, Object, Tracker


class Keypoint:
    def __init__(self, id: int, xy: tuple):
        self.id = id
        self.xy = xy

    @classmethod
    def from_json(cls, x: dict): return cls(**x)

    def serialize(self) -> dict: return {"id": self.id, "xy": self.xy}

    def asint(self) -> tuple: return tuple(int(v) for v in self.xy)


class Keypoints(Object):
    def __init__(self, keypoints: list):
        super().__init__()
        self.keypoints = sorted(keypoints, key=lambda k: k.id)
        self.keypoints_by_id = {k.id: k for k in keypoints}

    @classmethod
    def from_json(cls, x: list) -> "Keypoints":
        return cls([Keypoint.from_json(k) for k in x])

    def serialize(self) -> list: return [k.serialize() for k in self.keypoints]

    def __len__(self) -> int: return len(self.keypoints)

    def __iter__(self): return iter(self.keypoints)

    def __getitem__(self, id: int) -> Keypoint: return self.keypoints_by_id[id]


class KeypointsTracker(Tracker):
    NUMBER_KEYPOINTS = 12
    TRAIN_IMAGE_SIZE = 640
    CONF = 0.5
    IOU = 0.7
    # model output index -> court keypoint id (reference :215-228)
    POINTS_MAPPER = {0: 10, 1: 11, 2: 1, 3: 0, 4: 7, 5: 9, 6: 8, 7: 5, 8: 6, 9: 2, 10: 4, 11: 3}

    def __init__(self, model_path: str, batch_size: int, model_type: str = "resnet",
                 fixed_keypoints_detection: Optional[Keypoints] = None,
                 load_path: Optional[str | Path] = None, save_path: Optional[str | Path] = None):
        super().__init__(load_path=load_path, save_path=save_path)
        if model_type not in ("resnet", "yolo"):
            raise ValueError("Unknown model type")
        self.batch_size = batch_size
        self.model_type = model_type
        self.model_path = model_path
        self.fixed_keypoints_detection = fixed_keypoints_detection
        self._model = None

    @property
    def model(self):
        if self._model is None:
            if self.model_type != "yolo":
                raise NotImplementedError("the ResNet-50 court-keypoint regressor is outside this engine's op set; "
                                          "use model_type='yolo' or fixed_keypoints_detection")
            from ..yolo import YOLO
            self._model = YOLO(self.model_path)
        return self._model

    def video_info_post_init(self, video_info) -> "KeypointsTracker": return self

    def object(self) -> Type[Object]: return Keypoints

    def draw_kwargs(self) -> dict: return {}

    def __str__(self) -> str: return "keypoints_tracker"

    def restart(self) -> None: self.results.restart()

    def to(self, device: str) -> None:
        if self.fixed_keypoints_detection is None:
            self.model.to(device)

    def predict_sample(self, sample: Iterable[np.ndarray], **kwargs) -> list:
        sample = list(sample)
        if self.fixed_keypoints_detection is not None:
            print(f"{self.__str__()}: using fixed court keypoints")
            return [self.fixed_keypoints_detection for _ in sample]
        h_frame, w_frame = sample[0].shape[:2]
        ratio_x, ratio_y = w_frame / self.TRAIN_IMAGE_SIZE, h_frame / self.TRAIN_IMAGE_SIZE
        results = self.model.predict_frames(sample, self.CONF, self.IOU, self.TRAIN_IMAGE_SIZE, classes=None,
                                            max_det=self.NUMBER_KEYPOINTS, channel_reverse=True, pil_stretch=True)
        predictions = []
        for result in results:
            xy = result.keypoints.xy
            kps = []
            if len(xy):                                   # the reference assumes exactly one court detection
                for i, kp in enumerate(xy[0]):
                    kps.append(Keypoint(id=self.POINTS_MAPPER.get(i, i), xy=(float(kp[0]) * ratio_x, float(kp[1]) * ratio_y)))
            predictions.append(Keypoints(kps))
        return predictions

    def predict_frames(self, frame_generator: Iterable[np.ndarray], **kwargs) -> list:
        if self.fixed_keypoints_detection is not None:
            print(f"{self.__str__()}: using fixed court keypoints")
            return [self.fixed_keypoints_detection for _ in frame_generator]
        if self.model_type == "yolo":
            raise NoPredictFrames()
        raise NotImplementedError("ResNet-50 court keypoints are not supported by this engine")
