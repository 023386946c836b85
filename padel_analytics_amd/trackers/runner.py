"""TrackingRunner — drop-in for the tracker loop of the reference's ``trackers/runner.py``
(``__init__`` :37-79, ``run`` :175-236): trackers run one after the other over the whole clip, each is
moved to the device, timed with ``timeit`` exactly where the reference times it (:222-228), moved back,
and its predictions are cached as JSON; a tracker with cached predictions is skipped (:187-191).

Additions: ``run()`` also records ``self.timings`` (seconds / frames per tracker) and prints frames/s.
``draw_and_collect_data`` (video encode, homography, analytics: SURVEY.md §1 L3') is out of scope."""
from __future__ import annotations

import timeit
from pathlib import Path
from typing import Optional

from .. import video
from .tracker import Tracker


class TrackingRunner:
    def __init__(self, trackers: list, video_path: str | Path, inference_path: str | Path, start: int = 0,
                 end: Optional[int] = None, collect_data: bool = False) -> None:
        self.video_path = video_path
        self.inference_path = inference_path
        self.start = start
        self.stride = 1
        self.end = end
        self.video_info = video.VideoInfo.from_video_path(video_path)
        self.total_frames = self.video_info.total_frames if end is None else end - start
        self.trackers = {}
        for tracker in trackers:
            self.trackers[str(tracker)] = tracker.video_info_post_init(self.video_info)
        self.collect_data = collect_data
        self.data_analytics = None
        self.timings: dict = {}

    def restart(self) -> None:
        for tracker in self.trackers.values():
            tracker.restart()

    def draw_and_collect_data(self) -> None:
        print("runner: drawing / data collection is outside the hot path of this build (skipped)")

    def run(self) -> None:
        print(f"runner: Running {self.total_frames} frames")
        for tracker in self.trackers.values():
            if len(tracker) != 0:
                print(f"{tracker.__str__()}: {len(tracker)} predictions stored")
                continue
            tracker.to(tracker.DEVICE)
            print(f"{str(tracker)}: Running on {tracker.DEVICE} ...")
            frame_generator = video.get_video_frames_generator(self.video_path, start=self.start, stride=self.stride,
                                                               end=self.end)
            t0 = timeit.default_timer()
            tracker.predict_and_update(frame_generator, total_frames=self.total_frames)
            t1 = timeit.default_timer()
            tracker.to("cpu")
            print(f"{str(tracker)}: {t1 - t0} inference time.")
            n = len(tracker)
            self.timings[str(tracker)] = {"seconds": t1 - t0, "frames": n}
            if t1 > t0:
                print(f"{str(tracker)}: {n / (t1 - t0):.1f} frames/s")
            tracker.save_predictions()
        self.draw_and_collect_data()
