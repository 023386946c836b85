"""Drop-in replacements for the reference's ``trackers`` package exports (``trackers/__init__.py:1-7``)."""
from .tracker import NoPredictFrames, NoPredictSample, Object, Tracker, TrackingResults
from .players_tracker import Player, Players, PlayerTracker
from .players_keypoints_tracker import PlayerKeypoint, PlayerKeypoints, PlayersKeypoints, PlayerKeypointsTracker
from .ball_tracker import Ball, BallTracker
from .ball_detect_tracker import BallDetectTracker
from .keypoints_tracker import Keypoint, Keypoints, KeypointsTracker
from .runner import TrackingRunner


def set_eager_objects(eager: bool = True) -> None:
    """``True``: ``Players`` / ``PlayersKeypoints`` build their ``Player`` / ``PlayerKeypoints`` objects inside
    ``predict_sample`` exactly where the reference does (``players_tracker.py:371-378``,
    ``players_keypoints_tracker.py:303-320``); ``False`` (default): on first access — same objects, same values, created
    when a consumer touches them.  Also read from ``PADEL_EAGER_OBJECTS=1`` at import."""
    Players.EAGER = bool(eager)
    PlayersKeypoints.EAGER = bool(eager)


import os as _os
if _os.environ.get("PADEL_EAGER_OBJECTS") == "1":
    set_eager_objects(True)
