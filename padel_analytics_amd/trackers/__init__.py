"""Drop-in replacements for the reference's ``trackers`` package exports (``trackers/__init__.py:1-7``)."""
from .tracker import NoPredictFrames, NoPredictSample, Object, Tracker, TrackingResults
from .players_tracker import Player, Players, PlayerTracker
from .players_keypoints_tracker import PlayerKeypoint, PlayerKeypoints, PlayersKeypoints, PlayerKeypointsTracker
from .ball_tracker import Ball, BallTracker
from .ball_detect_tracker import BallDetectTracker
from .keypoints_tracker import Keypoint, Keypoints, KeypointsTracker
from .runner import TrackingRunner


def set_eager_objects(eager=True) -> None:
    """``True`` / 1: ``Players`` / ``PlayersKeypoints`` build their ``Player`` / ``PlayerKeypoints`` objects inside
    ``predict_sample`` exactly where the reference does (``players_tracker.py:371-378``,
    ``players_keypoints_tracker.py:303-320``) — array-backed, ~0.5 us each; a person's 13 ``PlayerKeypoint`` records and a
    player's one-row ``Detections`` still appear on first access.  2: those as well (every object the reference allocates,
    where it allocates it).  ``False`` (default): everything on first access — same objects, same values, created when a
    consumer touches them.  Also read from ``PADEL_EAGER_OBJECTS=1|2`` at import."""
    level = 2 if eager == 2 else (1 if eager else 0)
    Players.EAGER = bool(level)
    PlayersKeypoints.EAGER = level


import os as _os
if _os.environ.get("PADEL_EAGER_OBJECTS") in ("1", "2"):
    set_eager_objects(int(_os.environ["PADEL_EAGER_OBJECTS"]))
