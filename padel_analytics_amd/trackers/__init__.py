"""Drop-in replacements for the reference's ``trackers`` package exports (``trackers/__init__.py:1-7``)."""
from .tracker import NoPredictFrames, NoPredictSample, Object, Tracker, TrackingResults
from .players_tracker import Player, Players, PlayerTracker
from .players_keypoints_tracker import PlayerKeypoint, PlayerKeypoints, PlayersKeypoints, PlayerKeypointsTracker
from .ball_tracker import Ball, BallTracker
from .ball_detect_tracker import BallDetectTracker
from .keypoints_tracker import Keypoint, Keypoints, KeypointsTracker
from .runner import TrackingRunner
