"""Seeded synthetic inputs: there is no sample video in the reference checkout
(`.MISSING_LARGE_BLOBS:1-2`), so frames are generated: uint8 HWC **BGR** like
`sv.get_video_frames_generator` yields (`trackers/runner.py:215-220`), a low-frequency background plus
a few rectangles so activations sit in a realistic range (not white noise).

TEST INFRASTRUCTURE (moved out of the product package in round 4).  Importing this module registers the
``synthetic://?n=64&h=720&w=1280&fps=30&seed=0`` frame source with ``padel_analytics_amd.video``."""
from __future__ import annotations

import numpy as np


def synthetic_frames(n: int, h: int, w: int, seed: int = 0, return_rects: bool = False):
    """``return_rects``: also the painted rectangles per frame, (x0, y0, x1, y1) in frame pixels in paint order (a later one covers
    an earlier one) — the ground truth the least-squares head fit of oracle/synth_weights.py regresses."""
    rng = np.random.default_rng(seed)
    rects = []
    ys = np.linspace(0, 1, h, dtype=np.float32)[:, None]
    xs = np.linspace(0, 1, w, dtype=np.float32)[None, :]
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        img = np.empty((h, w, 3), np.float32)
        for c in range(3):
            a, b, p, q = rng.uniform(0.5, 3.0, 4)
            ph = rng.uniform(0, 6.28, 2)
            img[..., c] = 120 + 60 * np.sin(a * 6.28 * xs + ph[0]) * np.cos(b * 6.28 * ys + ph[1]) \
                + 30 * np.sin(p * 6.28 * (xs + ys)) + 10 * np.cos(q * 12.56 * (xs - ys))
        rects.append([])
        for _ in range(int(rng.integers(4, 9))):
            rh, rw = int(rng.integers(h // 12, h // 3)), int(rng.integers(w // 24, w // 6))
            y0, x0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
            img[y0:y0 + rh, x0:x0 + rw] = rng.uniform(0, 255, 3).astype(np.float32)
            rects[-1].append((x0, y0, x0 + rw, y0 + rh))
        img += rng.normal(0, 3.0, img.shape).astype(np.float32)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return (out, rects) if return_rects else out


def _query(p: str) -> dict:
    from urllib.parse import parse_qs, urlparse
    q = {k: int(v[0]) for k, v in parse_qs(urlparse(p).query).items()}
    return {"n": q.get("n", 64), "h": q.get("h", 720), "w": q.get("w", 1280), "fps": q.get("fps", 30), "seed": q.get("seed", 0)}


def _info(p: str):
    from padel_analytics_amd import video
    q = _query(p)
    return video.VideoInfo(q["w"], q["h"], q["fps"], q["n"])


def _frames(p: str, start: int, end, stride: int):
    q = _query(p)
    stop = q["n"] if end is None else min(end, q["n"])
    for i in range(start, stop, stride):
        yield synthetic_frames(1, q["h"], q["w"], seed=q["seed"] * 100003 + i)[0]


def _register():
    from padel_analytics_amd import video
    video.register_source("synthetic", _info, _frames)


_register()
