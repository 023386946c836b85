"""Pins the ball-path oracle (oracle/ball_ref.py) on CPU: the ensemble against a literal transcription of
the reference's streaming buffer loop (ball_tracker.py:421-523) for several batch sizes, the ensemble weights,
the heat-map decode rules, and the product's host-side rectangle pick against the oracle's."""
import numpy as np
import torch

from oracle import ball_ref as br
from padel_analytics_amd.trackers.ball_tracker import Ball, predict_location


def _reference_stream_ensemble(y_all: np.ndarray, batch: int) -> np.ndarray:
    """Transcription of the reference's loop over DataLoader batches with its 7-deep prediction buffer."""
    seq_len, H, W = 8, y_all.shape[2], y_all.shape[3]
    video_len = y_all.shape[0] + 7
    num_sample, sample_count = video_len - seq_len + 1, 0
    buffer_size = seq_len - 1
    sample_indices = torch.arange(seq_len)
    frame_indices = torch.arange(seq_len - 1, -1, -1)
    y_pred_buffer = torch.zeros((buffer_size, seq_len, H, W), dtype=torch.float32)
    weight = torch.from_numpy(br.ensemble_weight())
    out = []
    for b0 in range(0, y_all.shape[0], batch):
        y_pred = torch.from_numpy(y_all[b0:b0 + batch])
        bs = y_pred.shape[0]
        y_pred_buffer = torch.cat((y_pred_buffer, y_pred), 0)
        for sample_i in range(bs):
            if sample_count < buffer_size:
                y = y_pred_buffer[sample_indices + sample_i, frame_indices].sum(0) / (sample_count + 1)
            else:
                y = (y_pred_buffer[sample_indices + sample_i, frame_indices] * weight[:, None, None]).sum(0)
            out.append(y)
            sample_count += 1
            if sample_count == num_sample:
                y_pred_buffer = torch.cat((y_pred_buffer, torch.zeros((buffer_size, seq_len, H, W))), 0)
                for frame_i in range(1, seq_len):
                    out.append(y_pred_buffer[sample_indices + sample_i + frame_i, frame_indices].sum(0) / (seq_len - frame_i))
        y_pred_buffer = y_pred_buffer[-buffer_size:]
    return torch.stack(out).numpy()


def test_ensemble_weight():
    assert np.allclose(br.ensemble_weight(), np.array([1, 2, 3, 4, 4, 3, 2, 1], np.float32) / 20)


def test_ensemble_matches_reference_stream_loop():
    rng = np.random.default_rng(0)
    for nw, batch in ((1, 1), (5, 2), (9, 4), (20, 8), (13, 3)):
        y = rng.uniform(0, 1, (nw, 8, 6, 10)).astype(np.float32)
        want = _reference_stream_ensemble(y, batch)
        got = br.ensemble(y)
        assert got.shape == (nw + 7, 6, 10)
        assert np.array_equal(got, want), (nw, batch)


def test_decode_rules():
    m = np.zeros((288, 512), np.uint8)
    assert br.predict_location(m) == (0, 0, 0, 0) and predict_location(m) == (0, 0, 0, 0)
    m[10:14, 20:29] = 255            # 9 x 4 = 36
    m[100:103, 300:303] = 255        # 3 x 3 = 9
    m[103, 303] = 255                # diagonal neighbour: 8-connected -> same component, rect 4 x 4 = 16
    assert br.predict_location(m) == (20, 10, 9, 4) == predict_location(m)
    heat = np.zeros((2, 288, 512), np.float32)
    heat[0, 10:14, 20:29] = 0.9
    heat[1, 0, 0] = 0.5              # not > 0.5
    x, y, v = br.decode_heat(heat, (1280 / 512, 720 / 288))
    assert (x[0], y[0], v[0]) == (int(24 * 2.5), int(12 * 2.5), 1) and (x[1], y[1], v[1]) == (0, 0, 0)


def test_ball_json():
    b = Ball(frame=3, xy=(10, 20), visibility=1)
    assert b.serialize() == {"frame": 3, "xy": (10, 20), "visibility": 1, "projection": None}
    assert Ball.from_json(b.serialize()).xy == (10, 20) and b.asint() == (10, 20)
