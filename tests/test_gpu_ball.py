"""GPU: the streaming ball session (pa_ball_*: Pillow resize, window assembly, TrackNet, temporal ensemble,
threshold) and the BallTracker plugin against the oracle (oracle/ball_ref.py + oracle/tracknet_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import ball_ref as br, tracknet_ref as tr
from padel_analytics_amd import checkpoint, engine as E, graph as G, video
from tests import synth
from padel_analytics_amd.trackers import BallTracker

pytestmark = pytest.mark.gpu


def _clip(T, h, w, seed):
    frames = synth.synthetic_frames(T, h, w, seed=seed)
    # a bright moving blob so the clip is not static
    for t in range(T):
        cy, cx = h // 3 + 5 * t, w // 4 + 17 * t
        frames[t, cy:cy + 14, cx:cx + 14] = (255, 255, 255)
    return frames


def _calibrated_tracknet(frames):
    """TrackNet weights whose sigmoid output crosses 0.5 on a small fraction of pixels of `frames`."""
    sd = tr.synth_tracknet_state_dict(5)
    med = br.median_background(frames)
    small = [br.resize_frame(f) for f in frames[:8]]
    x = torch.from_numpy(br.window_input(med, small))[None]
    net = tr.TrackNetRef(sd)
    # pre-sigmoid logits of the predictor: shift the bias so ~1 % of pixels are above threshold
    sd["predictor.bias"] = np.zeros(8, np.float32)
    y = net.forward(x)
    logit = torch.log(y / (1 - y))
    sd["predictor.bias"] = (-torch.quantile(logit.flatten(), 0.99)).repeat(8).numpy().astype(np.float32)
    return sd


@pytest.mark.parametrize("T,feed", [(19, 4), (12, 8)])
def test_ball_session_matches_oracle(gpu_engine, T, feed):
    frames = _clip(T, 360, 640, seed=21)
    sd = _calibrated_tracknet(frames)
    net = tr.TrackNetRef(sd)
    x_ref, y_ref, v_ref, heat_ref = br.track(frames, net.forward, batch=4)
    m = E.Model(gpu_engine, G.build_tracknet(sd, dtype=E.graph_dtype()))
    m.set_max_batch(feed)
    sess = E.BallSession(m, 360, 640)
    sess.set_background(np.median(np.array([f[..., ::-1] for f in frames]), 0).astype("uint8"))
    masks, heats, rects = [], [], []
    for i in range(0, T, feed):
        mk, ht, rc = sess.feed(frames[i:i + feed], want_heat=True, want_rects=True)
        masks.append(mk); heats.append(ht); rects.append(rc)
    mk, ht, rc = sess.feed(None, flush=True, want_heat=True, want_rects=True)
    masks.append(mk); heats.append(ht); rects.append(rc)
    masks, heat, rects = np.concatenate(masks), np.concatenate(heats), np.concatenate(rects)
    # device predict_location == the oracle's, on the masks the device actually produced
    for i in range(T):
        assert tuple(rects[i]) == tuple(br.predict_location(masks[i])), i
    assert heat.shape == heat_ref.shape == (T, 288, 512)
    err = np.abs(heat - heat_ref).max()
    assert err < 2e-5, f"ensembled heat map max abs err {err:.2e}"
    want_mask = (heat_ref > 0.5)
    near = np.abs(heat_ref - 0.5) < 1e-4                       # threshold-adjacent pixels may flip
    assert np.array_equal((masks > 0)[~near], want_mask[~near])
    assert want_mask.any(), "calibration produced empty masks"
    sess.close()
    m.close()


def test_ball_locate_kernel_synthetic_masks(gpu_engine):
    """Connected-component pick on crafted masks: empty, single blob, diagonal (8-connectivity) links, equal-area
    ties (the component discovered last in raster order wins), a snake, blobs touching the borders, random
    speckle — against the oracle's predict_location."""
    m = E.Model(gpu_engine, G.build_tracknet(tr.synth_tracknet_state_dict(1), dtype=E.graph_dtype()))
    m.set_max_batch(8)
    sess = E.BallSession(m, 360, 640)
    rng = np.random.default_rng(0)
    masks = np.zeros((15, 288, 512), np.uint8)
    masks[1, 100:110, 200:215] = 255
    masks[2, 10:14, 20:29] = 255; masks[2, 100:103, 300:303] = 255; masks[2, 103, 303] = 255      # diagonal link
    masks[3, 5:9, 5:9] = 255; masks[3, 200:204, 400:404] = 255                                     # equal areas: last wins
    masks[4, 50:54, 60:64] = 255; masks[4, 50:54, 300:304] = 255; masks[4, 250:252, 10:18] = 255   # 16,16,16
    for i in range(60):                                                                             # snake
        masks[5, 20 + i, 30 + (i % 7)] = 255
        masks[5, 20 + i, 30 + ((i + 1) % 7)] = 255
    masks[6, 0:3, 0:5] = 255; masks[6, 285:288, 507:512] = 255; masks[6, 0:2, 509:512] = 255       # borders
    masks[7] = (rng.uniform(size=(288, 512)) > 0.97) * 255                                          # speckle (~4400 px)
    masks[8, 140:150, :] = 255                                                                      # full-width bar
    masks[9] = 255                                                                                  # overflow -> w = -1
    masks[10, 287, 511] = 255
    yy, xx = np.mgrid[0:288, 0:512]
    masks[11] = (((yy - 144) ** 2 + (xx - 256) ** 2) < 40 ** 2) * 255                              # disc (~5000 px)
    masks[12] = ((yy + xx) % 37 == 0) * 255                                                         # diagonal stripes
    masks[13, ::2, 100] = 255                                                                       # disconnected column
    masks[14, 30:60, 40:45] = 255; masks[14, 30:35, 40:90] = 255                                   # L shape
    rects = np.concatenate([sess.locate(masks[:8]), sess.locate(masks[8:])])
    for i in range(len(masks)):
        if i == 9:
            assert rects[i][2] == -1
            continue
        assert tuple(rects[i]) == tuple(br.predict_location(masks[i])), (i, rects[i], br.predict_location(masks[i]))
    sess.close(); m.close()


def test_ball_locate_kernel_hand_derived_ties(gpu_engine):
    """ball_locate_kernel on the equal-area cases whose winners are derived on paper from the chosen contour order
    (tests/known_answers.py:BALL_TIE_CASES) — literal expectations, not "equal to the oracle"."""
    from tests.known_answers import BALL_TIE_CASES
    m = E.Model(gpu_engine, G.build_tracknet(tr.synth_tracknet_state_dict(1), dtype=E.graph_dtype()))
    m.set_max_batch(8)
    sess = E.BallSession(m, 360, 640)
    masks = np.stack([c[1] for c in BALL_TIE_CASES])
    rects = sess.locate(masks)
    for (why, _, want), got in zip(BALL_TIE_CASES, rects):
        assert tuple(int(v) for v in got) == want, (why, got)
    sess.close(); m.close()


@pytest.mark.parametrize("n", [1, 2, 7, 16])
def test_device_median_matches_numpy(gpu_engine, n):
    m = E.Model(gpu_engine, G.build_tracknet(tr.synth_tracknet_state_dict(1), dtype=E.graph_dtype()))
    m.set_max_batch(4)
    sess = E.BallSession(m, 90, 160)
    rng = np.random.default_rng(n)
    frames = rng.integers(0, 256, (n, 90, 160, 3), dtype=np.uint8)
    frames[:, :10] = rng.integers(100, 104, (n, 10, 160, 3), dtype=np.uint8)      # many ties around the middle
    got = sess.background_from_frames(frames, want_median=True)
    want = np.median(np.array([f[..., ::-1] for f in frames]), 0).astype("uint8")
    assert np.array_equal(got, want)
    sess.close(); m.close()


def test_ball_tracker_plugin(gpu_engine, tmp_path):
    T = 20
    frames = _clip(T, 360, 640, seed=33)
    sd = _calibrated_tracknet(frames)
    ck = tmp_path / "TrackNet_synth.pt"
    checkpoint.save_checkpoint(ck, sd, "tracknet", param_dict={"seq_len": 8, "bg_mode": "concat"})
    x_ref, y_ref, v_ref, heat_ref = br.track(frames, tr.TrackNetRef(sd).forward, batch=4)
    t = BallTracker(str(ck), None, batch_size=6, median_max_sample_num=T)
    t.video_info_post_init(video.VideoInfo(640, 360, 30, T))
    balls = t.predict_and_update(iter(frames), total_frames=T)
    assert len(balls) == T
    near = [bool((np.abs(h - 0.5) < 1e-4).any()) for h in heat_ref]
    for i, b in enumerate(balls):
        assert b.frame == i
        if not near[i]:
            assert (b.xy[0], b.xy[1], b.visibility) == (x_ref[i], y_ref[i], v_ref[i]), i
    assert sum(v_ref) > 0
    t.to("cpu")
    # with an InpaintNet checkpoint the trajectory goes through the host repair stage (ball_tracker.py:525-673)
    from padel_analytics_amd import inpaint as ip
    sdi = tr.synth_inpaintnet_state_dict(8)
    ick = tmp_path / "InpaintNet_synth.pt"
    checkpoint.save_checkpoint(ick, sdi, "inpaintnet", param_dict={"seq_len": 16})
    t2 = BallTracker(str(ck), str(ick), batch_size=6, median_max_sample_num=T)
    t2.video_info_post_init(video.VideoInfo(640, 360, 30, T))
    balls2 = t2.predict_and_update(iter(frames), total_frames=T)
    # ... and must equal the ORACLE's restatement of that stage (oracle/ball_ref.py:inpaint_stage_ref, the reference's
    # streaming loop with the torch InpaintNet of tracknet_ref.py, itself pinned to the reference's models.py goldens)
    # applied to the TrackNet-stage trajectory — not the product's own function
    base = [(b.xy[0], b.xy[1], b.visibility) for b in balls]
    want, raw = br.inpaint_stage_ref([b[0] for b in base], [b[1] for b in base], [b[2] for b in base], 640, 360,
                                     tr.InpaintNetRef(sdi).forward, 16, batch_size=6)
    got = [(b.xy[0], b.xy[1], b.visibility) for b in balls2]
    assert len(got) == T == len(want)
    exact = 0
    for g in range(T):
        fx, fy = raw[g]
        if got[g] == want[g]:
            exact += 1
            continue
        # numpy vs torch InpaintNet differ by ~1e-6: only a coordinate sitting on an int() truncation boundary may move
        assert min(abs(fx - round(fx)), abs(fy - round(fy))) < 1e-2, (g, got[g], want[g], raw[g])
        assert abs(got[g][0] - want[g][0]) <= 1 and abs(got[g][1] - want[g][1]) <= 1 and got[g][2] == want[g][2], (g, got[g], want[g])
    assert exact >= T - 2
    t2.to("cpu")


@pytest.mark.parametrize("mode", ["h2", "bx3"])
def test_inpaintnet_on_device_matches_the_numpy_twin_and_the_oracle(gpu_engine, mode):
    """K12 on the device (round 4): graph.build_inpaintnet — Conv1d(k=3) + LeakyReLU layers as 3x3 convolutions over
    one-row images — against the numpy twin (padel_analytics_amd/inpaint.py:InpaintNetHost) and the torch oracle
    (oracle/tracknet_ref.py:InpaintNetRef, pinned to golden outputs of the reference's models.py): sigmoid outputs within
    2e-6; the repaired trajectory of a 200-frame clip (ints, in pixels) identical."""
    import torch
    from padel_analytics_amd import inpaint as ip
    sd = tr.synth_inpaintnet_state_dict(4)
    rng = np.random.default_rng(3)
    S, L = 300, 16
    coor = rng.uniform(0, 1, (S, L, 2)).astype(np.float32)
    mask = (rng.uniform(size=(S, L, 1)) > 0.7).astype(np.float32)
    coor[mask[..., 0] > 0] = 0.0
    host = ip.InpaintNetHost(sd)
    dev = ip.InpaintNetDevice(sd, gpu_engine, max_windows=128)           # 300 windows in three passes
    dev.mode = mode
    got = dev.forward(coor, mask)
    want = host.forward(coor, mask)
    ref = tr.InpaintNetRef(sd).forward(torch.from_numpy(coor), torch.from_numpy(mask)).numpy()
    assert got.shape == want.shape == (S, L, 2)
    assert np.abs(want - ref).max() < 2e-6
    assert np.abs(got - ref).max() < 2e-6, np.abs(got - ref).max()
    # through the whole repair: same integers as the host network
    T = 200
    xs = (640 + 300 * np.sin(np.arange(T) / 9.0)).astype(int).tolist()
    ys = (360 + 200 * np.cos(np.arange(T) / 7.0)).astype(int).tolist()
    vs = [1] * T
    for a, b in ((20, 26), (70, 73), (120, 131)):
        for i in range(a, b):
            xs[i], ys[i], vs[i] = 0, 0, 0
    r_dev = ip.inpaint_trajectory(xs, ys, vs, 1280, 720, dev, L)
    r_host = ip.inpaint_trajectory(xs, ys, vs, 1280, 720, host, L)
    assert r_dev == r_host
    assert any(r_host[i][2] == 1 for i in range(20, 26)), "the masked gap is repaired"
    dev.close()
