"""The two BASELINE.json configurations that no other GPU test runs as such (VERDICT r2 "configs_untested"):

* configs[3] — "1280x720 batch=256 frame-sharded across 8 x MI355X, all three trackers, RCCL weight bcast": ONE rank's
  share of it on the one GPU a test box has: the 32-frame shard ``dist.shard_range(256, r, 8)``, an RCCL communicator
  (``comm_init``, one rank), and for each of the three graphs a model created from a NULL blob whose weights arrive
  ONLY through ``ncclBroadcast`` (``pa_engine_bcast_weights_from``) — it must give BITWISE the detections of the model
  that loaded the blob.  (Seven of the eight shards get their weights exactly this way.)
* configs[4] — "1920x1080 fp16 batch=512, all trackers + batched NMS, 8 x MI355X": one GPU's 64 frames of it through
  ``TrackingRunner`` with ``half=True`` trackers (PolygonZone, ByteTrack, result objects, JSON caches), plus the parity
  statement of that precision IN PIXELS on low-noise heads (the fp16 path makes no 1e-3 px claim: activations carry 11
  bits; the number is asserted loosely and reported)."""
import json
import os

import numpy as np
import pytest

import bench
from oracle import yolov8_ref as ref
from padel_analytics_amd import checkpoint, detections as D, dist, engine as E, graph as G, video
from tests import synth
from padel_analytics_amd.trackers import BallDetectTracker, PlayerKeypointsTracker, PlayerTracker, TrackingRunner
from tests import parity

pytestmark = pytest.mark.gpu


def _infer(m, cfg, frames, n, h, w):
    return m.yolo_infer(frames, n, h, w, imgsz=cfg["imgsz"], conf=cfg["conf"], iou=0.7, classes=cfg["classes"],
                        pre_mode=E.PRE_PIL_STRETCH if cfg["pre"] == "pil" else E.PRE_LETTERBOX, channel_reverse=cfg["rev"])


def test_config3_shard_with_rccl_delivered_weights(gpu_engine):
    eng = gpu_engine
    if getattr(eng, "nranks", None) is None:
        eng.comm_init(E.comm_unique_id(), 1, 0)                  # a communicator of one rank: RCCL on one GPU
    lo, hi = dist.shard_range(256, 5, 8)
    assert (lo, hi) == (160, 192)
    frames = synth.synthetic_frames(hi - lo, 720, 1280, seed=1000 + 5)         # this rank's shard
    total = 0
    for name in ("players", "ball", "pose"):
        cfg = bench.TRACKERS[name]
        sd = bench.make_state_dict(name, cfg, frames)
        g = G.build_yolov8(sd, cfg["nc"], cfg["kpt"], dtype=E.graph_dtype())
        loaded = E.Model(eng, g)
        loaded.set_max_batch(32)
        want = _infer(loaded, cfg, frames, 32, 720, 1280)
        empty = E.Model(eng, g, empty=True)                      # NULL blob: zero weights in HBM
        empty.set_max_batch(32)
        if name == "players":                                    # it really is empty: zero logits -> score 0.5, not > conf 0.5
            zero = _infer(empty, cfg, frames[:2], 2, 720, 1280)
            assert int(zero[2].sum()) == 0 and int(want[2][:2].sum()) > 0
        eng.bcast_weights_from(loaded, empty, root=0)            # ncclBroadcast: loaded blob -> empty model's blob
        got = _infer(empty, cfg, frames, 32, 720, 1280)
        for a, b, what in zip(want, got, ("boxes", "kpts", "counts")):
            if a is not None:
                assert np.array_equal(a, b), f"{name}: {what} differ between the loaded and the RCCL-delivered weights"
        assert not loaded.take_overflow() and not empty.take_overflow()
        total += int(want[2].sum())
        loaded.close(); empty.close()
    assert total > 0


def _trackers_half(tmp, frames, B, H, W, eng, scales=None):
    trackers = {}
    for name in ("players", "ball", "pose"):
        cfg = dict(bench.TRACKERS[name])
        sd = bench.make_state_dict(name, cfg, frames)
        path = tmp / f"{name}.pt"
        checkpoint.save_checkpoint(path, sd, "pose" if cfg["kpt"] else "detect", cfg["nc"], cfg["kpt"], cfg["scale"],
                                   {0: "person" if name != "ball" else "ball"})
        if name == "players":
            sx, sy = W / 1280.0, H / 720.0
            zone = D.PolygonZone(np.array([[int(x * sx), int(y * sy)] for x, y in bench.ZONE_720P]), frame_resolution_wh=(W, H))
            t = PlayerTracker(str(path), zone, batch_size=B, half=True, save_path=tmp / "players.json")
        elif name == "pose":
            t = PlayerKeypointsTracker(str(path), cfg["imgsz"], batch_size=B, half=True, save_path=tmp / "pose.json")
        else:
            t = BallDetectTracker(str(path), batch_size=B, conf=cfg["conf"], half=True, save_path=tmp / "ball.json")
        t.model.set_max_batch(B)
        t.model.attach(eng)
        trackers[name] = t
    return trackers


def test_config4_1080p_fp16_all_trackers_through_the_runner(gpu_engine, tmp_path):
    H, W, B = 1080, 1920, 64
    frames = synth.synthetic_frames(B, H, W, seed=1000)
    trackers = _trackers_half(tmp_path, frames, B, H, W, gpu_engine)
    clip = video.DeviceClip(gpu_engine, frames)
    runner = TrackingRunner(list(trackers.values()), clip, tmp_path / "out.mp4")
    runner.run()
    assert all(len(t) == B for t in trackers.values())
    for nm in ("players.json", "pose.json", "ball.json"):
        assert len(json.loads((tmp_path / nm).read_text())) == B
    n_players = sum(len(p) for p in trackers["players"].results.predictions)
    n_pose = sum(len(p) for p in trackers["pose"].results.predictions)
    assert n_players > 0 and n_pose > 0
    assert all(t.model.graph.dtype == G.DTYPE_F16 for t in trackers.values())
    # ---- parity statement of this precision, in pixels, on low-noise heads, vs the fp32 CPU oracle: three independently
    # seeded clips + checkpoints per tracker (2 frames each).  fp16 activations carry 11 bits: the statement is in pixels,
    # not the fp32 path's 1e-3 px, and it is asserted at measured + margin (round 3 asserted "< 8 px")
    report = {"frames": B, "tracked_players": n_players, "pose_detections": n_pose, "low_noise_heads": {}}
    # per tracker: bound on the geometric mean of the RMS error over the seeds, bound on the worst L-inf of any seed
    # measured (gpurun r4d, three seeds): geomean RMS 0.114 / 0.054 / 0.028 px, worst L-inf 1.57 / 0.42 / 1.06 px (round 3 saw up to
    # 4 px on one draw of the players checkpoint: the maximum of a heavy-tailed sample gets the wider margin)
    BOUNDS = {"players": (0.25, 4.0), "ball": (0.12, 1.0), "pose": (0.07, 2.5)}
    for name in ("players", "ball", "pose"):
        cfg = bench.TRACKERS[name]
        f = 0.004 if cfg["imgsz"] > 640 else 0.02
        per_seed = []
        for sidx in range(3):
            sample = synth.synthetic_frames(2, H, W, seed=1000 + 17 * sidx)
            sd = dict(bench.make_state_dict(name, cfg, sample, seed_offset=101 * sidx))
            for branch in ("cv2", "cv4"):
                for l in range(3):
                    for nm in ("weight", "bias"):
                        k = f"model.22.{branch}.{l}.2.{nm}"
                        if k in sd:
                            sd[k] = (sd[k] * np.float32(f)).astype(np.float16).astype(np.float32)
            srcs = bench.source_for_oracle(cfg, sample)
            r32 = ref.predict(ref.YoloV8Ref(sd, cfg["nc"], cfg["kpt"]), srcs, cfg["conf"], 0.7, cfg["imgsz"], cfg["classes"])
            m = E.Model(gpu_engine, G.build_yolov8(sd, cfg["nc"], cfg["kpt"], dtype="f16"))
            m.set_max_batch(2)
            boxes, kpts, counts = _infer(m, cfg, np.ascontiguousarray(sample), 2, H, W)
            m.close()
            tot = mt = 0
            worst, sq, cnt = 0.0, 0.0, 0
            for i, r in enumerate(r32):
                gb = boxes[i, :counts[i]]
                pairs, ru, gu = parity.match(r["boxes"], gb, tol_match=6.0)
                tot += len(r["boxes"]); mt += len(pairs)
                for i_r, i_g in pairs:
                    d = np.abs(gb[i_g, :4] - r["boxes"][i_r, :4]).astype(np.float64)
                    worst = max(worst, float(d.max())); sq += float((d ** 2).sum()); cnt += 4
                    if kpts is not None and r["kpts"] is not None:
                        gk = kpts[i, i_g].reshape(*cfg["kpt"])
                        dk = np.abs(gk[..., :2] - r["kpts"][i_r][..., :2]).astype(np.float64)
                        worst = max(worst, float(dk.max())); sq += float((dk ** 2).sum()); cnt += dk.size
            rms = (sq / max(cnt, 1)) ** 0.5
            per_seed.append({"detections": tot, "matched": mt, "linf_px_vs_fp32_oracle": round(worst, 4), "rms_px_vs_fp32_oracle": round(rms, 4)})
        gm_rms = float(np.exp(np.mean(np.log([max(p_["rms_px_vs_fp32_oracle"], 1e-6) for p_ in per_seed]))))
        worst_linf = max(p_["linf_px_vs_fp32_oracle"] for p_ in per_seed)
        report["low_noise_heads"][name] = {"per_seed": per_seed, "geomean_rms_px": round(gm_rms, 4), "worst_linf_px": worst_linf,
                                           "bounds": {"geomean_rms_px": BOUNDS[name][0], "worst_linf_px": BOUNDS[name][1]}}
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "config4_report.json"), "w"), indent=1)
    for name, (b_rms, b_linf) in BOUNDS.items():
        e = report["low_noise_heads"][name]
        # near `conf` fp16 scores decide differently from fp32 ones: the sets overlap, they are not equal (pose: 86-94 % matched)
        assert all(p_["detections"] > 0 and p_["matched"] >= 0.8 * p_["detections"] for p_ in e["per_seed"]), (name, e)
        assert e["geomean_rms_px"] <= b_rms and e["worst_linf_px"] <= b_linf, (name, e)
    print("configs[4] (one GPU's 64 frames, fp16, 1080p):", report)
    clip.free()
    for t in trackers.values():
        t.model.close()


def test_config0_32_frames_640_yolov8n_through_the_player_tracker(gpu_engine, tmp_path):
    """BASELINE configs[0] AS STATED: a 32-frame 640 x 640 clip through `PlayerTracker` (yolov8n, conf .5,
    classes=[0], batch 8) driven by `TrackingRunner` from host frames — the reference's plumbing case.  Against the CPU
    oracle on all 32 frames: identical detection sets and classes before the zone, coordinates as close to the exact (fp64)
    evaluation as the fp32 oracle itself; then the plugin's own output: the zone keeps exactly the oracle's detections
    whose bottom-centre anchor lies inside, ByteTrack ids equal those of the python twin fed with the ORACLE's boxes."""
    import torch
    from padel_analytics_amd import bytetrack as BT
    H = W = 640
    N, B = 32, 8
    frames = synth.synthetic_frames(N, H, W, seed=77)
    srcs = [f[..., ::-1] for f in frames]
    from tests.helpers import calibrated_state_dict
    sd = calibrated_state_dict("n", 80, None, srcs, 640, 0.5, seed=19, frac=0.004)
    checkpoint.save_checkpoint(tmp_path / "yolov8n.pt", sd, "detect", 80, None, "n", {0: "person"})
    zone = D.PolygonZone(np.array([[40, 60], [600, 60], [620, 630], [20, 630]]), frame_resolution_wh=(W, H))
    t = PlayerTracker(str(tmp_path / "yolov8n.pt"), zone, batch_size=B, save_path=tmp_path / "players.json")
    t.model.attach(gpu_engine)
    # --- raw detections of the device stage, batch by batch as the runner feeds them
    boxes = np.zeros((N, 300, 6), np.float32)
    counts = np.zeros(N, np.int32)
    for lo in range(0, N, B):
        b, c = t.infer_sample(list(frames[lo:lo + B]))
        boxes[lo:lo + B], counts[lo:lo + B] = b, c
    r32 = ref.predict(ref.YoloV8Ref(sd, 80, None), srcs, 0.5, 0.7, 640, classes=[0])
    r64 = ref.predict(ref.YoloV8Ref(sd, 80, None, dtype=torch.float64), srcs, 0.5, 0.7, 640, classes=[0])
    b64 = np.zeros((N, 300, 6), np.float32)
    c64 = np.zeros(N, np.int32)
    for i, r in enumerate(r64):
        c64[i] = len(r["boxes"])
        b64[i, :c64[i]] = r["boxes"]
    floor = parity.compare_batch(r32, b64, None, c64, 0.5, 0.7)
    g32 = parity.compare_batch(r32, boxes, None, counts, 0.5, 0.7)          # raises on class / set mismatch
    g64 = parity.compare_batch(r64, boxes, None, counts, 0.5, 0.7)
    assert g32["n"] >= N, "the calibration should give at least a detection per frame"
    assert g64["worst_px"] <= max(1e-3, 4 * floor["worst_px"]), (g64["worst_px"], floor["worst_px"])
    assert g64["rms_px"] <= max(2e-4, 1.5 * floor["rms_px"]), (g64["rms_px"], floor["rms_px"])
    # --- the plugin's output through the runner (host frames, batch 8)
    runner = TrackingRunner([t], video.ArrayClip(frames), tmp_path / "out.mp4")
    runner.run()
    assert len(t) == N and len(json.loads((tmp_path / "players.json").read_text())) == N
    twin = BT.ByteTrack(frame_rate=30)
    n_ids = 0
    for i, r in enumerate(r32):
        ob = r["boxes"]
        ob = ob[zone.trigger_boxes(ob[:, :4])] if len(ob) else ob
        out = twin.update_with_detections(D.Detections(xyxy=ob[:, :4].copy(), confidence=ob[:, 4].copy(),
                                                       class_id=ob[:, 5].astype(int)))
        got = t.results.predictions[i]
        assert len(got) == len(out), (i, len(got), len(out))
        for p_, wb, wid in zip(got.players, out.xyxy, out.tracker_id):
            assert np.abs(np.asarray(p_.xyxy, np.float64) - wb).max() <= max(1e-3, 5 * floor["worst_px"]), (i, p_.xyxy, wb)
            assert (p_.id or 0) == int(wid), (i, p_.id, wid)
            n_ids += 1
    assert n_ids > 0
    print(f"configs[0]: {g32['n']} detections on 32 frames, L-inf vs fp64 {g64['worst_px']:.2e} px (oracle floor {floor['worst_px']:.2e}), "
          f"{n_ids} tracked boxes with ids equal to the python twin's")
    t.model.close()
