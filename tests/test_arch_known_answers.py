"""Known answers that pin the restated YOLOv8 graph without the upstream package:
Ultralytics' published parameter counts and GFLOPs at 640x640."""
import pytest

from padel_analytics_amd import yolo_arch as ya


@pytest.mark.parametrize("scale,params,gflops", [("n", 3157200, 8.74), ("s", 11166560, 28.60), ("m", 25902640, 78.94),
                                                 ("l", 43691520, 165.15), ("x", 68229648, 257.80)])
def test_detect_params_and_flops(scale, params, gflops):
    assert ya.count_parameters(ya.yolov8_state_spec(scale, 80)) == params
    assert abs(ya.conv_flops(ya.conv_inventory(scale, 80, None, 640, 640)) / 1e9 - gflops) < 0.01


@pytest.mark.parametrize("scale,params,gflops", [("n", 3295470, 9.18), ("m", 26464462, 81.02)])
def test_pose_params_and_flops(scale, params, gflops):
    assert ya.count_parameters(ya.yolov8_state_spec(scale, 1, (17, 3))) == params
    assert abs(ya.conv_flops(ya.conv_inventory(scale, 1, (17, 3), 640, 640)) / 1e9 - gflops) < 0.01


def test_baseline_table_flops():
    # BASELINE.md §2
    f = lambda *a: round(ya.conv_flops(ya.conv_inventory(*a)) / 1e9, 2)
    assert f("n", 80, None, 384, 640) == 5.25
    assert f("n", 1, None, 384, 640) == 4.85
    assert f("m", 80, None, 384, 640) == 47.36
    assert f("n", 1, (13, 3), 1280, 1280) == 35.36
    assert f("m", 1, (13, 3), 1280, 1280) == 323.41


def test_infer_arch_roundtrip():
    for scale, nc, kpt in (("n", 80, None), ("m", 1, (13, 3)), ("s", 3, (13, 2))):
        sd = {k: __import__("numpy").zeros(v) for k, v in ya.yolov8_state_spec(scale, nc, kpt).items()}
        info = ya.infer_arch_from_state_dict(sd)
        assert info == {"scale": scale, "nc": nc, "nk": 0 if kpt is None else kpt[0] * kpt[1]}
