"""Pins the preprocessing restatements: Pillow bicubic bit-exact against Pillow itself; LetterBox geometry
known answers; cv2 bilinear special cases that are provable without OpenCV."""
import numpy as np
import pytest
from PIL import Image

from oracle import preprocess_ref as pre, yolov8_ref as ref


@pytest.mark.parametrize("hw,out", [((72, 128), (128, 128)), ((108, 192), (128, 128)), ((720, 1280), (512, 288)),
                                    ((90, 160), (64, 36)), ((37, 53), (64, 64)), ((64, 64), (64, 32))])
def test_pil_bicubic_bit_exact(hw, out):
    rng = np.random.default_rng(hw[0] * 7 + out[0])
    img = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize(out))          # PIL default = BICUBIC
    got = pre.pil_resize_bicubic_u8(img, out[0], out[1])
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_pil_bicubic_reference_sizes_rows():
    # full 720x1280 -> 1280x1280 (pose, config.py:30): width unchanged -> only the vertical pass runs
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((1280, 1280)))
    got = pre.pil_resize_bicubic_u8(img[:, :64], 64, 1280)      # columns are independent in the vertical pass
    assert np.array_equal(got, want[:, :64])


def test_letterbox_geometry_known_answers():
    assert ref.letterbox_geometry(720, 1280) == (640, 360, 12, 12, 0, 0)
    assert ref.letterbox_geometry(1080, 1920) == (640, 360, 12, 12, 0, 0)
    assert ref.letterbox_geometry(640, 640) == (640, 640, 0, 0, 0, 0)
    assert ref.letterbox_geometry(720, 1280, auto=False) == (640, 360, 140, 140, 0, 0)
    out = ref.letterbox_u8(np.zeros((720, 1280, 3), np.uint8))
    assert out.shape == (384, 640, 3) and (out[:12] == 114).all() and (out[-12:] == 114).all() and (out[12:-12] == 0).all()


def test_cv2_linear_special_cases():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (36, 48, 3), dtype=np.uint8)
    half = ref.cv2_resize_linear_u8(img, 24, 18)                 # area-fast path
    a = img.astype(np.int32)
    assert np.array_equal(half, ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    third = ref.cv2_resize_linear_u8(img, 16, 12)                # exact 1/3: source coord 3i+1, weight 1
    assert np.array_equal(third, img[1::3, 1::3])
    assert np.array_equal(ref.cv2_resize_linear_u8(img, 48, 36), img)


def test_scale_boxes_known_answers():
    import torch
    b = torch.tensor([[0.0, 12.0, 640.0, 372.0]])
    assert ref.scale_boxes((384, 640), b, (720, 1280)).tolist() == [[0.0, 0.0, 1280.0, 720.0]]
    k = torch.tensor([[[320.0, 192.0, 1.0]]])
    assert ref.scale_coords((384, 640), k, (720, 1280))[0, 0, :2].tolist() == [640.0, 360.0]
