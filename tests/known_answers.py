"""Hand-derived known answers for the [upstream] pieces that cannot be pinned against ultralytics / torchvision here
(reference call sites players_tracker.py:351-359, players_keypoints_tracker.py:285-299): the
Detect / Pose inference branch (DFL expectation, dist2bbox, keypoint decode), ops.non_max_suppression (strict
``score > conf``, strict ``IoU > thr``, stable order on equal scores, the per-class 7680 offset, ``classes=``) and
scale_boxes / scale_coords (rounded vs unrounded letterbox pads, clamping).  Every expected number below is worked out
in the comments from the published formulas — none is produced by running code.  The same cases go through the CPU
oracle (tests/test_known_answers_oracle.py) and through the HIP decode / NMS kernels (tests/test_gpu_known_answers.py).

Head-map layout (pa_yolo_head_shape): per level (n, H_l, W_l, c): channels [0, 64) = DFL logits, side-major
(left, top, right, bottom) x 16 bins; [64, 64 + nc) class logits; then nk keypoint values.  Source frames are
720 x 1280 -> network 384 x 640 (gain 1/2, letterbox pads: x 0, y 12), levels 48x80 / 24x40 / 12x20 at strides 8 / 16 / 32."""
import numpy as np

H0, W0, IMGSZ = 720, 1280, 640
LEVELS = [(48, 80, 8), (24, 40, 16), (12, 20, 32)]
OFF = -20.0            # class logit of every untouched anchor: sigmoid = 2e-9, far below any threshold
HOT = 50.0             # DFL logit of the chosen bin: softmax = 1 - 15 e^-50, expectation == the bin index in fp32


def blank_heads(n, c):
    heads = [np.zeros((n, h, w, c), np.float32) for (h, w, _) in LEVELS]
    return heads


def put(heads, img, level, y, x, ltrb=None, cls_logits=None, nc=1, kpts=None):
    """One anchor: DFL one-hot at integer distances ltrb (None: all-zero logits = uniform bins, expectation 7.5 each)."""
    t = heads[level][img, y, x]
    if ltrb is not None:
        for side, d in enumerate(ltrb):
            t[side * 16 + d] = HOT
    for k, v in enumerate(cls_logits):
        t[64 + k] = v
    if kpts is not None:
        t[64 + nc:64 + nc + len(kpts)] = kpts


def detect_cases():
    """nc = 2.  Returns (heads, expected) with expected[img] = list of rows [x1, y1, x2, y2, score, cls] in output
    order, for conf = 0.5, iou = 0.7, classes = None."""
    nc, c = 2, 68
    heads = blank_heads(4, c)
    for hd in heads:
        hd[..., 64:64 + nc] = OFF
    sig = lambda v: float(1.0 / (1.0 + np.exp(-np.float64(v))))
    exp = {}
    # ---- image 0: dist2bbox + scale_boxes.
    # level 0 (stride 8), cell (y 10, x 20): anchor centre (20.5, 10.5); l, t, r, b = 2, 3, 4, 5 cells
    #   x1 = 20.5 - 2 = 18.5, y1 = 10.5 - 3 = 7.5, x2 = 24.5, y2 = 15.5  -> x 8: (148, 60, 196, 124) network px
    #   scale_boxes: minus pad (0, 12) -> (148, 48, 196, 112); / gain 0.5 -> (296, 96, 392, 224)
    put(heads, 0, 0, 10, 20, (2, 3, 4, 5), (2.0, OFF), nc)
    # level 1 (stride 16), cell (5, 5), all DFL logits zero: uniform over bins 0..15 -> every distance 7.5
    #   (5.5 -+ 7.5) * 16 = (-32, -32, 208, 208); minus pad -> (-32, -44, 208, 196); / 0.5 -> (-64, -88, 416, 392); clamp to
    #   [0, 1280] x [0, 720] -> (0, 0, 416, 392).  class 1 wins (logit 1.0 vs 0.5)
    put(heads, 0, 1, 5, 5, None, (0.5, 1.0), nc)
    # an anchor whose best logit is exactly 0: sigmoid = 0.5, and the filter is STRICT (score > conf) -> no detection
    put(heads, 0, 0, 30, 60, (1, 1, 1, 1), (0.0, OFF), nc)
    # bottom-right corner cell of level 2 (stride 32), cell (11, 19): centre (19.5, 11.5), l, t, r, b = 1, 1, 15, 15
    #   (18.5, 10.5, 34.5, 26.5) * 32 = (592, 336, 1104, 848); minus pad -> (592, 324, 1104, 836); / 0.5 ->
    #   (1184, 648, 2208, 1672) -> clamp (1184, 648, 1280, 720)
    put(heads, 0, 2, 11, 19, (1, 1, 15, 15), (3.0, OFF), nc)
    exp[0] = [[1184, 648, 1280, 720, sig(3.0), 0], [296, 96, 392, 224, sig(2.0), 0], [0, 0, 416, 392, sig(1.0), 1]]
    # ---- image 1: NMS.  All on level 2 (stride 32), class 0.
    #   A: cell (5, 5), l t r b = 5 5 5 5: (0.5, 0.5, 10.5, 10.5) * 32 = (16, 16, 336, 336), area 102400, score sig(2.2)
    #   B: cell (4, 5) [y 4, x 5], l t r b = 5 4 5 3: x (0.5, 10.5), y (0.5, 7.5) -> (16, 16, 336, 240), area 71680 = 0.7 * area A,
    #      B inside A: IoU(A, B) = 71680 / 102400 = float32(0.7) — NOT greater than the threshold float32(0.7): B is KEPT
    #   C: cell (5, 4) [y 5, x 4], l t r b = 4 5 6 3: x (0.5, 10.5), y (0.5, 8.5) -> (16, 16, 336, 272), area 81920:
    #      IoU(A, C) = 0.8 > 0.7 -> suppressed by A (higher score)
    #   D: far away, same score as B (equal logits): stable order -> B (lower anchor index) before D
    put(heads, 1, 2, 5, 5, (5, 5, 5, 5), (2.2, OFF), nc)
    put(heads, 1, 2, 4, 5, (5, 4, 5, 3), (1.5, OFF), nc)
    put(heads, 1, 2, 5, 4, (4, 5, 6, 3), (2.0, OFF), nc)
    put(heads, 1, 2, 9, 15, (1, 1, 1, 1), (1.5, OFF), nc)
    #   scale: A (16, 16, 336, 336) -> minus (0, 12) -> (16, 4, 336, 324) -> x2: (32, 8, 672, 648)
    #          B (16, 16, 336, 240) -> (16, 4, 336, 228) -> (32, 8, 672, 456)
    #          D cell (9, 15): centre (15.5, 9.5) -+ 1 -> (14.5, 8.5, 16.5, 10.5) * 32 = (464, 272, 528, 336) -> (464, 260, 528, 324) -> (928, 520, 1056, 648)
    exp[1] = [[32, 8, 672, 648, sig(2.2), 0], [32, 8, 672, 456, sig(1.5), 0], [928, 520, 1056, 648, sig(1.5), 0]]
    # ---- image 2: the class offset.  Two anchors produce the SAME box (16, 16, 336, 336): cell (5, 5) with l r = 5 5 and
    #   cell (5, 4) with l r = 4 6; one is class 0, the other class 1 -> boxes are shifted by cls * 7680 before NMS:
    #   no overlap, both kept.  A third, class-0 copy of the box from cell (5, 6) [l r = 6 4] with a lower score IS suppressed (IoU 1).
    put(heads, 2, 2, 5, 5, (5, 5, 5, 5), (2.0, OFF), nc)
    put(heads, 2, 2, 5, 4, (4, 5, 6, 5), (OFF, 1.0), nc)
    put(heads, 2, 2, 5, 6, (6, 5, 4, 5), (0.8, OFF), nc)
    exp[2] = [[32, 8, 672, 648, sig(2.0), 0], [32, 8, 672, 648, sig(1.0), 1]]
    # ---- image 3: nothing above the threshold
    exp[3] = []
    return nc, heads, exp


def detect_cases_class_filter():
    """The same heads with classes=[1]: only class-1 rows survive (upstream filters BEFORE nms)."""
    nc, heads, exp = detect_cases()
    return nc, heads, {i: [r for r in rows if r[5] == 1] for i, rows in exp.items()}


def pose_cases():
    """nc = 1, 13 keypoints x 3.  One anchor; conf 0.25.
    level 0 (stride 8), cell (y 10, x 20): box l t r b = 2 3 4 5 -> (296, 96, 392, 224) as in detect image 0.
    keypoint k raw (vx, vy, vl): x = (vx * 2 + 20) * 8, y = (vy * 2 + 10) * 8 (anchor - 0.5 = the cell index), vis = sigmoid(vl);
    scale_coords uses the UNROUNDED pads (0, 12.0): ((x - 0) / 0.5, (y - 12) / 0.5), clamped to the frame.
      k0: (0.25, 0.75, 4.0)  -> x = 20.5 * 8 = 164 -> 328;  y = 11.5 * 8 = 92 -> (92 - 12) / 0.5 = 160
      k1: (-1.0, -0.5, -4.0) -> x = 18 * 8 = 144 -> 288;    y = 9 * 8 = 72 -> 120        (visibility 0.018: Keypoints.xy zeroes it)
      k2: (100, 100, 0.0)    -> clamp: x -> 1280, y -> 720   (visibility exactly 0.5: NOT < 0.5, kept)
      others: (0, 0, OFF)    -> x = 160 -> 320, y = 80 -> 136"""
    nc, nk, c = 1, 39, 104
    heads = blank_heads(1, c)
    for hd in heads:
        hd[..., 64] = OFF
    k = np.zeros((13, 3), np.float32)
    k[:, 2] = OFF
    k[0] = (0.25, 0.75, 4.0)
    k[1] = (-1.0, -0.5, -4.0)
    k[2] = (100.0, 100.0, 0.0)
    put(heads, 0, 0, 10, 20, (2, 3, 4, 5), (1.0,), nc, k.reshape(-1))
    sig = lambda v: float(1.0 / (1.0 + np.exp(-np.float64(v))))
    ek = np.zeros((13, 3), np.float64)
    ek[:, 0], ek[:, 1], ek[:, 2] = 320.0, 136.0, sig(OFF)
    ek[0] = (328.0, 160.0, sig(4.0))
    ek[1] = (288.0, 120.0, sig(-4.0))
    ek[2] = (1280.0, 720.0, 0.5)
    return nc, (13, 3), heads, [[296, 96, 392, 224, sig(1.0), 0]], ek


GEOMETRY = [
    # (h0, w0) -> LetterBox(640, auto, stride 32): resized (w, h), (top, bottom, left, right), network (h, w); scale_boxes gain, pads
    # 720p: r = min(640/720, 640/1280) = 0.5 -> 640 x 360; dh = 280 % 32 = 24 -> 12 + 12 (round(11.9), round(12.1))
    ((720, 1280), (640, 360), (12, 12, 0, 0), (384, 640), 0.5, (0, 12)),
    # 1080p: r = 1/3 -> 640 x 360, same pads; scale_boxes gain = min(384/1080, 640/1920) = 1/3, pad_y = round((384 - 360)/2 - 0.1) = 12
    ((1080, 1920), (640, 360), (12, 12, 0, 0), (384, 640), 1.0 / 3.0, (0, 12)),
    # 480 x 854: r = min(1.333, 0.74941) -> round(854 r) = 640, round(480 r) = round(359.72) = 360; same pads;
    # scale_boxes gain = min(384/480, 640/854) = 0.749414..., pad_y = round((384 - 359.719) / 2 - 0.1) = round(12.04) = 12
    ((480, 854), (640, 360), (12, 12, 0, 0), (384, 640), 640.0 / 854.0, (0, 12)),
    # square: nothing to do
    ((640, 640), (640, 640), (0, 0, 0, 0), (640, 640), 1.0, (0, 0)),
    # 600 x 800 (odd pad): r = 0.8 -> 640 x 480; dh = 160 % 32 = 0 -> no pad
    ((600, 800), (640, 480), (0, 0, 0, 0), (480, 640), 0.8, (0, 0)),
    # 500 x 1000: r = 0.64 -> 640 x 320; dh = 320 % 32 = 0
    ((500, 1000), (640, 320), (0, 0, 0, 0), (320, 640), 0.64, (0, 0)),
    # 700 x 1000: r = 0.64 -> 640 x 448; dh = 192 % 32 = 0
    ((700, 1000), (640, 448), (0, 0, 0, 0), (448, 640), 0.64, (0, 0)),
    # 725 x 1280: r = 0.5 -> 640 x round(362.5) = 362 (banker's rounding of Python's round); dh = 278 % 32 = 22 -> 11 + 11
    ((725, 1280), (640, 362), (11, 11, 0, 0), (384, 640), 0.5, (0, 11)),
]


# ---------------------------------------------------------------------------------------------------------------------
# Heat-map decode ties (reference predict.py:21-33): ``predict_location`` keeps the FIRST rectangle of maximal w * h
# (strict ``>``, bounding-box area, not pixel count) in the order ``cv2.findContours(mask, RETR_EXTERNAL, ...)`` returns the
# contours.  cv2 is not installable here; the CHOSEN rule (oracle/ball_ref.py, trackers/ball_tracker.py:predict_location, the
# device kernel csrc/tracknet_post.hip:ball_locate_kernel): contours come back in REVERSE raster-discovery order — a raster
# scan (row by row, left to right) discovers a component at its first foreground pixel, the component discovered LAST is
# returned first — so among equal-area rectangles the winner is the component whose first pixel comes LAST in raster order.
# Every expectation below is worked out from that sentence alone.  (mask size 288 x 512; rectangles as (x, y, w, h).)
def _paint(rects, extra=()):
    import numpy as np
    m = np.zeros((288, 512), np.uint8)
    for x, y, w, h in rects:
        m[y:y + h, x:x + w] = 255
    for x, y in extra:
        m[y, x] = 255
    return m


BALL_TIE_CASES = [
    # two 4 x 4 squares, first pixels at (row 5, col 5) and (row 200, col 400): the lower one is discovered last -> wins
    ("two equal squares", _paint([(5, 5, 4, 4), (400, 200, 4, 4)]), (400, 200, 4, 4)),
    # same row of first pixels (row 50): the scan meets col 60 before col 300 -> the right one is discovered last -> wins
    ("same first row", _paint([(60, 50, 4, 4), (300, 50, 4, 4)]), (300, 50, 4, 4)),
    # three equal areas (16): 4x4 at row 50, 4x4 at row 50 further right, 8x2 at row 250: the 8x2 is discovered last
    ("three equal areas, different shapes", _paint([(60, 50, 4, 4), (300, 50, 4, 4), (10, 250, 8, 2)]), (10, 250, 8, 2)),
    # the component discovered last is SMALLER: strict '>' lets an earlier-returned... no — it is returned first but loses
    # to the larger one found later in the list: 9 x 4 = 36 beats 3 x 3 = 9 wherever it sits
    ("larger beats later", _paint([(20, 10, 9, 4), (300, 100, 3, 3)]), (20, 10, 9, 4)),
    # tie on BOUNDING-BOX area, not pixel count: an L of 7 pixels spanning 4 x 4 (area 16) starting at row 20 and a full
    # 4 x 4 square (16 pixels) starting at row 10: equal box areas -> the one discovered last (the L, row 20) wins
    ("bounding box area, not pixel count", _paint([(100, 10, 4, 4), (200, 20, 1, 4), (200, 23, 4, 1)]), (200, 20, 4, 4)),
    # discovery order is decided by the FIRST pixel, not by the rectangle's top-left corner: component P = a column
    # x = 105, rows 10..13 plus a foot at (row 13, col 102..105) -> box (102, 10, 4, 4), first pixel (row 10, col 105);
    # component Q = the square (103, 10, 4, 4) shifted right: box (300, 10, 4, 4), first pixel (row 10, col 300).
    # Both first pixels are on row 10, col 105 < col 300 -> Q is discovered last -> Q wins although P's box starts further left
    ("first pixel decides", _paint([(105, 10, 1, 4), (102, 13, 4, 1), (300, 10, 4, 4)]), (300, 10, 4, 4)),
    # a diagonal link makes ONE component (8-connectivity): squares (10,10,3,3) and (13,13,3,3) touch at a corner -> box
    # (10, 10, 6, 6) = 36 > the lone 5 x 5 = 25 further down
    ("diagonal link merges", _paint([(10, 10, 3, 3), (13, 13, 3, 3), (200, 200, 5, 5)]), (10, 10, 6, 6)),
]
