"""PolygonZone known answers, derived BY HAND from the documented rules (reference call sites
main.py:108-119 builds the zone, players_tracker.py:364-365 applies it):

supervision 0.2x ``PolygonZone(polygon, frame_resolution_wh=(w, h))``:
  mask   = cv2.fillPoly(zeros((h + 1, w + 1)), [polygon], 1)     — interior AND boundary pixels are set;
  trigger: boxes are clipped to [0, w] x [0, h]; anchor = BOTTOM_CENTER = ((x1 + x2) / 2, y2); ceil to int;
           inside iff mask[anchor_y, anchor_x] != 0.

The shapes below have edges that are axis-aligned or at 45 degrees, where cv2's fixed-point edge walker has no
rounding freedom, so the expected masks can be written down exactly (general slopes stay *parity unpinned*)."""
import numpy as np

from padel_analytics_amd.detections import Detections, PolygonZone, polygon_to_mask


def test_rectangle_mask_includes_boundary():
    m = polygon_to_mask(np.array([[2, 2], [6, 2], [6, 5], [2, 5]]), (11, 9))
    want = np.zeros((9, 11), np.uint8)
    want[2:6, 2:7] = 1                      # x in [2, 6], y in [2, 5], edges and vertices included
    assert np.array_equal(m, want)


def test_right_triangle_mask_45_degrees():
    m = polygon_to_mask(np.array([[0, 0], [4, 0], [0, 4]]), (6, 6))
    want = np.zeros((6, 6), np.uint8)
    for y in range(6):
        for x in range(6):
            want[y, x] = 1 if x + y <= 4 else 0          # hypotenuse x + y = 4 passes through pixel centres
    assert np.array_equal(m, want)


def test_diamond_mask():
    m = polygon_to_mask(np.array([[3, 0], [6, 3], [3, 6], [0, 3]]), (7, 7))
    want = np.zeros((7, 7), np.uint8)
    for y in range(7):
        for x in range(7):
            want[y, x] = 1 if abs(x - 3) + abs(y - 3) <= 3 else 0
    assert np.array_equal(m, want)


def test_mask_is_w_plus_1_by_h_plus_1_and_anchor_rules():
    w, h = 10, 8
    z = PolygonZone(np.array([[2, 2], [6, 2], [6, 5], [2, 5]]), frame_resolution_wh=(w, h))
    assert z.mask.shape == (h + 1, w + 1)
    boxes = np.array([
        [1.2, 0.0, 2.6, 4.0],      # anchor (1.9, 4.0) -> ceil (2, 4): on the left edge            -> inside
        [0.2, 0.0, 1.7, 3.0],      # anchor (0.95, 3) -> (1, 3): one pixel left of the edge        -> outside
        [5.0, 1.0, 7.0, 5.0],      # anchor (6, 5): the bottom-right VERTEX                        -> inside
        [5.0, 1.0, 7.2, 5.0],      # anchor (6.1, 5) -> ceil 7: past the right edge                -> outside
        [3.0, 0.0, 5.0, 5.01],     # anchor (4, 5.01) -> ceil y = 6: below the bottom edge          -> outside
        [3.0, 0.0, 5.0, 1.01],     # anchor (4, 1.01) -> (4, 2): on the top edge                    -> inside
        [3.0, 0.0, 5.0, 1.0],      # anchor (4, 1): one above the top edge                          -> outside
        [2.0, 3.0, 40.0, 4.0],     # x2 clipped to w = 10: anchor ((2 + 10) / 2, 4) = (6, 4)        -> inside
        [-30.0, 3.0, 4.0, 4.0],    # x1 clipped to 0: anchor (2, 4)                                  -> inside
        [3.0, 2.0, 5.0, 90.0],     # y2 clipped to h = 8: anchor (4, 8)                              -> outside
    ], np.float32)
    want = [True, False, True, False, False, True, False, True, True, False]
    got = z.trigger(Detections(boxes, np.ones(len(boxes), np.float32), np.zeros(len(boxes), int)))
    assert got.tolist() == want
    assert z.current_count == sum(want)
    assert z.trigger_boxes(boxes).tolist() == want


def test_far_corner_anchor_is_addressable():
    """A box touching the bottom-right corner of the frame has its anchor at (w, h): the (h+1, w+1) mask exists
    precisely so that this index is valid."""
    w, h = 10, 8
    z = PolygonZone(np.array([[0, 0], [w, 0], [w, h], [0, h]]), frame_resolution_wh=(w, h))
    boxes = np.array([[10.0, 0.0, 10.0, 8.0], [9.0, 7.0, 11.0, 9.0]], np.float32)
    assert z.trigger_boxes(boxes).tolist() == [True, True]


def test_empty_detections():
    z = PolygonZone(np.array([[0, 0], [4, 0], [0, 4]]), frame_resolution_wh=(5, 5))
    assert z.trigger(Detections.empty()).shape == (0,)
