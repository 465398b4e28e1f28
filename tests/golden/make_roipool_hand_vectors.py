"""tests/golden/make_roipool_hand_vectors.py -- ROIPool known-answer vectors DERIVED BY HAND from the reference's CUDA kernel.

The reference has no CPU ROIPool (csrc/ROIPool.h:47,68 -> AT_ERROR "Not implemented on the CPU"), so nothing can be recorded from
it here; SURVEY.md 8c calls ROIPool "parity unpinned".  These vectors pin it the only other way: every case below is worked out
with pencil and paper from external/maskrcnn_benchmark/csrc/cuda/ROIPool_cuda.cu:40-132, line by line, and what is committed is
the BIN TABLE of each case -- for every pooled cell the half-open row / column ranges [hstart, hend) x [wstart, wend) the kernel
scans after rounding, the "+1" extents, floor / ceil and clipping -- with the source line that decides the case.

This script contains NO ROIPool arithmetic.  It only turns a hand-written bin table into numbers, using three feature channels
whose maximum over ANY rectangle is known without computing anything:
    ch0 = 10 h + w (+100 per image)   strictly increasing in h and w  -> the max sits in the bin's LAST row, LAST column
    ch1 = -(10 h + w) - 1 (-100 ...)  strictly decreasing, all < 0     -> FIRST row, FIRST column (exercises the -FLT_MAX start, :79)
    ch2 = 5.0 everywhere              all ties                         -> the FIRST cell scanned wins (strict `>`, :86-88)
An empty bin yields value 0 and argmax -1 (:76-81).  argmax = h * width + w inside the image plane (:85).
Output: roipool_hand_vectors.json (inputs + expected out / argmax / backward), checked by tests/test_oracle_golden.py (the C
restatement) and by the GPU kernel tests (the HIP kernels, both layouts).

    python tests/golden/make_roipool_hand_vectors.py
"""
import json
import os

H = W = 8
E = None            # an empty bin

# Each case: rois [K][5] = (batch index as float, x1, y1, x2, y2), spatial_scale, pooled (ph, pw), and per roi the bin table
# bins[ph][pw] = (hstart, hend, wstart, wend) or E.  `why` = the derivation, `line` = the source line(s) that decide the case.
CASES = [
    dict(name="plus_one_extent", B=1, rois=[[0, 1, 1, 4, 4]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:61-62 (roi_width = max(end - start + 1, 1))",
         why="start = round(1) = 1, end = round(4) = 4 -> width = height = 4 - 1 + 1 = 4 (NOT 3); bin = 4 / 2 = 2.0; "
             "ph0: floor(0) = 0 .. ceil(2) = 2, + start 1 -> [1,3); ph1: floor(2) = 2 .. ceil(4) = 4 -> [3,5); same along w",
         bins=[[[(1, 3, 1, 3), (1, 3, 3, 5)], [(3, 5, 1, 3), (3, 5, 3, 5)]]]),
    dict(name="round_half_away_from_zero", B=1, rois=[[0, 0.5, 1.5, 2.5, 3.5]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:55-58 (CUDA round(): halves away from zero, not to even)",
         why="round(0.5) = 1, round(1.5) = 2, round(2.5) = 3, round(3.5) = 4 (to-even would give 0, 2, 2, 4) -> start (w 1, h 2), "
             "end (w 3, h 4): width = height = 3; bin = 1.5; ph0: 0 .. ceil(1.5) = 2 -> [2,4); ph1: floor(1.5) = 1 .. ceil(3) = 3 -> "
             "[3,5); pw0 -> [1,3), pw1 -> [2,4): neighbouring bins OVERLAP by one row / column",
         bins=[[[(2, 4, 1, 3), (2, 4, 2, 4)], [(3, 5, 1, 3), (3, 5, 2, 4)]]]),
    dict(name="spatial_scale_sixteenth", B=1, rois=[[0, 24, 40, 72, 104]], scale=0.0625, pooled=(2, 2),
         line="ROIPool_cuda.cu:55-58 (coordinates are scaled, THEN rounded)",
         why="24/16 = 1.5 -> 2, 40/16 = 2.5 -> 3, 72/16 = 4.5 -> 5, 104/16 = 6.5 -> 7 (all exact in fp32): start (w 2, h 3), end (w 5, h 7); "
             "width 4, height 5; bin_w = 2, bin_h = 2.5; ph0: 0 .. ceil(2.5) = 3 -> [3,6); ph1: floor(2.5) = 2 .. ceil(5) = 5 -> [5,8); "
             "pw0 -> [2,4), pw1 -> [4,6)",
         bins=[[[(3, 6, 2, 4), (3, 6, 4, 6)], [(5, 8, 2, 4), (5, 8, 4, 6)]]]),
    dict(name="malformed_roi_forced_to_1x1", B=1, rois=[[0, 5, 5, 2, 2]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:60-62 ('Force malformed ROIs to be 1x1')",
         why="start 5, end 2: width = max(2 - 5 + 1, 1) = max(-2, 1) = 1; bin = 0.5; ph0: floor(0) = 0 .. ceil(0.5) = 1 -> [5,6); "
             "ph1: floor(0.5) = 0 .. ceil(1.0) = 1 -> [5,6): all four cells pool the single pixel (5,5)",
         bins=[[[(5, 6, 5, 6), (5, 6, 5, 6)], [(5, 6, 5, 6), (5, 6, 5, 6)]]]),
    dict(name="clipped_at_the_origin_empty_bins", B=1, rois=[[0, -3, -2, 1, 1]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:71-81 (clip to [0, height]; is_empty -> value 0, argmax -1)",
         why="start (w -3, h -2), end (1, 1): width 1 + 3 + 1 = 5, height 1 + 2 + 1 = 4; bin_w = 2.5, bin_h = 2; ph0: [0,2) - 2 = [-2,0) "
             "-> clipped [0,0): EMPTY; ph1: [2,4) - 2 = [0,2); pw0: 0 .. ceil(2.5) = 3, - 3 = [-3,0) -> [0,0): EMPTY; pw1: floor(2.5) = 2 "
             ".. ceil(5) = 5, - 3 = [-1,2) -> [0,2).  Three cells are empty: value 0 although channel 1 is negative everywhere",
         bins=[[[E, E], [E, (0, 2, 0, 2)]]]),
    dict(name="clipped_at_the_far_edge", B=1, rois=[[0, 6, 6, 9, 9]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:71-75 (min(., height) / min(., width))",
         why="start 6, end 9: width 4, bin 2; ph0: [0,2) + 6 = [6,8); ph1: [2,4) + 6 = [8,10) -> clipped [8,8): EMPTY",
         bins=[[[(6, 8, 6, 8), E], [E, E]]]),
    dict(name="single_pixel_roi_pooled_2x2", B=1, rois=[[0, 3, 4, 3, 4]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:63-70 (bins narrower than a pixel repeat it)",
         why="start = end = (w 3, h 4): width = height = 1; bin 0.5; both ph: floor(..) = 0 .. ceil(..) = 1 -> [4,5); both pw -> [3,4)",
         bins=[[[(4, 5, 3, 4), (4, 5, 3, 4)], [(4, 5, 3, 4), (4, 5, 3, 4)]]]),
    dict(name="batch_index_selects_the_image", B=2, rois=[[1, 1, 1, 4, 4], [0, 1, 1, 4, 4]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:54,82-83 (roi_batch_ind = (int) rois[0]; plane offset (b * channels + c) * H * W)",
         why="same geometry as plus_one_extent on image 1 (values + 100) and then on image 0; argmax stays an index INSIDE the plane",
         bins=[[[(1, 3, 1, 3), (1, 3, 3, 5)], [(3, 5, 1, 3), (3, 5, 3, 5)]], [[(1, 3, 1, 3), (1, 3, 3, 5)], [(3, 5, 1, 3), (3, 5, 3, 5)]]]),
    dict(name="non_square_pooled_1x3", B=1, rois=[[0, 0, 0, 5, 1]], scale=1.0, pooled=(1, 3),
         line="ROIPool_cuda.cu:47-50,63-66 (pooled_height and pooled_width are independent)",
         why="width 6, height 2; bin_h = 2 / 1 = 2, bin_w = 6 / 3 = 2; ph0 -> [0,2); pw0 -> [0,2), pw1 -> [2,4), pw2 -> [4,6)",
         bins=[[[(0, 2, 0, 2), (0, 2, 2, 4), (0, 2, 4, 6)]]]),
    dict(name="negative_half_rounds_away_from_zero", B=1, rois=[[0, -0.5, -0.5, 2.5, 2.5]], scale=1.0, pooled=(2, 2),
         line="ROIPool_cuda.cu:55-58 (round(-0.5) = -1)",
         why="start = round(-0.5) = -1 (to-even: 0), end = round(2.5) = 3: width 3 + 1 + 1 = 5; bin 2.5; ph0: 0 .. ceil(2.5) = 3, - 1 = "
             "[-1,2) -> [0,2); ph1: floor(2.5) = 2 .. ceil(5) = 5, - 1 = [1,4).  (With start 0 the bins would be [0,2) / [2,4): channel 1, "
             "whose maximum sits in a bin's FIRST row / column, tells the two apart)",
         bins=[[[(0, 2, 0, 2), (0, 2, 1, 4)], [(1, 4, 0, 2), (1, 4, 1, 4)]]]),
    dict(name="whole_map_7x7", B=1, rois=[[0, 0, 0, 7, 7]], scale=1.0, pooled=(7, 7),
         line="ROIPool_cuda.cu:63-75 (the STEP configuration: 7x7 cells, fractional bins 8/7)",
         why="width 8; bin = 8/7 = 1.1428572f; floor(ph * bin) = 0,1,2,3,4,5,6 (1.14, 2.29, 3.43, 4.57, 5.71, 6.86); ceil((ph+1) * bin) = "
             "2,3,4,5,6,7 and for ph = 6: 7 * 1.1428572f = 8.0000004 -> fp32 8.0 -> 8 (a 9 would be clipped to 8 anyway): bins [ph, ph+2)",
         bins=[[[(i, i + 2, j, j + 2) for j in range(7)] for i in range(7)]]),
    dict(name="whole_frame_box_scaled_7x7", B=1, rois=[[0, 0, 0, 127, 127]], scale=0.0625, pooled=(7, 7),
         line="ROIPool_cuda.cu:55-75 (a whole-frame tube: 127/16 = 7.9375 rounds UP to 8 = one past the last column)",
         why="end = round(7.9375) = 8: width 8 - 0 + 1 = 9; bin = 9/7 = 1.2857143f; floor(ph * bin) = 0,1,2,3,5,6,7 (1.29, 2.57, 3.86, 5.14, "
             "6.43, 7.71); ceil((ph+1) * bin) = 2,3,4,6,7,8 and for ph = 6: 9.0 -> 9 -> clipped to 8: rows [0,2) [1,3) [2,4) [3,6) [5,7) [6,8) "
             "[7,8); the same along w",
         bins=[[[(hs, he, ws, we) for (ws, we) in ((0, 2), (1, 3), (2, 4), (3, 6), (5, 7), (6, 8), (7, 8))]
                for (hs, he) in ((0, 2), (1, 3), (2, 4), (3, 6), (5, 7), (6, 8), (7, 8))]]),
]


def feature(B):
    """[B][3][H][W] as nested lists (see the module docstring)"""
    return [[[[float(10 * h + w + 100 * b) for w in range(W)] for h in range(H)],
             [[float(-(10 * h + w + 100 * b) - 1) for w in range(W)] for h in range(H)],
             [[5.0 for _ in range(W)] for _ in range(H)]] for b in range(B)]


def expected(case):
    out, arg = [], []
    for roi, table in zip(case["rois"], case["bins"]):
        b = int(roi[0])
        o3, a3 = [], []
        for c in range(3):
            o2, a2 = [], []
            for row in table:
                o1, a1 = [], []
                for cell in row:
                    if cell is E:
                        o1.append(0.0); a1.append(-1)
                        continue
                    hs, he, ws, we = cell
                    assert 0 <= hs < he <= H and 0 <= ws < we <= W, (case["name"], cell)
                    h, w = (he - 1, we - 1) if c == 0 else (hs, ws)          # corner rule of the three channels
                    v = (10 * h + w + 100 * b) if c == 0 else (-(10 * h + w + 100 * b) - 1 if c == 1 else 5.0)
                    o1.append(float(v)); a1.append(h * W + w)
                o2.append(o1); a2.append(a1)
            o3.append(o2); a3.append(a2)
        out.append(o3); arg.append(a3)
    return out, arg


def backward(case, arg):
    """grad_out[k][c][i][j] = 1 + running index; grad_in = sum of the gradients of the cells whose argmax is the pixel (:121-127:
    atomicAdd at argmax, cells with argmax -1 are dropped) -- a scatter of the hand-derived argmax table, nothing else."""
    B = case["B"]
    gin = [[[[0.0] * W for _ in range(H)] for _ in range(3)] for _ in range(B)]
    gout, n = [], 0
    for k, roi in enumerate(case["rois"]):
        b = int(roi[0])
        g3 = []
        for c in range(3):
            g2 = []
            for i, row in enumerate(arg[k][c]):
                g1 = []
                for j, a in enumerate(row):
                    n += 1
                    g1.append(float(n))
                    if a >= 0:
                        gin[b][c][a // W][a % W] += float(n)
                g2.append(g1)
            g3.append(g2)
        gout.append(g3)
    return gout, gin


def main():
    vec = []
    for case in CASES:
        out, arg = expected(case)
        gout, gin = backward(case, arg)
        vec.append(dict(name=case["name"], decided_by=case["line"], derivation=case["why"], B=case["B"], C=3, H=H, W=W,
                        feature=feature(case["B"]), rois=case["rois"], spatial_scale=case["scale"], pooled=list(case["pooled"]),
                        out=out, argmax=arg, grad_out=gout, grad_in=gin))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "roipool_hand_vectors.json")
    json.dump(vec, open(path, "w"))
    print("%d hand-derived ROIPool vectors -> %s" % (len(vec), path))


if __name__ == "__main__":
    main()
