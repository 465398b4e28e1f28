"""oracle/postprocess_ref.py -- plain numpy restatement of the reference's evaluation / post-processing loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows test.py:157-210 (the same loop is inlined in
train.py:512-573 and demo.py:123-174) statement for statement, with the C restatement of the CPU NMS operator
(oracle.nms <-> cpu/nms_cpu.cpp:29-89).  Pinned by tests/golden/postprocess_golden.npz, which oracle/make_golden.py
records by executing the reference's own loop on a seeded `history` (tests/test_oracle_golden.py).
"""
import numpy as np

from . import nms as _nms


def valid_tubes(boxes, width=400, height=400):
    """utils/tube_utils.py:59-93 on [n,4] boxes: clamp into the image, then replace boxes that are not at least 2 px
    wide and high by the whole image."""
    b = np.array(boxes, np.float32, copy=True).reshape(-1, 4)
    b[:, 0] = np.maximum(0, b[:, 0])                                  # :73-76
    b[:, 1] = np.maximum(0, b[:, 1])
    b[:, 2] = np.minimum(width, b[:, 2])
    b[:, 3] = np.minimum(height, b[:, 3])
    for i in range(b.shape[0]):                                       # :84-88
        if not (b[i, 0] < b[i, 2] - 2 and b[i, 1] < b[i, 3] - 2):
            b[i, :2] = 0
            b[i, 2] = width
            b[i, 3] = height
    return b


def postprocess(history, num_classes=60, conf_thresh=0.01, nms_thresh=0.4, evaluate_topk=-1, topk=-1, width=400, height=400):
    """history: list of dicts {pred_prob [N,Tl,classes], pred_loc [N,Tl,4], tubes_nums} (numpy).
    Returns rows (iteration, clip, class, x1, y1, x2, y2 normalised, score, tube index within the clip) in the order
    test.py writes them to its per-iteration CSV files."""
    rows = []
    for i in range(len(history)):                                     # test.py:157
        prob = np.asarray(history[i]["pred_prob"], np.float32)
        prob = prob[:, int(prob.shape[1] / 2)]                        # :159 middle frame
        tubes = np.asarray(history[i]["pred_loc"], np.float32)
        tubes = tubes[:, int(tubes.shape[1] / 2)]                     # :161
        nums = history[i]["tubes_nums"]
        count = 0
        for b in range(len(nums)):                                    # :166
            cur_p = prob[count:count + nums[b]]
            cur_t = tubes[count:count + nums[b]]
            count += nums[b]
            all_scores, all_boxes, all_idx = [], [], []
            for c in range(num_classes):                              # :178
                scores = cur_p[:, c].reshape(-1)
                mask = scores > conf_thresh                           # :180
                scores = scores[mask]
                idx = np.where(mask)[0]
                if len(scores) == 0:                                  # :183-186
                    all_scores.append([]); all_boxes.append([]); all_idx.append([])
                    continue
                boxes = valid_tubes(cur_t[mask])                      # :187-191 (valid_tubes' own 400 x 400 default)
                keep = _nms(boxes, scores, nms_thresh)                # :192 kept original indices, ascending
                boxes = boxes[keep]
                scores = scores[keep]
                idx = idx[keep]
                boxes[:, ::2] /= width                                # :197-198
                boxes[:, 1::2] /= height
                all_scores.append(scores); all_boxes.append(boxes); all_idx.append(idx)
            lst = [(s, c, j) for c, sc in enumerate(all_scores) for j, s in enumerate(sc)]     # :204
            if evaluate_topk > 0:                                     # :205-208: stable ascending sort, reversed, then args.topk
                lst.sort(key=lambda x: x[0])
                lst = lst[::-1]
                lst = lst[:topk]
            for s, c, j in lst:                                       # :210-218
                bx = all_boxes[c][j]
                rows.append((i, b, c, bx[0], bx[1], bx[2], bx[3], s, all_idx[c][j]))
    return rows


def csv_line(video_name, fid, box, label, score):
    """The row format of test.py:212-218."""
    return "{0},{1:04},{2:.4},{3:.4},{4:.4},{5:.4},{6},{7:.4}\n".format(video_name, fid, box[0], box[1], box[2], box[3], label, score)
