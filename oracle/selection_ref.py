"""oracle/selection_ref.py -- TEST INFRASTRUCTURE ONLY: scalar restatement of the reference's tube IoU, the arithmetic the
training sample selection ranks proposals by.  One pair at a time, np.float32 scalars, the reference's operation order:

    box IoU   utils/tube_utils.py:269-306   intersection only if both extents > 0, areas WITHOUT the +1 pixel convention,
                                            union = area1 + area2 - intersection, all in fp32
    tube IoU  utils/tube_utils.py:308-351   per-frame box IoUs added up as Python floats (double), divided by T, stored as fp32;
                                            a pair whose either tube sums to zero contributes zeros (padding tubes)

Pinned by tests/golden/selection_golden.npz (outputs of the reference's own compute_tube_iou); step_amd/selection.py's
array version is then checked against this one on random inputs (tests/test_host_logic.py)."""
import numpy as np


def box_iou_pair(b1, b2):
    xmin, ymin = max(b1[0], b2[0]), max(b1[1], b2[1])
    xmax, ymax = min(b1[2], b2[2]), min(b1[3], b2[3])
    iw, ih = np.maximum(xmax - xmin, np.float32(0)), np.maximum(ymax - ymin, np.float32(0))
    inter = iw * ih if (iw > 0 and ih > 0) else np.float32(0)
    union = (b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.float32(inter / union)


def tube_iou(t1, t2):
    t1, t2 = np.asarray(t1, np.float32), np.asarray(t2, np.float32)
    out = np.zeros((t1.shape[0], t2.shape[0]), np.float32)
    T = t1.shape[1]
    for i in range(t1.shape[0]):
        for j in range(t2.shape[0]):
            acc = 0.0
            if np.sum(t1[i]) and np.sum(t2[j]):
                for t in range(T):
                    acc += float(box_iou_pair(t1[i, t], t2[j, t]))
            out[i, j] = acc / T if T > 0 else acc
    return out
