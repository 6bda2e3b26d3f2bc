"""CPU ORACLE (test infrastructure) for greedy NMS.

Restated from the reference: csrc/cpu/nms_cpu.cpp:5-75 (areas with the legacy +1, descending score order, suppression when
IoU >= threshold, result = ascending original indices of the survivors); `strict=True` gives the CUDA op's test instead
(csrc/cuda/nms.cu:56: IoU > threshold).  float32 arithmetic like the reference's scalar_t = float.
Pinned by tests/golden/nms_golden.npz, recorded from the reference's own nms_cpu compiled by oracle/build_ref.py
(tests/golden/make_golden_nms.py); tests/test_oracle_nms.py also checks it live against that binary when present.
"""
import numpy as np


def nms(dets, scores, threshold, strict=False):
    dets = np.asarray(dets, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = dets.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    f = np.float32
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = ((x2 - x1 + f(1)) * (y2 - y1 + f(1))).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    thr = f(threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        if rest.size == 0:
            continue
        w = np.maximum(f(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + f(1)).astype(np.float32)
        h = np.maximum(f(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + f(1)).astype(np.float32)
        inter = (w * h).astype(np.float32)
        ovr = (inter / ((areas[i] + areas[rest]).astype(np.float32) - inter).astype(np.float32)).astype(np.float32)
        suppressed[rest[(ovr > thr) if strict else (ovr >= thr)]] = True
    return np.nonzero(~suppressed)[0].astype(np.int64)
