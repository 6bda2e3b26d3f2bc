#!/usr/bin/env python3
"""Golden fixtures for NMS, recorded from the REFERENCE'S OWN CPU KERNEL (authoring container only).

    python oracle/build_ref.py && python tests/golden/make_golden_nms.py

Calls `nms` of the module oracle/build_ref.py compiles from /root/reference/disprcnn/csrc/cpu/nms_cpu.cpp (the function behind
`disprcnn._C.nms` for CPU tensors, csrc/nms.h:27).  Boxes / scores come from disprcnn_amd.utils.synth (rebuilt by the tests):
clustered proposals around a few objects, so that suppression chains occur; only the kept indices are stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import build_ref  # noqa: E402
from disprcnn_amd.utils import synth  # noqa: E402

CASES = [("n50", 50, 0.5), ("n200", 200, 0.7), ("n1000", 1000, 0.7), ("n1000_t3", 1000, 0.3), ("n130", 130, 0.5), ("n1", 1, 0.5), ("n4000", 4000, 0.7)]


def proposals(tag, n):
    """n boxes clustered around max(n // 12, 1) objects on a 1242 x 375 image + scores."""
    k = max(n // 12, 1)
    c = synth.hash_uniform(f"nms:{tag}:c", (k, 4), 0.0, 1.0)
    cx, cy = 40 + c[:, 0] * 1160, 30 + c[:, 1] * 310
    w, h = 20 + c[:, 2] * 200, 20 + c[:, 3] * 150
    which = (synth.hash_uniform(f"nms:{tag}:w", (n,), 0.0, 1.0) * k).long().clamp(max=k - 1)
    j = synth.hash_uniform(f"nms:{tag}:j", (n, 4), -0.25, 0.25)
    bx, by = cx[which] + j[:, 0] * w[which], cy[which] + j[:, 1] * h[which]
    bw, bh = w[which] * (1 + j[:, 2]), h[which] * (1 + j[:, 3])
    dets = torch.stack([bx - bw / 2, by - bh / 2, bx + bw / 2, by + bh / 2], 1).float()
    scores = synth.hash_uniform(f"nms:{tag}:s", (n,), 0.0, 1.0)
    return dets, scores


def main():
    ref = build_ref.load() or (build_ref.build() and build_ref.load())
    assert ref is not None
    out = {}
    for tag, n, thr in CASES:
        dets, scores = proposals(tag, n)
        keep = ref.nms(dets, scores, thr).numpy()
        out[f"{tag}_keep"] = keep.astype(np.int64)
        out[f"{tag}_thr"] = np.array(thr)
        print(tag, n, thr, "kept", len(keep))
    path = os.path.join(HERE, "nms_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
