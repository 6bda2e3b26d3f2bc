"""Build libdisprcnn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m disprcnn_amd.csrc.build [--force]

The .so stays in-tree (git-ignored, but it travels to the GPU box with the snapshot).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["tapconv.hip", "tapdirect.hip", "wino3d.hip", "wino3d_rb.hip", "wino2d.hip", "tapdeconv.hip", "deconvdirect.hip", "downdirect.hip", "stemconv.hip", "pointwise.hip", "linear.hip", "volume_ops.hip", "cout1_mfma.hip", "roi_ops.hip", "nms_ops.hip", "det_ops.hip", "post_ops.hip", "train_ops.hip", "bwd_ops.hip", "wgrad.hip", "wgrad_slide.hip", "conv16.hip", "conv16t.hip", "conv16x.hip", "ops16.hip", "convs16.hip", "convs16w.hip", "convs16d.hip", "convs16u.hip", "convs16r.hip", "s16_ops.hip"]
LIB = os.path.join(HERE, "libdisprcnn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _sources():
    return [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]


def _headers():
    return [os.path.join(HERE, "..", "..", "include", "disprcnn_hip.h")] + \
        [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith(".h")]


def source_digest():
    """sha256 over the kernel sources and headers (names + contents): what a committed PMC profile is valid for (profiles/summarize.py stores it,
    bench.py nulls `roofline.traffic` when it differs -- the GPU box has no .git to diff against)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_sources() + _headers(), key=os.path.basename):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _sources() + _headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs, jobs = [], []
    for src in _sources():
        obj = src[:-4] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in [src] + _headers()):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
        objs.append(obj)
    if jobs:                                    # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            list(ex.map(run, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
