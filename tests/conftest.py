import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Run order of the gpu-marked tests (the driver runs `pytest -m gpu -x`, so one failure hides everything after it): tests that compare
# with the oracle / the reference's recorded goldens first, along the hot path (a1..a7 kernels, whole-path goldens, a8/a12, the caller,
# post-processing, the 2D stage), then the fp16 and training parity tests, and tests that compare the build with ITSELF (reproducibility,
# graph replay, invariances, memory accounting, adjointness, round trips) last -- a self-comparison can never shadow an oracle test.
_FILE_ORDER = ["test_hip_parity", "test_hip_default_kernels", "test_backbone", "test_hip_caller", "test_hip_post", "test_hip_nms",
               "test_hip_detector2d", "test_detector2d", "test_hip_f16", "test_hip_train_caller", "test_hip_train", "test_hip_graph"]
_SELF_COMPARISON = ("reproducib", "graph", "invarian", "memory_flat", "adjoint", "round_trip", "roundtrip", "permutation", "refus",
                    "alone_vs_batch", "fails_loudly", "property")


def _gpu_rank(item):
    mod = item.module.__name__.rsplit(".", 1)[-1] if item.module is not None else ""
    frank = _FILE_ORDER.index(mod) if mod in _FILE_ORDER else len(_FILE_ORDER)
    name = item.name.lower()
    self_cmp = 1 if (mod == "test_hip_graph" or any(k in name for k in _SELF_COMPARISON)) else 0
    return (self_cmp, frank)


def pytest_collection_modifyitems(config, items):
    """Order the gpu tests (oracle parity first, self-comparisons last; stable within a group).  On a host without a GPU the gpu-marked
    tests are skipped, so a plain `pytest tests` reports CPU regressions only."""
    import torch
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if gpu_items:
        order = {id(it): i for i, it in enumerate(items)}
        gpu_sorted = sorted(gpu_items, key=lambda it: _gpu_rank(it) + (order[id(it)],))
        rest = [it for it in items if "gpu" not in it.keywords]
        items[:] = rest + gpu_sorted
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no GPU on this host)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
