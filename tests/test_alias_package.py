"""The reference's import names resolve to the HIP implementation: tools/train_net.py:10-20 / tools/test_net.py of the
reference drive the path with `disprcnn.*` imports; here `disprcnn` is an alias package over `disprcnn_amd` (same module objects)."""
import importlib

import pytest


def test_reference_import_names_resolve_to_the_hip_modules():
    from disprcnn.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn.layers import ROIAlign, nms
    from disprcnn.modeling.backbone import build_backbone
    from disprcnn.modeling.detector import build_detection_model
    from disprcnn.utils.loss_utils import PSMLoss
    from disprcnn.structures.bounding_box import BoxList
    from disprcnn.structures.disparity import DisparityMap
    import disprcnn_amd.modeling.psmnet.stackhourglass as real
    import disprcnn_amd.layers as real_layers
    assert PSMNet is real.PSMNet and ROIAlign is real_layers.ROIAlign and nms is real_layers.nms
    assert importlib.import_module("disprcnn.modeling.psmnet.stackhourglass") is real          # the same module object, not a copy
    assert build_backbone.__module__.startswith("disprcnn_amd.") and build_detection_model.__module__.startswith("disprcnn_amd.")
    assert PSMLoss.__module__.startswith("disprcnn_amd.") and BoxList.__module__.startswith("disprcnn_amd.")
    assert DisparityMap.__module__.startswith("disprcnn_amd.")
    m = PSMNet(48, 0)          # the reference constructor signature (stackhourglass.py:55-58)
    assert len(m.state_dict()) == 514


def test_unknown_names_fail_like_a_missing_reference_module():
    with pytest.raises(ImportError):
        importlib.import_module("disprcnn.modeling.pointnet_module")       # out of scope: not provided


def test_aliasing_leaves_the_real_modules_import_identity_alone():
    """importlib stamps the alias spec onto the module create_module returns; the loader puts the real one back, so the real
    module's relative imports keep resolving through its own package and no ImportWarning ('__package__ != __spec__.parent') appears."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", ImportWarning)
        mod = importlib.import_module("disprcnn.structures.disparity")
        import disprcnn_amd.structures.disparity as real
        assert mod is real
        assert real.__spec__.name == "disprcnn_amd.structures.disparity" and real.__spec__.parent == "disprcnn_amd.structures"
        assert real.__package__ == "disprcnn_amd.structures" and not hasattr(real, "__path__")        # a plain module stays one
        pkg = importlib.import_module("disprcnn.structures")
        assert pkg.__spec__.name == "disprcnn_amd.structures" and list(pkg.__path__)                  # a package keeps its search path
        importlib.import_module("disprcnn.utils.loss_utils")                                           # (lazy relative imports inside still work)


def test_extension_surface_of_vision_cpp_is_importable():
    """`from disprcnn import _C` (reference layers/roi_align.py:3, layers/nms.py:3): the pybind names of csrc/vision.cpp:7-15."""
    import inspect
    from disprcnn import _C
    import disprcnn_amd._C as real
    assert _C is real
    for name in ("nms", "roi_align_forward", "roi_align_backward", "roi_pool_forward", "roi_pool_backward", "sigmoid_focalloss_forward",
                 "sigmoid_focalloss_backward"):
        assert callable(getattr(_C, name)), name
    # positional signatures of csrc/nms.h:12 and csrc/ROIAlign.h:12,28
    assert list(inspect.signature(_C.nms).parameters) == ["dets", "scores", "threshold"]
    assert list(inspect.signature(_C.roi_align_forward).parameters) == ["input", "rois", "spatial_scale", "pooled_height", "pooled_width", "sampling_ratio"]
    assert list(inspect.signature(_C.roi_align_backward).parameters) == ["grad", "rois", "spatial_scale", "pooled_height", "pooled_width", "batch_size",
                                                                         "channels", "height", "width", "sampling_ratio"]
    import torch
    with pytest.raises(RuntimeError):                       # CPU tensors: no CPU kernels in this build (the reference's CPU backward raises too)
        _C.roi_align_forward(torch.zeros(1, 1, 4, 4), torch.zeros(1, 5), 1.0, 2, 2, 0)
    with pytest.raises(NotImplementedError):
        _C.roi_pool_forward(torch.zeros(1, 1, 4, 4), torch.zeros(1, 5), 1.0, 2, 2)
