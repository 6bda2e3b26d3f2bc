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
