"""iDispNet / PSMNet operator -- MI355X-native.

Drop-in for ``disprcnn.modeling.psmnet.stackhourglass.PSMNet`` (reference
stackhourglass.py:54-174): same constructor, same ``forward(inputs)`` contract
(dict with 'left'/'right' or a 2-sequence of [N,3,H,W] fp32 -> eval: [N,H,W]; train:
3-tuple), same ``state_dict()`` keys/shapes so ``bestmodel.pth['model']`` and KITTI2015
PSMNet checkpoints load.  The forward pass is executed by hand-written HIP kernels
(cost volume, MFMA tap-convolutions, fused soft-argmin) through the C ABI of
libdisprcnn_hip.so; there is no torch/CPU fallback.
"""
import torch
from torch import nn

from .submodule import conv_bn_3d, deconv_bn_3d, feature_extraction, reference_init_, with_relu_slots


class hourglass(nn.Module):
    """Weights of one stacked-hourglass block (reference stackhourglass.py:7-30).
    conv1/conv3/conv4 = [convbn3d, relu]; conv2 = convbn3d; conv5/conv6 = [deconv, bn]."""

    def __init__(self, inplanes):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(conv_bn_3d(c, 2 * c, stride=2), nn.ReLU(inplace=True))
        self.conv2 = conv_bn_3d(2 * c, 2 * c)
        self.conv3 = nn.Sequential(conv_bn_3d(2 * c, 2 * c, stride=2), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(conv_bn_3d(2 * c, 2 * c), nn.ReLU(inplace=True))
        self.conv5 = deconv_bn_3d(2 * c, 2 * c)
        self.conv6 = deconv_bn_3d(2 * c, c)

    def forward(self, *a):
        raise RuntimeError("hourglass is a parameter holder; run it through PSMNet (HIP engine)")


def _head():
    return nn.Sequential(conv_bn_3d(32, 32), nn.ReLU(inplace=True),
                         nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False))


class PSMNet(nn.Module):
    def __init__(self, maxdisp, mindisp=0, input_size=224, is_module=False, feature_level=1,
                 single_modal_weight_average=False, conv_layers=(), use_disparity_regression=True):
        # only maxdisp/mindisp are used, as in the reference (stackhourglass.py:55-61)
        super().__init__()
        self.maxdisp, self.mindisp = maxdisp, mindisp
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(*with_relu_slots(conv_bn_3d(64, 32), conv_bn_3d(32, 32)))
        self.dres1 = nn.Sequential(conv_bn_3d(32, 32), nn.ReLU(inplace=True), conv_bn_3d(32, 32))
        self.dres2, self.dres3, self.dres4 = hourglass(32), hourglass(32), hourglass(32)
        self.classif1, self.classif2, self.classif3 = _head(), _head(), _head()
        reference_init_(self)
        self._rt = None
        # "f32" (the reference's precision, default) or "f16": cost volume + 3D regressor on fp16-storage tensors with fp32
        # accumulation (inference only; BASELINE configs[3]).  Not a constructor argument: the reference signature is kept.
        self.regressor_storage = "f32"
        self.regressor_math = "auto"          # "f32": every 3D layer on the fp32 MFMA; "auto"/"f16x2": eval runs the 3x3x3 layers of the regressor in split-f16 (runtime._use_s16)
        self.feature_math = "auto"            # the same switch for the stride-1 undilated 3x3 layers of the 2D feature CNN (runtime._use_s16_2d)
        self.graph_eval = "auto"              # eval batches of <= runtime.GRAPH_MAX_UNITS units replay a captured HIP graph (True: always, False: never)
        self.feature_storage = "f32"          # "f16" (with regressor_storage "f16"): the 2D feature CNN in fp16 storage as well

    # ------------------------------------------------------------------ engine plumbing
    def _runtime(self, device):
        from .runtime import PSMNetRuntime
        if self._rt is None or self._rt.device != device:
            self._rt = PSMNetRuntime(self, device)
        return self._rt

    @staticmethod
    def _unpack(inputs):
        if isinstance(inputs, dict):
            return inputs["left"], inputs["right"]
        if len(inputs) == 2:
            return inputs[0], inputs[1]
        raise ValueError("PSMNet.forward expects {'left','right'} or a pair of tensors")

    def forward(self, inputs):
        left, right = self._unpack(inputs)
        if left.shape != right.shape or left.dim() != 4 or left.shape[1] != 3:
            raise ValueError(f"expected two [N,3,H,W] tensors, got {tuple(left.shape)} / {tuple(right.shape)}")
        return self._runtime(left.device).forward_images(left, right, self.training)

    def forward_from_features(self, left_feat, right_feat, out_hw):
        """Config-A entry point: [N,32,H/4,W/4] feature pairs -> disparity [N,H,W].
        The reference cannot run its 2D CNN below 224x224 (fixed AvgPool2d(56), submodule.py:76-90), so the
        112x112x48 headline shape enters the path here: cost volume -> 3D regressor -> soft-argmin."""
        return self._runtime(left_feat.device).forward_features(left_feat, right_feat, out_hw, self.training)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        if self._rt is not None:
            self._rt.invalidate()
