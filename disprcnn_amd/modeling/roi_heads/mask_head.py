"""ROI mask head -- drop-in for ``disprcnn.modeling.roi_heads.mask_head`` in its shipped configuration (inference): ROIMaskHead
(mask_head.py:33-105) = MaskRCNNFPNFeatureExtractor (roi_mask_feature_extractors.py:16-64) + MaskRCNNC4Predictor
(roi_mask_predictors.py:9-29) + MaskPostProcessor (inference.py:13-60; no masker in the shipped config).

state_dict keys as in the reference (``feature_extractor.mask_fcn{1..4}``, ``predictor.{conv5_mask,mask_fcn_logits}``).  Data path:
14x14 ROIAlign (sampling ratio 2) of the LEFT pyramid at the detections (HIP), four conv3x3(256)+ReLU on the HIP engine (the
backbone's Winograd kernel: 14x14 maps, batch = detections), then ConvTranspose2d(256,256,2,2)+ReLU and the 1x1 class logits -- a
2x2/stride-2 transposed convolution has no overlapping taps, so both are plain GEMMs over the pixels (library GEMM) -- sigmoid, and
the mask of each detection's own label."""
import torch
from torch import nn

from ...structures.bounding_box import BoxList
from ... import engine as E
from ..head_ops import BlockedConv2d, gemm_bias_act
from ..poolers import Pooler


class MaskRCNNFPNFeatureExtractor(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        m = cfg.MODEL.ROI_MASK_HEAD
        if m.USE_GN or m.DILATION != 1:
            raise NotImplementedError("mask head with GroupNorm / dilation is not part of the shipped configs")
        self.pooler = Pooler((m.POOLER_RESOLUTION, m.POOLER_RESOLUTION), m.POOLER_SCALES, m.POOLER_SAMPLING_RATIO)
        self.blocks, nxt, self._engine = [], in_channels, []
        for i, feat in enumerate(m.CONV_LAYERS, 1):
            name = f"mask_fcn{i}"
            conv = nn.Conv2d(nxt, feat, kernel_size=3, stride=1, padding=1)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)
            self.add_module(name, conv)
            self.blocks.append(name)
            self._engine.append(BlockedConv2d(conv, relu=True))
            nxt = feat
        self.out_channels = nxt
        self._bufs = {}          # (h, w, device) -> two blocked ping-pong tensors sized for a capacity bucket of the detection count

    def forward(self, x, proposals):
        """ROIAlign, then the conv chain WITHOUT leaving the engine's blocked layout (round 3): one conversion in, one out."""
        x = self.pooler(x, proposals)
        n, c, h, w = x.shape
        if n == 0:
            return x.new_zeros(0, self.out_channels, h, w)
        key = (h, w, x.device)
        cmax = max([c] + [e.cout for e in self._engine])
        buf = self._bufs.get(key)
        if buf is None or buf[0].N < n:
            cap = E.bucket_units(n)
            buf = self._bufs[key] = tuple(E.Blocked(cap, cmax, 1, h, w, 0, 1, 1, x.device) for _ in range(2))
            while len(self._bufs) > 4:
                self._bufs.pop(next(iter(self._bufs)))
        view = lambda b, ch: E.Blocked(n, ch, 1, h, w, 0, 1, 1, x.device, storage=b.storage) if ch == cmax else None
        if any(e.cout != cmax for e in self._engine) or c != cmax:
            raise NotImplementedError("mask head conv chain with varying widths")      # (the shipped CONV_LAYERS are 256 throughout)
        src, dst = view(buf[0], c), view(buf[1], cmax)
        src.from_dense(x.float())
        for conv in self._engine:
            conv(src, dst)
            src, dst = dst, src
        return src.to_dense()[:, :, 0]


class MaskRCNNC4Predictor(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        ncls, dim = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES, cfg.MODEL.ROI_MASK_HEAD.CONV_LAYERS[-1]
        self.conv5_mask = nn.ConvTranspose2d(in_channels, dim, 2, 2, 0)
        self.mask_fcn_logits = nn.Conv2d(dim, ncls, 1, 1, 0)
        for name, p in self.named_parameters():
            if "bias" in name:
                nn.init.constant_(p, 0)
            else:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        """x [R,C,h,w] -> class logits [R,ncls,2h,2w]."""
        r, c, h, w = x.shape
        cout, ncls = self.conv5_mask.weight.shape[1], self.mask_fcn_logits.weight.shape[0]
        # the 2x2 / stride-2 transposed convolution has no overlapping taps: a GEMM of the input pixels [R*h*w, Cin] against
        # W [(Cout, dy, dx), Cin] with the bias repeated over (dy, dx) and the ReLU fused (drc_linear_fwd)
        t = gemm_bias_act(x.permute(0, 2, 3, 1).reshape(-1, c), self.conv5_mask.weight, lambda wt: wt.reshape(c, cout * 4).t(),
                          self.conv5_mask.bias.detach().repeat_interleave(4), True, tag="deconv2x2")
        t = t.view(r, h, w, cout, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(-1, cout)             # rows = output pixels (y, dy, x, dx)
        lg = gemm_bias_act(t, self.mask_fcn_logits.weight, lambda wl: wl.view(ncls, cout), self.mask_fcn_logits.bias, False)
        return lg.view(r, 2 * h, 2 * w, ncls).permute(0, 3, 1, 2).contiguous()


class MaskPostProcessor(nn.Module):
    def forward(self, x, boxes):
        prob = x.sigmoid()
        labels = torch.cat([b.get_field("labels") for b in boxes])
        prob = prob[torch.arange(x.shape[0], device=labels.device), labels][:, None]
        out = []
        for p, box in zip(prob.split([len(b) for b in boxes], dim=0), boxes):
            b = BoxList(box.bbox, box.size, mode="xyxy")
            for f in box.fields():
                b.add_field(f, box.get_field(f))
            b.add_field("mask", p)
            out.append(b)
        return out


class ROIMaskHead(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        m = cfg.MODEL.ROI_MASK_HEAD
        if m.FEATURE_EXTRACTOR != "MaskRCNNFPNFeatureExtractor" or m.PREDICTOR != "MaskRCNNC4Predictor" or m.SHARE_BOX_FEATURE_EXTRACTOR:
            raise NotImplementedError("only the mask head of the shipped configs (MaskRCNNFPNFeatureExtractor + MaskRCNNC4Predictor) is built")
        self.feature_extractor = MaskRCNNFPNFeatureExtractor(cfg, in_channels)
        self.predictor = MaskRCNNC4Predictor(cfg, self.feature_extractor.out_channels)
        self.post_processor = MaskPostProcessor()

    def forward(self, features, proposals, targets=None):
        if self.training:
            raise NotImplementedError("mask head training belongs to the 2D stage's training, which is not built")
        x = self.feature_extractor(features, proposals)
        return x, self.post_processor(self.predictor(x), proposals), {}


def build_roi_mask_head(cfg, in_channels):
    return ROIMaskHead(cfg, in_channels)
