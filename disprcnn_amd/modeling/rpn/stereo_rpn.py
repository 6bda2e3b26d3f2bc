"""StereoRPN -- drop-in for ``disprcnn.modeling.rpn.stereo_rpn.srpn.StereoRPN`` (srpn.py:14-137), inference.

state_dict keys as in the reference: ``anchor_generator.cell_anchors.{0..4}``, ``head.conv``, ``head.cls_logits``, ``head.bbox_pred``.
Per FPN level the head is conv3x3(256->512)+ReLU on the left and the right map (one engine launch over both), then two 1x1
convolutions on their channel concatenation: 2A objectness logits and 6A box codes (x, y, w, h of the left box, x' and w' of the
right one).  drc_srpn_proposals_fwd turns the raw maps of a level into per-anchor (score, left box, right box) -- the reference's
pairwise softmax, flattening, decode, left/right split and clip (stereo_rpn/inference.py:121-150, 287-299) in one pass -- and the
selection (descending sort, PRE_NMS_TOP_N, min size, NMS on both views with intersected keeps, POST_NMS_TOP_N; :151-196) follows.
Training (loss_evaluator / box_selector_train) belongs to the 2D stage's training, which is not built."""
import torch
from torch import nn

from ... import _lib
from ... import engine as E
from ...structures.bounding_box import BoxList
from ...structures.boxlist_ops import double_view_boxlist_nms
from ..box_coder import BoxCoder
from ..head_ops import BlockedConv2d, EngineConv2d
from .anchor_generator import make_anchor_generator


class SRPNHead(nn.Module):
    """Parameter holder (srpn.py:14-50); the arithmetic runs in StereoRPN._head."""

    def __init__(self, cfg, in_channels, num_anchors):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels * 2, kernel_size=3, stride=1, padding=1)
        self.cls_logits = nn.Conv2d(in_channels * 4, num_anchors * 2, kernel_size=1, stride=1)
        self.bbox_pred = nn.Conv2d(in_channels * 4, num_anchors * 6, kernel_size=1, stride=1)
        for layer in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)

    def forward(self, left_features, right_features):
        raise RuntimeError("SRPNHead is a parameter holder; call StereoRPN (HIP engine)")


class StereoRPN(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.cfg = cfg
        r = cfg.MODEL.RPN
        self.anchor_generator = make_anchor_generator(cfg)
        self.head = SRPNHead(cfg, in_channels, self.anchor_generator.num_anchors_per_location()[0])
        self.box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.pre_nms_top_n, self.post_nms_top_n = r.PRE_NMS_TOP_N_TEST, r.POST_NMS_TOP_N_TEST
        self.nms_thresh, self.min_size = r.NMS_THRESH, r.MIN_SIZE
        self._conv = EngineConv2d(self.head.conv, relu=True)
        self._cls = EngineConv2d(self.head.cls_logits, relu=False)
        self._reg = EngineConv2d(self.head.bbox_pred, relu=False)
        self._bconv = BlockedConv2d(self.head.conv, relu=True)
        self._bpred = BlockedConv2d([self.head.cls_logits, self.head.bbox_pred], relu=False)      # one read of the 1024-channel map
        self._bws = {}

    # ------------------------------------------------------------------ head: raw maps per level
    def _head(self, left_features, right_features):
        logits, regs = [], []
        for fl, fr in zip(left_features, right_features):
            n = fl.shape[0]
            t = self._conv(torch.cat((fl, fr), 0))                 # left and right maps share the 3x3 convolution: one launch
            t = torch.cat((t[:n], t[n:]), 1)
            logits.append(self._cls(t))
            regs.append(self._reg(t))
        return logits, regs

    def _head_blocked(self, levels):
        """The same maps from the backbone's blocked pyramid (units [left_0..left_n-1, right_0..right_n-1] of each level, halo 1)
        without leaving the blocked layout.  The 3x3 convolution writes image i's left map into channel blocks 0..31 and its right map
        into blocks 32..63 of unit i of one [n][1024-channel] tensor -- the concatenation the predictors read, for free -- and both
        predictors run as one 1x1 convolution (2A + 6A outputs) over it."""
        logits, regs = [], []
        a2, c2 = self.head.cls_logits.out_channels, self.head.conv.out_channels
        for xb in levels:
            n, dev = xb.N // 2, xb.device
            key = (n, xb.H, xb.W, dev)
            ws = self._bws.get(key)
            if ws is None:
                t = E.Blocked(n, 2 * c2, 1, xb.H, xb.W, 0, 0, 0, dev)
                half = (c2 // 16) * t.cb_stride                              # floats of one view's channel blocks within a unit
                if n == 1:                                                   # units (left, right) are already adjacent: one launch
                    pairs = [(xb, E.Blocked(2, c2, 1, xb.H, xb.W, 0, 0, 0, dev, storage=t.storage))]
                else:
                    pairs = []
                    for side in (0, 1):
                        xv = E.Blocked(n, xb.C, 1, xb.H, xb.W, 0, xb.ph, xb.pw, dev, storage=xb.storage[side * n * xb.n_stride:])
                        yv = E.Blocked.geometry(n, c2, 1, xb.H, xb.W, 0, 0, 0, dev)
                        yv.n_stride, yv.storage = t.n_stride, t.storage[side * half:]
                        pairs.append((xv, yv))
                ws = self._bws[key] = dict(t=t, pairs=pairs, o=E.Blocked(n, self._bpred.cout, 1, xb.H, xb.W, 0, 0, 0, dev), src=xb.storage.data_ptr())
                while len(self._bws) > 16:
                    self._bws.pop(next(iter(self._bws)))
            if ws["src"] != xb.storage.data_ptr():                           # a different backbone workspace of the same geometry
                self._bws.pop(key)
                return self._head_blocked(levels)
            for xv, yv in ws["pairs"]:
                self._bconv(xv, yv)
            d = self._bpred(ws["t"], ws["o"]).to_dense()[:, :, 0]
            logits.append(d[:, :a2].contiguous())
            regs.append(d[:, a2:].contiguous())
        return logits, regs

    def proposals_dense(self, left_images, left_features, right_features, blocked_levels=None):
        """-> scores [N,T], left [N,T,4], right [N,T,4] over all T anchors of the pyramid (levels concatenated like the reference)."""
        dev = left_features[0].device
        if blocked_levels is not None and blocked_levels[0].N == 2 * left_features[0].shape[0] and self.head.conv.out_channels % 16 == 0:
            logits, regs = self._head_blocked(blocked_levels)
        else:
            logits, regs = self._head(left_features, right_features)
        anchors = self.anchor_generator.level_anchors([tuple(f.shape[-2:]) for f in left_features], dev)
        n, a = logits[0].shape[0], logits[0].shape[1] // 2
        total = sum(int(x.shape[0]) for x in anchors)
        scores = torch.empty(n, total, dtype=torch.float32, device=dev)
        left = torch.empty(n, total, 4, dtype=torch.float32, device=dev)
        right = torch.empty(n, total, 4, dtype=torch.float32, device=dev)
        wh = torch.tensor([[float(w), float(h)] for (h, w) in left_images.image_sizes], dtype=torch.float32).to(dev)
        off = 0
        for lg, rg, an in zip(logits, regs, anchors):
            h, w = lg.shape[-2:]
            lg, rg = lg.contiguous(), rg.contiguous()              # (named: a temporary would be freed -- and its block reused -- before the launch)
            st = _lib.lib().drc_srpn_proposals_fwd(E._ptr(lg), E._ptr(rg), E._ptr(an), E._ptr(wh), n, a, h, w, total, off,
                                                   self.box_coder.bbox_xform_clip, E._ptr(scores), E._ptr(left), E._ptr(right), E._stream_ptr(dev))
            _lib.check(st, "drc_srpn_proposals_fwd")
            off += h * w * a
        return scores, left, right

    def forward(self, left_images, right_images, left_features, right_features, left_targets=None, right_targets=None,
                blocked_levels=None):
        """blocked_levels: optionally the same pyramid in the engine's layout (BackboneRuntime.blocked_levels(): units [left, right],
        halo 1), which lets the head skip the layout round trips; the results are the same maps."""
        if self.training:
            raise NotImplementedError("Stereo RPN training (losses, proposal sampling) belongs to the 2D stage's training, which is not built")
        E.require_gpu(left_features[0], "StereoRPN")
        scores, left, right = self.proposals_dense(left_images, left_features, right_features, blocked_levels)
        n = scores.shape[0]
        order = torch.sort(scores, 1, True)[1]
        left_result, right_result = [], []
        for i in range(n):
            o = order[i]
            if 0 < self.pre_nms_top_n < scores.numel():            # (numel over the batch: as the reference has it, :166)
                o = o[: self.pre_nms_top_n]
            ih, iw = left_images.image_sizes[i]
            lb = BoxList(left[i].index_select(0, o), (iw, ih), mode="xyxy")
            rb = BoxList(right[i].index_select(0, o), (iw, ih), mode="xyxy")
            s = scores[i].index_select(0, o)
            lb.add_field("objectness", s)
            rb.add_field("objectness", s)
            if self.min_size > 1:                                   # boxes are already clipped, so sides are >= 1
                # the reference filters each view on its own (:185-186), which misaligns the pairs whenever the two views differ;
                # here a pair survives when both of its boxes do (identical for the shipped MIN_SIZE = 0)
                wl, wr = lb.xywh(), rb.xywh()
                ok = ((wl[:, 2] >= self.min_size) & (wl[:, 3] >= self.min_size) & (wr[:, 2] >= self.min_size) & (wr[:, 3] >= self.min_size)).nonzero().squeeze(1)
                lb, rb = lb[ok], rb[ok]
            # (the lists are in descending-score order -- `order` above -- and min_size filtering keeps the order)
            lb, rb = double_view_boxlist_nms(lb, rb, self.nms_thresh, max_proposals=self.post_nms_top_n, score_field="objectness",
                                             scores_sorted=True)
            left_result.append(lb)
            right_result.append(rb)
        return left_result, right_result, {}


def build_stereorpn(cfg, in_channels):
    return StereoRPN(cfg, in_channels)
