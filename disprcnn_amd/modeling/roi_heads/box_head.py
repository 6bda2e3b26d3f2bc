"""Stereo ROI box head -- drop-in for ``disprcnn.modeling.roi_heads.box_head`` in its shipped configuration (inference):
ROIBoxHead.forward_double_view (box_head.py:67-121) = StereoFPN2MLPFeatureExtractor (roi_box_feature_extractors.py:85-120) +
StereoFPNPredictor (roi_box_predictors.py:61-83) + PostProcessor.forward_double_view (inference.py:84-120, 213-263).

state_dict keys as in the reference (``feature_extractor.RCNN_top.{0,3}``, ``predictor.{cls_score,bbox_pred}``).  Data path: 7x7
ROIAlign of the left and the right pyramid (HIP), channel concat, then RCNN_top -- a 7x7/stride-7 convolution on a 7x7 map and a 1x1
convolution on a 1x1 map, i.e. two fully connected layers (25088 -> 2048 -> 2048; Dropout is the identity in eval) -- and the two
predictors: plain library GEMMs.  Post-processing: softmax, per-class box decode + clip in one HIP launch (drc_box_decode_fwd),
score threshold, NMS on the LEFT view (use_keep='left'), detections_per_img by k-th value."""
import torch
from torch import nn

from ...structures.bounding_box import BoxList
from ...structures.boxlist_ops import cat_boxlist, double_view_boxlist_nms
from ..box_coder import BoxCoder
from ..head_ops import linear
from ..poolers import Pooler


class StereoFPN2MLPFeatureExtractor(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        b = cfg.MODEL.ROI_BOX_HEAD
        self.pooler = Pooler((b.POOLER_RESOLUTION, b.POOLER_RESOLUTION), b.POOLER_SCALES, b.POOLER_SAMPLING_RATIO)
        rep = b.MLP_HEAD_DIM
        self.RCNN_top = nn.Sequential(nn.Conv2d(2 * in_channels, rep, kernel_size=b.POOLER_RESOLUTION, stride=b.POOLER_RESOLUTION, padding=0),
                                      nn.ReLU(True), nn.Dropout(p=0.2), nn.Conv2d(rep, rep, kernel_size=1, stride=1, padding=0), nn.ReLU(True),
                                      nn.Dropout(p=0.2))
        self.out_channels = rep

    def forward(self, x, proposals):
        if self.training:
            raise NotImplementedError("box head training is not built")
        lx = self.pooler(x["left"], proposals["left"])
        rx = self.pooler(x["right"], proposals["right"])
        t = torch.cat([lx, rx], dim=1).flatten(1)                  # [R, 2C*7*7]: the 7x7/7 convolution sees the whole window
        t = linear(t, self.RCNN_top[0], relu=True)
        return linear(t, self.RCNN_top[3], relu=True)               # the mean over the 1x1 map is the identity


class StereoFPNPredictor(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        ncls = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        self.cls_score = nn.Linear(in_channels, ncls)
        self.bbox_pred = nn.Linear(in_channels, (2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else ncls) * 6)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for layer in (self.cls_score, self.bbox_pred):
            nn.init.constant_(layer.bias, 0)

    def forward(self, x):
        x = x.flatten(1)
        return linear(x, self.cls_score), linear(x, self.bbox_pred)


class PostProcessor(nn.Module):
    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None, cls_agnostic_bbox_reg=False):
        super().__init__()
        self.score_thresh, self.nms, self.detections_per_img = score_thresh, nms, detections_per_img
        self.box_coder = box_coder or BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
        if cls_agnostic_bbox_reg:
            raise NotImplementedError("CLS_AGNOSTIC_BBOX_REG is off in the shipped configs")

    def forward(self, x, boxes):
        left_boxes, right_boxes = boxes["left"], boxes["right"]
        class_logits, box_regression = x
        prob = torch.softmax(class_logits, -1)
        ncls = prob.shape[1]
        if box_regression.shape[1] != 6 * ncls:
            raise ValueError("stereo box regression carries 6 codes per class")
        counts = [len(b) for b in left_boxes]
        # codes (x, y, w, h, x', w') per class -> left takes (0,1,2,3), right (4,1,5,3): decoded per image so that the clip is fused
        reg = box_regression.view(-1, ncls, 6)
        dl = reg[:, :, [0, 1, 2, 3]].reshape(-1, 4 * ncls)
        dr = reg[:, :, [4, 1, 5, 3]].reshape(-1, 4 * ncls)
        left_results, right_results = [], []
        for p, l_codes, r_codes, lb, rb in zip(prob.split(counts), dl.split(counts), dr.split(counts), left_boxes, right_boxes):
            lp = self.box_coder.decode(l_codes, lb.bbox, clip_to=lb.size, per=4)
            rp = self.box_coder.decode(r_codes, rb.bbox, clip_to=rb.size, per=4)
            lr, rr = self._filter(lp, rp, p, lb.size, ncls)
            left_results.append(lr)
            right_results.append(rr)
        return left_results, right_results

    def _filter(self, lp, rp, scores, size, ncls):
        dev = scores.device
        lres, rres = [], []
        inds_all = scores > self.score_thresh
        for j in range(1, ncls):
            inds = inds_all[:, j].nonzero().squeeze(1)
            s = scores[inds, j]
            lj = BoxList(lp[inds, 4 * j: 4 * (j + 1)], size, mode="xyxy")
            rj = BoxList(rp[inds, 4 * j: 4 * (j + 1)], size, mode="xyxy")
            lj.add_field("scores", s)
            rj.add_field("scores", s)
            lj, rj = double_view_boxlist_nms(lj, rj, self.nms, use_keep="left")
            lab = torch.full((len(lj),), j, dtype=torch.int64, device=dev)
            lj.add_field("labels", lab)
            rj.add_field("labels", lab)
            lres.append(lj)
            rres.append(rj)
        lres, rres = cat_boxlist(lres), cat_boxlist(rres)
        nd = len(lres)
        if nd > self.detections_per_img > 0:
            sc = lres.get_field("scores")
            thr = torch.kthvalue(sc, nd - self.detections_per_img + 1)[0]          # on the device (the reference goes through .cpu())
            keep = torch.nonzero(sc >= thr).squeeze(1)
            lres, rres = lres[keep], rres[keep]
        return lres, rres


class ROIBoxHead(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        b, h = cfg.MODEL.ROI_BOX_HEAD, cfg.MODEL.ROI_HEADS
        if b.FEATURE_EXTRACTOR != "StereoFPN2MLPFeatureExtractor" or b.PREDICTOR != "StereoFPNPredictor":
            raise NotImplementedError("only the stereo box head of the shipped configs (StereoFPN2MLPFeatureExtractor + StereoFPNPredictor) is built")
        self.feature_extractor = StereoFPN2MLPFeatureExtractor(cfg, in_channels)
        self.predictor = StereoFPNPredictor(cfg, self.feature_extractor.out_channels)
        self.post_processor = PostProcessor(h.SCORE_THRESH, h.NMS, h.DETECTIONS_PER_IMG, BoxCoder(weights=h.BBOX_REG_WEIGHTS),
                                            cfg.MODEL.CLS_AGNOSTIC_BBOX_REG)

    def forward(self, features, proposals, targets=None):
        if self.training:
            raise NotImplementedError("box head training (sampling, losses) belongs to the 2D stage's training, which is not built")
        x = self.feature_extractor(features, proposals)
        class_logits, box_regression = self.predictor(x)
        left, right = self.post_processor((class_logits, box_regression), proposals)
        return x, left, right, {}


def build_roi_box_head(cfg, in_channels):
    return ROIBoxHead(cfg, in_channels)
