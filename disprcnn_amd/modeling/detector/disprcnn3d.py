"""DispRCNN3D -- the caller of the disparity hot path (reference: disprcnn/modeling/detector/disprcnn3d.py).

Restates the disparity stage of the reference meta-architecture for MI355X:
  remove_illegal_detections (:286-294) -> ROI pairing (:118-146) -> ROIAlign crop of the IMAGE to
  RESOLUTION x RESOLUTION + ImageNet normalisation (:44-50) -> PSMNet (:273) -> 'disparity' field per image (:277-280).
Training entry (:209-264): remove_low_score_rois (:192-207) -> crops + targets (prepare_psmnet_input_and_target with
require_mask_tgts, :52-112: Masker paste & ground-truth mask, DisparityMap crop / offset / resize) -> MAX_ROI_FOR_TRAINING
truncation (:223-243) -> PSMNet (train mode, three heads) -> EndPointErrorLoss -> {'disp_loss'}.
Differences by design: the per-ROI box arithmetic, the crop + normalisation and the whole target preparation are single HIP
kernels (no ``.tolist()`` host syncs, no Python per-ROI loop, no full-size per-ROI mask images); an empty ROI set yields a
zero loss that still reaches every parameter, so data-parallel ranks never skip a step.  PointRCNN (DET3D_ON) is
downstream of the hot path and raises.
"""
from types import SimpleNamespace

import torch
from torch import nn

from ... import _lib
from ... import engine as E
from ...layers.roi_align import roi_align_forward
from ...utils.loss_utils import EndPointErrorLoss
from ..psmnet.stackhourglass import PSMNet

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def default_cfg(max_disp=48, min_disp=-48, resolution=224):
    """The handful of yacs keys this stage reads (reference config/defaults.py:530-549), as a plain namespace."""
    return SimpleNamespace(MODEL=SimpleNamespace(
        META_ARCHITECTURE="DispRCNN3D", DISPNET_ON=True, DET3D_ON=False, DEVICE="cuda",
        DISPNET=SimpleNamespace(MAX_DISP=max_disp, MIN_DISP=min_disp, RESOLUTIONS=(resolution,), TRAINED_MODEL="",
                                ROI_MIN_SCORE=0.0, MAX_ROI_FOR_TRAINING=12)),
        SOLVER=SimpleNamespace(TRAIN_PSM=True, TRAIN_PC=False))


class DispRCNN3D(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if getattr(cfg.MODEL, "DET3D_ON", False):
            raise NotImplementedError("PointRCNN (MODEL.DET3D_ON) is downstream of the hot path and not built (SURVEY 2)")
        d = cfg.MODEL.DISPNET
        self.dispnet = PSMNet(d.MAX_DISP, d.MIN_DISP)
        self.dispnet_lossfn = EndPointErrorLoss()
        self.disp_resolution = d.RESOLUTIONS[0]
        self.mask_threshold, self.mask_padding = 0.7, 1          # Masker(0.7, 1), reference :27
        if getattr(d, "TRAINED_MODEL", ""):
            self.dispnet.load_state_dict(torch.load(d.TRAINED_MODEL, "cpu")["model"])     # reference :29-32
        self.register_buffer("_mean", torch.tensor(_MEAN, dtype=torch.float32), persistent=False)
        self.register_buffer("_std", torch.tensor(_STD, dtype=torch.float32), persistent=False)

    # ------------------------------------------------------------------ reference :286-294
    @staticmethod
    def remove_illegal_detections(left_result, right_result):
        lrs, rrs = [], []
        for lr, rr in zip(left_result, right_result):
            lk = (lr.bbox[:, 2] > lr.bbox[:, 0] + 1) & (lr.bbox[:, 3] > lr.bbox[:, 1] + 1)
            rk = (rr.bbox[:, 2] > rr.bbox[:, 0] + 1) & (rr.bbox[:, 3] > rr.bbox[:, 1] + 1)
            keep = lk & rk
            lrs.append(lr[keep]); rrs.append(rr[keep])
        return lrs, rrs

    # ------------------------------------------------------------------ reference :113-159 (require_mask_tgts=False)
    def _crops(self, left_images, right_images, left_result, right_result):
        """Pairing + crops: -> left, right [R,3,res,res] (normalised), geom [R,4] int32, rois_left [R,5] f32, left boxes [R,4] f32."""
        dev = left_images.tensors.device
        E.require_gpu(left_images.tensors, "DispRCNN3D images")
        res = self.disp_resolution
        counts = [len(a) for a in left_result]
        R = sum(counts)
        if R == 0:
            z = torch.empty(0, 3, res, res, device=dev)
            return z, z.clone(), torch.empty(0, 4, dtype=torch.int32, device=dev), torch.empty(0, 5, device=dev), torch.empty(0, 4, device=dev)
        lb = torch.cat([a.bbox for a in left_result]).to(dev).float().contiguous()
        rb = torch.cat([a.bbox for a in right_result]).to(dev).float().contiguous()
        idx = torch.cat([torch.full((c,), i, dtype=torch.int32) for i, c in enumerate(counts)]).to(dev)
        sizes = {(a.width, a.height) for a in left_result}
        if len(sizes) != 1:
            raise ValueError("all images of a batch must share one size on this path")   # KITTI batches do
        w, h = sizes.pop()
        rois_l = torch.empty(R, 5, device=dev); rois_r = torch.empty(R, 5, device=dev)
        geom = torch.empty(R, 4, dtype=torch.int32, device=dev)
        st = _lib.lib().drc_align_roi_pairs(E._ptr(lb), E._ptr(rb), E._ptr(idx), R, int(w), int(h), E._ptr(rois_l), E._ptr(rois_r),
                                            E._ptr(geom), E._stream_ptr(dev))
        _lib.check(st, "drc_align_roi_pairs")
        left = roi_align_forward(left_images.tensors, rois_l, 1.0, res, res, 0, self._mean, self._std)
        right = roi_align_forward(right_images.tensors, rois_r, 1.0, res, res, 0, self._mean, self._std)
        return left, right, geom, rois_l, lb

    def prepare_psmnet_input(self, left_images, right_images, left_result, right_result):
        """-> left_roi_images, right_roi_images [R,3,res,res] (normalised), geom [R,4] int32 = (x1, x1p, x2, x2p)."""
        left, right, geom, _, _ = self._crops(left_images, right_images, left_result, right_result)
        return left, right, geom

    # ------------------------------------------------------------------ reference :266-284
    def _forward_eval(self, left_images, right_images, left_result, right_result):
        left, right, geom = self.prepare_psmnet_input(left_images, right_images, left_result, right_result)
        if left.shape[0] > 0:
            output = self.dispnet((left, right))
        else:
            output = torch.zeros(0, self.disp_resolution, self.disp_resolution, device=left.device)
        counts = [len(a) for a in left_result]
        for lr, o, gm in zip(left_result, torch.split(output, counts), torch.split(geom, counts)):
            lr.add_field("disparity", o)             # ROI-normalised pixel units, as the reference
            lr.add_field("roi_geom", gm)             # (x1, x1p, x2, x2p): offset x1-x1p and scale (x2-x1)/res for consumers
        return {"left": left_result, "right": right_result}

    # ------------------------------------------------------------------ reference :192-207
    def remove_low_score_rois(self, left_result, right_result):
        thresh = self.cfg.MODEL.DISPNET.ROI_MIN_SCORE
        counts = [len(a) for a in left_result]
        scores = torch.cat([a.get_field("scores") for a in left_result]) if left_result else torch.zeros(0)
        keep = scores > thresh
        n = int(keep.sum())
        if 1 < n < 2:                                   # (sic) the reference's "keep at least 2" branch can never trigger
            idxs = scores.argsort(descending=True)
            keep[idxs[0]] = keep[idxs[1]] = True
        elif n == 1:
            keep.fill_(True)
        ret_lr, ret_rr = [], []
        for lr, rr, k in zip(left_result, right_result, torch.split(keep, counts)):
            ret_lr.append(lr[k]); ret_rr.append(rr[k])
        return ret_lr, ret_rr

    # ------------------------------------------------------------------ reference :52-112 (require_mask_tgts=True)
    def prepare_psmnet_input_and_target(self, left_images, right_images, left_result, right_result, left_targets):
        """-> left_roi_images, right_roi_images [R,3,res,res], roi_disp_targets [R,res,res] f32, roi_masks [R,res,res] u8."""
        left, right, geom, rois_l, lb = self._crops(left_images, right_images, left_result, right_result)
        dev, res = left.device, self.disp_resolution
        R = left.shape[0]
        targets = torch.empty(R, res, res, dtype=torch.float32, device=dev)
        masks = torch.empty(R, res, res, dtype=torch.uint8, device=dev)
        if R == 0:
            return left, right, targets, masks
        w, h = left_result[0].width, left_result[0].height
        disp_maps, gt_masks = [], []
        for t in left_targets:
            dm = t.get_map("disparity")
            disp_maps.append(torch.as_tensor(getattr(dm, "data", dm), dtype=torch.float32))
            gm = t.get_field("masks")
            gm = gm.get_full_image_mask_tensor() if hasattr(gm, "get_full_image_mask_tensor") else torch.as_tensor(gm)
            if gm.dim() == 3:                           # instance stack -> union (SegmentationMask.get_full_image_mask_tensor)
                gm = gm.sum(dim=0).clamp(max=1) if gm.shape[0] else gm.new_zeros(gm.shape[1:])
            gt_masks.append(gm.to(torch.uint8))
        disp_maps = torch.stack(disp_maps).to(dev).contiguous()
        gt_masks = torch.stack(gt_masks).to(dev).contiguous()
        if tuple(disp_maps.shape[1:]) != (h, w) or tuple(gt_masks.shape[1:]) != (h, w):
            raise ValueError("ground-truth disparity maps / masks must have the image size of the detections")
        M = next(int(a.get_field("mask").shape[-1]) for a in left_result if len(a))
        probs = torch.cat([a.get_field("mask").reshape(len(a), M * M) for a in left_result]).to(dev).float().contiguous()
        st = _lib.lib().drc_roi_train_targets_fwd(E._ptr(disp_maps), E._ptr(gt_masks), E._ptr(probs), M, self.mask_padding, self.mask_threshold,
                                                  E._ptr(lb), E._ptr(rois_l), E._ptr(geom), R, int(h), int(w), res, E._ptr(targets), E._ptr(masks),
                                                  E._stream_ptr(dev))
        _lib.check(st, "drc_roi_train_targets_fwd")
        return left, right, targets, masks

    @staticmethod
    def _truncate(results, max_rois):
        """Keep the first max_rois ROIs of the batch (reference :229-243)."""
        out, s = [], 0
        for r in results:
            k = 0 if s >= max_rois else min(max_rois - s, len(r))
            out.append(r[torch.arange(k)])
            s += k
        return out

    # ------------------------------------------------------------------ reference :209-264
    def _forward_train(self, left_images, right_images, left_result, right_result, left_targets):
        losses = {}
        left_result, right_result = self.remove_low_score_rois(left_result, right_result)
        left, right, targets, masks = self.prepare_psmnet_input_and_target(left_images, right_images, left_result, right_result, left_targets)
        max_rois = self.cfg.MODEL.DISPNET.MAX_ROI_FOR_TRAINING
        if self.dispnet.training and left.shape[0] > max_rois:
            left, right, targets, masks = left[:max_rois], right[:max_rois], targets[:max_rois], masks[:max_rois]
            left_result, right_result = self._truncate(left_result, max_rois), self._truncate(right_result, max_rois)
        if left.shape[0] > 0:
            output = self.dispnet((left.contiguous(), right.contiguous()))
            disp_loss = self.dispnet_lossfn(targets, output, masks)
        else:
            # no ROI on this rank: a zero loss that still touches every parameter, so the data-parallel gradient exchange of
            # this step is entered by every rank (the reference would crash or, under DDP, hang here -- SURVEY 5)
            res = self.disp_resolution
            output = tuple(torch.zeros(0, res, res, device=left.device) for _ in range(3)) if self.dispnet.training else \
                torch.zeros(0, res, res, device=left.device)
            disp_loss = sum(p.sum() for p in self.dispnet.parameters() if p.requires_grad) * 0.0
        if self.cfg.SOLVER.TRAIN_PSM:
            losses.update(disp_loss=disp_loss)
        out3 = output[2] if isinstance(output, (list, tuple)) else output
        for o3, lr in zip(torch.split(out3.detach(), [len(r) for r in left_result]), left_result):
            lr.add_field("disparity", o3)
        return losses

    def forward(self, lr_images, lr_result, lr_targets=None):
        left_result, right_result = self.remove_illegal_detections(lr_result["left"], lr_result["right"])
        if self.training:
            if lr_targets is None:
                raise ValueError("DispRCNN3D in training mode needs lr_targets (reference disprcnn3d.py:303)")
            return self._forward_train(lr_images["left"], lr_images["right"], left_result, right_result, lr_targets["left"])
        return self._forward_eval(lr_images["left"], lr_images["right"], left_result, right_result)

    def load_state_dict(self, state_dict, strict=True):
        """Reference :310-316: after loading a whole-detector checkpoint the disparity net is re-loaded from
        MODEL.DISPNET.TRAINED_MODEL when that is set (the iDispNet weights trained stand-alone win)."""
        ret = super().load_state_dict(state_dict, strict)
        tm = getattr(self.cfg.MODEL.DISPNET, "TRAINED_MODEL", "")
        if tm:
            self.dispnet.load_state_dict(torch.load(tm, "cpu")["model"])
        return ret
