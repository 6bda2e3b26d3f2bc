"""DispRCNN3D -- the caller of the disparity hot path (reference: disprcnn/modeling/detector/disprcnn3d.py).

Restates the disparity stage of the reference meta-architecture for MI355X:
  remove_illegal_detections (:286-294) -> ROI pairing (:118-146) -> ROIAlign crop of the IMAGE to
  RESOLUTION x RESOLUTION + ImageNet normalisation (:44-50) -> PSMNet (:273) -> 'disparity' field per image (:277-280).
Differences by design: the per-ROI box arithmetic, the crop and the normalisation are single HIP kernels (no
``.tolist()`` host syncs, no Python per-ROI loop); PointRCNN (DET3D_ON) and the mask-based training targets are out of
scope and raise.
"""
from types import SimpleNamespace

import torch
from torch import nn

from ... import _lib
from ... import engine as E
from ...layers.roi_align import roi_align_forward
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
        self.disp_resolution = d.RESOLUTIONS[0]
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
    def prepare_psmnet_input(self, left_images, right_images, left_result, right_result):
        """-> left_roi_images, right_roi_images [R,3,res,res] (normalised), geom [R,4] int32 = (x1, x1p, x2, x2p)."""
        dev = left_images.tensors.device
        E.require_gpu(left_images.tensors, "DispRCNN3D images")
        res = self.disp_resolution
        counts = [len(a) for a in left_result]
        R = sum(counts)
        if R == 0:
            z = torch.empty(0, 3, res, res, device=dev)
            return z, z.clone(), torch.empty(0, 4, dtype=torch.int32, device=dev)
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

    def forward(self, lr_images, lr_result, lr_targets=None):
        left_result, right_result = self.remove_illegal_detections(lr_result["left"], lr_result["right"])
        if self.training:
            raise NotImplementedError("DispRCNN3D training targets (Masker, DisparityMap crop/resize) are not built yet; "
                                      "train iDispNet on ROI crops with PSMNet + PSMLoss (reference tools/kitti_object/train_idispnet_fa.py)")
        return self._forward_eval(lr_images["left"], lr_images["right"], left_result, right_result)
