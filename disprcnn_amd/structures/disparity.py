"""DisparityMap: the value holder the post-processing hands on and the training-target code crops and resamples
(reference: disprcnn/structures/disparity.py:12-83; callers modeling/detector/disprcnn3d.py:89-91,173-174 and
tools/kitti_object/generate_psmnet_input_inf.py:99-104).  `resize` of a GPU map is the HIP kernel
(drc_disparity_resize_fwd, csrc/post_ops.hip); the batched forms of the same arithmetic live in drc_roi_train_targets_fwd /
drc_disparity_paste_fwd.  A CPU map (the data pipeline: the reference attaches CPU DisparityMaps to its targets in data/datasets/kitti_*.py
and resamples them in dataloader workers, data/transforms) is resampled with the same arithmetic in torch -- that is the data side in
front of the path, not a fallback of the device path: a GPU tensor never takes it."""
import warnings

import torch


class DisparityMap:
    def __init__(self, data):
        self.data = torch.as_tensor(data).float()
        if self.data.dim() != 2:
            raise ValueError(f"a disparity map is [H,W], got {tuple(self.data.shape)}")

    def clone(self):
        return DisparityMap(self.data.clone())

    @property
    def size(self):
        return self.data.shape[::-1]

    @property
    def width(self):
        return self.data.shape[1]

    @property
    def height(self):
        return self.data.shape[0]

    def to(self, device):
        return DisparityMap(self.data.to(device))

    def resize(self, dst_size, use_max_pooling=False):
        """-> the map resampled to dst_size = (width, height), values scaled by dst_width / width (a disparity is a horizontal pixel
        distance).  Bilinear with align_corners=True, or (use_max_pooling) the signed max pooling of the reference; a negative
        size leaves the map unchanged with a warning, like the reference.  GPU maps: HIP kernel; CPU maps (dataloader side): torch."""
        from .. import _lib
        from .. import engine as E
        if any(s < 0 for s in dst_size):
            warnings.warn("dst size < 0, size will not change.")
            return self.clone()
        ow, oh = (int(round(s)) for s in dst_size)
        if not self.data.is_cuda:
            return DisparityMap(self._resize_cpu(oh, ow, use_max_pooling))
        E.require_gpu(self.data, "DisparityMap.resize")
        src = self.data.contiguous()
        out = torch.empty(oh, ow, dtype=torch.float32, device=src.device)
        st = _lib.lib().drc_disparity_resize_fwd(E._ptr(src), self.height, self.width, E._ptr(out), oh, ow, int(bool(use_max_pooling)),
                                                 E._stream_ptr(src.device))
        _lib.check(st, "drc_disparity_resize_fwd")
        return DisparityMap(out)

    def _resize_cpu(self, oh, ow, use_max_pooling):
        """The arithmetic of drc_disparity_resize_fwd on a CPU tensor (reference disparity.py:39-62): bilinear with align_corners=True, or
        max over the positive part minus max over the negative part per adaptive cell; values times ow / width."""
        import torch.nn.functional as F
        x = self.data[None, None]
        if use_max_pooling:
            pos = F.adaptive_max_pool2d(x.clamp_min(0), (oh, ow))
            neg = F.adaptive_max_pool2d((-x).clamp_min(0), (oh, ow))
            y = pos - neg
        else:
            y = F.interpolate(x, (oh, ow), mode="bilinear", align_corners=True)
        return y[0, 0] / self.width * ow

    def crop(self, box):
        """box = (left, upper, right, lower), rounded to integers -> the [lower-upper, right-left] window; where the box reaches past the
        bottom / right edge the window is zero-filled (reference :68-77).  Coordinates are expected non-negative (callers clamp first)."""
        x1, y1, x2, y2 = (int(round(float(v))) for v in box)
        if min(x1, y1) < 0 or x2 < x1 or y2 < y1:
            raise ValueError(f"DisparityMap.crop: box {tuple(box)} is not a forward window of non-negative coordinates")
        out = self.data.new_zeros((y2 - y1, x2 - x1))
        part = self.data[y1:y2, x1:x2]
        out[: part.shape[0], : part.shape[1]] = part
        return DisparityMap(out)

    def __sub__(self, other):
        return DisparityMap(self.data - other)
