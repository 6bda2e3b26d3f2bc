"""ImageList: batched image tensor + per-image sizes (reference: disprcnn/structures/image_list.py:7-102)."""
import torch


class ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, list(image_sizes)     # sizes are (h, w) as in the reference

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    """[B,C,H,W] tensor, [C,H,W] tensor or list of [C,H,W] tensors -> ImageList (zero padded to a common size)."""
    if isinstance(tensors, ImageList):
        return tensors
    if torch.is_tensor(tensors):
        if tensors.dim() == 3:
            tensors = tensors[None]
        return ImageList(tensors, [tuple(t.shape[-2:]) for t in tensors])
    mh = max(t.shape[-2] for t in tensors)
    mw = max(t.shape[-1] for t in tensors)
    if size_divisible > 0:
        mh = -(-mh // size_divisible) * size_divisible
        mw = -(-mw // size_divisible) * size_divisible
    out = tensors[0].new_zeros(len(tensors), tensors[0].shape[0], mh, mw)
    for t, o in zip(tensors, out):
        o[:, : t.shape[-2], : t.shape[-1]].copy_(t)
    return ImageList(out, [tuple(t.shape[-2:]) for t in tensors])
