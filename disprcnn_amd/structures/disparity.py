"""DisparityMap: the value holder the post-processing hands on (reference: disprcnn/structures/disparity.py:12-36; the
resize / crop arithmetic of that class runs inside drc_disparity_paste_fwd on this path)."""
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
