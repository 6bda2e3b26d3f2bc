"""Minimal BoxList: the three behaviours the disparity stage touches (bbox / get_field / add_field,
plus indexing and len).  Reference: disprcnn/structures/bounding_box.py:10-455 (the rest of that class is 2D-detection
plumbing and out of scope)."""
import torch


class BoxList:
    def __init__(self, bbox, image_size, mode="xyxy"):
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.dim() != 2 or bbox.size(-1) != 4:
            raise ValueError(f"bbox should be [R,4], got {tuple(bbox.shape)}")
        if mode != "xyxy":
            raise ValueError("only mode 'xyxy' is supported on this path")
        self.bbox, self.size, self.mode = bbox, tuple(image_size), mode    # size = (width, height)
        self.extra_fields = {}
        self.PixelWise_map = {}                     # full-image maps ('disparity': DisparityMap), reference bounding_box.py:39,80-94

    @property
    def width(self):
        return self.size[0]

    @property
    def height(self):
        return self.size[1]

    def add_field(self, field, data):
        self.extra_fields[field] = data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def add_map(self, name, data):
        self.PixelWise_map[name] = data

    def get_map(self, name):
        return self.PixelWise_map[name]

    def has_map(self, name):
        return name in self.PixelWise_map

    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item].reshape(-1, 4), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item] if torch.is_tensor(v) else v)
        out.PixelWise_map = dict(self.PixelWise_map)      # image-level maps are not indexed by ROI
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def __repr__(self):
        return f"BoxList(num_boxes={len(self)}, image_width={self.size[0]}, image_height={self.size[1]}, mode={self.mode})"
