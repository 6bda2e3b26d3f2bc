"""BoxList: boxes of one image + per-box fields + image-level maps.  Reference: disprcnn/structures/bounding_box.py:10-455 --
bbox / get_field / add_field / indexing / len (the disparity stage), area, clip_to_image, copy_with_fields and the xyxy<->xywh
view (the 2D stage's heads); transpose / crop / resize of annotations belong to the data pipeline and are not built."""
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

    def area(self):
        """(x2 - x1 + 1) * (y2 - y1 + 1): the legacy +1 pixel convention (reference :373-382)."""
        b = self.bbox
        return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)

    def xywh(self):
        """[R,4] (x, y, w, h) with w = x2 - x1 + 1 (reference convert('xywh'), :96-128)."""
        b = self.bbox
        return torch.stack((b[:, 0], b[:, 1], b[:, 2] - b[:, 0] + 1, b[:, 3] - b[:, 1] + 1), dim=1)

    def clip_to_image(self, remove_empty=True):
        """In place, like the reference (:317-330)."""
        w, h = self.size
        self.bbox[:, 0].clamp_(min=0, max=w - 1); self.bbox[:, 1].clamp_(min=0, max=h - 1)
        self.bbox[:, 2].clamp_(min=0, max=w - 1); self.bbox[:, 3].clamp_(min=0, max=h - 1)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        for f in ([fields] if isinstance(fields, str) else fields):
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError(f"Field '{f}' not found in {self}")
        return out

    def __repr__(self):
        return f"BoxList(num_boxes={len(self)}, image_width={self.size[0]}, image_height={self.size[1]}, mode={self.mode})"
