"""BoxList: boxes of one image + per-box fields + image-level maps.  Reference: disprcnn/structures/bounding_box.py:10-455 --
bbox / get_field / add_field / indexing / len (the disparity stage), area, clip_to_image, copy_with_fields (the 2D stage's heads), and the
geometry the callers of the path apply to whole annotations: convert (xyxy <-> xywh, legacy +1 widths), resize, transpose, crop
(reference :96-277; used by the transforms in front of the path and by disprcnn3d.py / generate_psmnet_input_inf.py)."""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class BoxList:
    def __init__(self, bbox, image_size, mode="xyxy"):
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.dim() != 2 or bbox.size(-1) != 4:
            raise ValueError(f"bbox should be [R,4], got {tuple(bbox.shape)}")
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
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
        for k, v in self.PixelWise_map.items():
            out.add_map(k, v.to(device) if hasattr(v, "to") else v)
        return out

    # ---- geometry of whole annotations.  Boxes are transformed here; a field / map that is not a tensor and knows the operation
    # (masks, DisparityMap) is asked to follow, tensors pass through untouched (which is why 'disparity' fields stay ROI-normalised).
    def _xyxy(self):
        """[R,4] corners whatever the mode; xywh widths use the legacy +1 pixel convention and are clamped at one pixel."""
        if self.mode == "xyxy":
            return self.bbox
        b = self.bbox
        return torch.cat((b[:, :2], b[:, :2] + (b[:, 2:] - 1).clamp(min=0)), dim=1)

    def _derive(self, corners, size, fields, maps):
        out = BoxList(corners, size, "xyxy")
        out.extra_fields, out.PixelWise_map = dict(fields), dict(maps)
        return out.convert(self.mode)

    def convert(self, mode):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        c = self._xyxy()
        box = c if mode == "xyxy" else torch.cat((c[:, :2], c[:, 2:] - c[:, :2] + 1), dim=1)
        out = BoxList(box, self.size, mode)
        out.extra_fields, out.PixelWise_map = dict(self.extra_fields), dict(self.PixelWise_map)
        return out

    def resize(self, size, *args, **kwargs):
        """-> the annotations of the image resampled to size = (width, height): coordinates times the per-axis ratio (with equal ratios the
        stored box is scaled as it is, whatever its mode, like the reference)."""
        follow = lambda v: v.resize(size, *args, **kwargs) if not torch.is_tensor(v) and hasattr(v, "resize") else v
        fields = {k: follow(v) for k, v in self.extra_fields.items()}
        maps = {k: follow(v) for k, v in self.PixelWise_map.items()}
        rw, rh = (float(n) / float(o) for n, o in zip(size, self.size))
        if rw == rh:
            out = BoxList(self.bbox * rw, size, self.mode)
            out.extra_fields, out.PixelWise_map = fields, maps
            return out
        return self._derive(self._xyxy() * self.bbox.new_tensor([rw, rh, rw, rh]), size, fields, maps)

    def transpose(self, method):
        """FLIP_LEFT_RIGHT: x -> width - x - 1 (corners swapped); FLIP_TOP_BOTTOM: y -> height - y (no -1: the reference's convention).
        Non-tensor fields are flipped with the boxes; image-level maps are handed over as they are, like the reference."""
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        w, h = self.size
        c = self._xyxy()
        if method == FLIP_LEFT_RIGHT:
            flipped = torch.stack((w - c[:, 2] - 1, c[:, 1], w - c[:, 0] - 1, c[:, 3]), dim=1)
        else:
            flipped = torch.stack((c[:, 0], h - c[:, 3], c[:, 2], h - c[:, 1]), dim=1)
        fields = {k: (v if torch.is_tensor(v) else v.transpose(method)) for k, v in self.extra_fields.items()}
        return self._derive(flipped, self.size, fields, self.PixelWise_map)

    def crop(self, box, crop_map=False):
        """-> the annotations inside the window box = (left, upper, right, lower): corners shifted and clamped into the window (empty boxes are
        kept), image size = the window's.  Non-tensor fields are cropped; image-level maps only with crop_map (else handed over)."""
        w, h = box[2] - box[0], box[3] - box[1]
        c = self._xyxy() - self.bbox.new_tensor([box[0], box[1], box[0], box[1]])
        c = torch.stack((c[:, 0].clamp(min=0, max=w), c[:, 1].clamp(min=0, max=h), c[:, 2].clamp(min=0, max=w), c[:, 3].clamp(min=0, max=h)), dim=1)
        fields = {k: (v if torch.is_tensor(v) else v.crop(box)) for k, v in self.extra_fields.items()}
        maps = {k: (v.crop(box) if crop_map and not torch.is_tensor(v) else v) for k, v in self.PixelWise_map.items()}
        return self._derive(c, (w, h), fields, maps)

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
        if self.mode == "xywh":
            return self.bbox[:, 2] * self.bbox[:, 3]
        b = self.bbox
        return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)

    def xywh(self):
        """[R,4] (x, y, w, h) with w = x2 - x1 + 1 (reference convert('xywh'), :96-128)."""
        return self.convert("xywh").bbox

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
