"""Post-processing of the disparity stage: per-ROI disparities -> one full-image DisparityMap per image.

Same interface as the reference's DisparityMapProcessor (disprcnn/modeling/psmnet/inference.py:18-60): called with the left and
right predictions (BoxList or lists of them; the left ones carry the field 'disparity' [R,S,S]) and returning a DisparityMap
or a list.  All images of a call with the same size go through ONE launch of drc_disparity_paste_fwd; the boxes never leave
the device.  GPU only (no CPU fallback)."""
import torch

from ... import ops
from ...structures.bounding_box import BoxList
from ...structures.disparity import DisparityMap


def sanity_check(left_predictions, right_predictions):
    """The preconditions the reference asserts (inference.py:10-15): paired lists of BoxLists, image by image the same number
    of ROIs and the same image size."""
    if len(left_predictions) != len(right_predictions):
        raise AssertionError(f"{len(left_predictions)} left vs {len(right_predictions)} right predictions")
    for k, (lp, rp) in enumerate(zip(left_predictions, right_predictions)):
        if not (isinstance(lp, BoxList) and isinstance(rp, BoxList)):
            raise AssertionError(f"image {k}: predictions must be BoxLists")
        if len(lp) != len(rp) or lp.size != rp.size:
            raise AssertionError(f"image {k}: left {len(lp)} ROIs on {lp.size}, right {len(rp)} ROIs on {rp.size}")


class DisparityMapProcessor:
    def __call__(self, left_predictions, right_predictions):
        single = isinstance(left_predictions, BoxList) and isinstance(right_predictions, BoxList)
        if single:
            left_predictions, right_predictions = [left_predictions], [right_predictions]
        sanity_check(left_predictions, right_predictions)
        results = [None] * len(left_predictions)
        by_size = {}
        for i, l in enumerate(left_predictions):
            by_size.setdefault(l.size, []).append(i)
        for (width, height), idx in by_size.items():
            disps = [left_predictions[i].get_field("disparity") for i in idx]
            for i, d in zip(idx, disps):
                assert len(d) == len(left_predictions[i]), f"{len(d), len(left_predictions[i])}"
            dev = next((d.device for d in disps if d.numel()), None)
            if dev is None:                       # no ROI in any of these images
                for i in idx:
                    results[i] = DisparityMap(torch.zeros((height, width)))
                continue
            S = next(d.shape[-1] for d in disps if d.numel())
            disp = torch.cat([d.to(dev).reshape(-1, S, S) for d in disps])
            lb = torch.cat([left_predictions[i].bbox.to(dev) for i in idx])
            rb = torch.cat([right_predictions[i].bbox.to(dev) for i in idx])
            maps = ops.disparity_paste(disp, ops.integer_roi_boxes(lb, rb), [len(left_predictions[i]) for i in idx], height, width)
            for k, i in enumerate(idx):
                results[i] = DisparityMap(maps[k])
        return results[0] if len(results) == 1 else results
