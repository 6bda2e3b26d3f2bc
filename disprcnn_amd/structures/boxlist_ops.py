"""BoxList operations of the 2D stage.  Reference: disprcnn/structures/boxlist_ops.py:11-96,178-217 -- boxlist_nms,
double_view_boxlist_nms (left and right views suppressed separately, the kept sets intersected), remove_small_boxes, cat_boxlist.
NMS itself runs in libdisprcnn_hip.so (layers/nms.py); index bookkeeping is torch plumbing."""
import torch

from ..layers import nms as _box_nms, nms_pair as _box_nms_pair, nms_pair_sorted_joint as _box_nms_joint
from .bounding_box import BoxList


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    if nms_thresh <= 0:
        return boxlist
    keep = _box_nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep]


def intersect_sorted(a, b):
    """Ascending int64 index sets (as nms returns them) -> their ascending intersection (reference intersect_pytorch, :36-46)."""
    aux = torch.cat((torch.unique(a), torch.unique(b))).sort().values
    return aux[:-1][aux[1:] == aux[:-1]]


def double_view_boxlist_nms(left_boxlist, right_boxlist, nms_thresh, max_proposals=-1, score_field="scores", use_keep="joint",
                            scores_sorted=False):
    """scores_sorted=True (an addition to the reference signature): the caller guarantees descending scores -- the Stereo RPN's lists --
    so the joint keep needs no sort and the walk stops at max_proposals (same result)."""
    if use_keep not in ("joint", "left", "right"):
        raise ValueError(use_keep)
    if nms_thresh <= 0:
        return left_boxlist, right_boxlist
    if use_keep == "joint" and scores_sorted:
        keep = _box_nms_joint(left_boxlist.bbox, right_boxlist.bbox, nms_thresh, max_proposals)
        return left_boxlist[keep], right_boxlist[keep]
    if use_keep == "joint":       # both views share their scores, hence their order: one sort and one launch pair for the two
        keep = _box_nms_pair(left_boxlist.bbox, right_boxlist.bbox, left_boxlist.get_field(score_field), nms_thresh, joint=True)
    elif use_keep == "left":
        keep = _box_nms(left_boxlist.bbox, left_boxlist.get_field(score_field), nms_thresh)
    else:
        keep = _box_nms(right_boxlist.bbox, right_boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return left_boxlist[keep], right_boxlist[keep]


def remove_small_boxes(boxlist, min_size):
    wh = boxlist.xywh()
    return boxlist[((wh[:, 2] >= min_size) & (wh[:, 3] >= min_size)).nonzero().squeeze(1)]


def cat_boxlist(bboxes):
    """Concatenate BoxLists of one image (same size, mode and fields)."""
    bboxes = list(bboxes)
    if not bboxes:
        raise ValueError("cat_boxlist needs at least one BoxList")
    size, mode, fields = bboxes[0].size, bboxes[0].mode, set(bboxes[0].fields())
    if not all(b.size == size and b.mode == mode and set(b.fields()) == fields for b in bboxes):
        raise ValueError("cat_boxlist: BoxLists differ in size, mode or fields")
    out = BoxList(torch.cat([b.bbox for b in bboxes], 0), size, mode)
    for f in bboxes[0].fields():
        out.add_field(f, torch.cat([b.get_field(f) for b in bboxes], 0))
    return out
