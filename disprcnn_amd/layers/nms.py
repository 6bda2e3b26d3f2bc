"""nms -- drop-in for ``disprcnn.layers.nms`` (= ``disprcnn._C.nms``, reference layers/nms.py:8, csrc/nms.h:12-28).

``nms(dets[N,4] xyxy, scores[N], threshold) -> int64 indices of the kept boxes, ascending`` (the CUDA op sorts the kept original
indices, csrc/cuda/nms.cu:126-130; the CPU op returns nonzero(), csrc/cpu/nms_cpu.cpp:64).  GPU only: the suppression bitmask
and the greedy walk run in libdisprcnn_hip.so (drc_nms_sorted_fwd); torch supplies the score sort and the final compaction
(whose data-dependent output length is the one host sync, as in the reference).  Any n up to 524,288 boxes: up to 32,768 the walk keeps its
removed-words in registers, beyond that in LDS; the suppression mask takes n * ceil(n / 64) * 8 bytes, like the reference's."""
import torch

from .. import _lib
from .. import engine as E


_MASK_CHECK_BYTES = 256 << 20      # masks above this size (~46k boxes per view) are checked against the free device memory first


def _mask_workspace(views, n, device):
    """The dense suppression mask: views * n * ceil(n / 64) words of 8 bytes (1.8 GB per view at 120k boxes, 34 GB at the 524,288-box limit).
    Checked against the free device memory first, so that an oversized call fails with the reason instead of an allocator error."""
    words = views * n * ((n + 63) // 64)
    need = words * 8
    if need <= _MASK_CHECK_BYTES:           # the per-image RPN / box-head calls (a few thousand boxes): no driver query on the hot path
        return torch.empty(words, dtype=torch.int64, device=device)
    free, _total = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    if need > free + cached:
        raise RuntimeError(f"nms: the suppression mask of {n} boxes x {views} view(s) needs {need / 2**30:.1f} GiB "
                           f"(n * ceil(n/64) * 8 bytes per view), {(free + cached) / 2**30:.1f} GiB are free; pre-select the top-scoring boxes "
                           f"(the reference keeps PRE_NMS_TOP_N <= 12,000 per level)")
    return torch.empty(words, dtype=torch.int64, device=device)


def nms(dets, scores, threshold, strict=True):
    """strict=True: suppress when IoU > threshold (the reference's CUDA op); False: >= (its CPU op)."""
    E.require_gpu(dets, "nms")
    if dets.dim() != 2 or dets.shape[1] != 4 or scores.shape != dets.shape[:1]:
        raise RuntimeError("nms expects dets [N,4] and scores [N]")
    n = dets.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    order = torch.sort(scores.float(), dim=0, descending=True, stable=True)[1]
    boxes = dets.float().index_select(0, order).contiguous()
    mask = _mask_workspace(1, n, dets.device)
    keep = torch.empty(n, dtype=torch.uint8, device=dets.device)
    st = _lib.lib().drc_nms_sorted_fwd(E._ptr(boxes), n, float(threshold), int(bool(strict)), E._ptr(mask), E._ptr(keep), E._stream_ptr(dets.device))
    _lib.check(st, "drc_nms_sorted_fwd")
    return torch.sort(order[keep.bool()])[0]


def nms_pair_sorted_joint(dets_a, dets_b, threshold, max_keep=-1, strict=True):
    """The joint keep of two box sets that are ALREADY in descending-score order (the Stereo RPN's proposals after its own sort):
    ascending indices kept in both views, at most max_keep of them (the reference's intersect + keep[:max_proposals],
    boxlist_ops.py:49-79).  No sort, the two views walked together and the walk stopped at max_keep (drc_nms_sorted_pair_joint_fwd)."""
    E.require_gpu(dets_a, "nms_pair_sorted_joint")
    n = dets_a.shape[0]
    if dets_b.shape != dets_a.shape or dets_a.dim() != 2 or dets_a.shape[1] != 4:
        raise RuntimeError("nms_pair_sorted_joint expects two [N,4] box sets")
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=dets_a.device)
    if n > 32768:                                          # beyond the joint walk's register budget: the batched walk and an AND
        s = torch.arange(n, 0, -1, dtype=torch.float32, device=dets_a.device)
        k = nms_pair(dets_a, dets_b, s, threshold, strict, joint=True)
        return k[:max_keep] if max_keep > 0 else k
    boxes = torch.stack((dets_a.float(), dets_b.float())).contiguous()
    mask = _mask_workspace(2, n, dets_a.device)
    keep = torch.empty(n, dtype=torch.uint8, device=dets_a.device)
    st = _lib.lib().drc_nms_sorted_pair_joint_fwd(E._ptr(boxes), n, float(threshold), int(bool(strict)), int(max_keep), E._ptr(mask), E._ptr(keep),
                                                  E._stream_ptr(dets_a.device))
    _lib.check(st, "drc_nms_sorted_pair_joint_fwd")
    k = keep.bool().nonzero().squeeze(1)
    return k[:max_keep] if max_keep > 0 else k


def nms_pair(dets_a, dets_b, scores, threshold, strict=True, joint=False):
    """NMS of two box sets that share their scores (the left and right views of a stereo detection list) in one launch pair:
    -> (keep_a, keep_b), each as nms() would return it; joint=True -> the ascending intersection of the two (what
    double_view_boxlist_nms keeps, boxlist_ops.py:49-79).  One sort, one mask launch, one walk launch for both views."""
    E.require_gpu(dets_a, "nms_pair")
    n = dets_a.shape[0]
    if dets_b.shape != dets_a.shape or dets_a.dim() != 2 or dets_a.shape[1] != 4 or scores.shape != dets_a.shape[:1]:
        raise RuntimeError("nms_pair expects two [N,4] box sets and scores [N]")
    if n == 0:
        e = torch.empty(0, dtype=torch.int64, device=dets_a.device)
        return e if joint else (e, e.clone())
    order = torch.sort(scores.float(), dim=0, descending=True, stable=True)[1]
    boxes = torch.stack((dets_a.float().index_select(0, order), dets_b.float().index_select(0, order))).contiguous()
    mask = _mask_workspace(2, n, dets_a.device)
    keep = torch.empty(2, n, dtype=torch.uint8, device=dets_a.device)
    st = _lib.lib().drc_nms_sorted_batch_fwd(E._ptr(boxes), 2, n, float(threshold), int(bool(strict)), E._ptr(mask), E._ptr(keep), E._stream_ptr(dets_a.device))
    _lib.check(st, "drc_nms_sorted_batch_fwd")
    kb = keep.bool()
    if joint:           # the kept sets' intersection straight from the flags: one compaction instead of two + unique/sort/compare
        return torch.sort(order[kb[0] & kb[1]])[0]
    return torch.sort(order[kb[0]])[0], torch.sort(order[kb[1]])[0]
