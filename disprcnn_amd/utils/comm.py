"""Distributed helpers (reference: disprcnn/utils/comm.py:12-116), RCCL over xGMI on MI355X (backend "nccl"), gloo on CPU.

ROIs (and images) are independent units: inference shards them across ranks with NO collective in the compute path;
the only exchange is one tensor ``all_gather`` of the [R,H,W] disparities at the end, replacing the reference's
pickle-over-NCCL gather (comm.py:47-87)."""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    """Barrier; no-op for world size 1 (reference comm.py:34-44)."""
    if get_world_size() > 1:
        dist.barrier()


def shard_range(n_items, rank=None, world=None):
    """Contiguous, balanced shard [lo,hi) of n_items for this rank (sizes differ by at most 1; empty shards allowed)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(t):
    """Gather tensors that differ only in dim 0 from every rank and concatenate them in rank order.
    Two collectives: sizes (int64[1]) then zero-padded payloads -- tensors, not pickles."""
    world = get_world_size()
    if world == 1:
        return t
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = t.new_zeros((mx,) + tuple(t.shape[1:]))
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


_DTYPE_CODES = (torch.float32, torch.int64, torch.int32, torch.uint8, torch.bool, torch.float16, torch.float64, torch.int16, torch.int8,
                torch.bfloat16)


def gather_predictions(predictions, fields=("scores", "labels", "disparity")):
    """{image_id: BoxList} of this rank -> list of BoxLists ordered by image id on the main process (None elsewhere).

    Reference: engine/inference.py:53-72 -- `all_gather` pickles every rank's dict (utils/comm.py:47-87: three collectives on a
    ByteTensor of the pickle).  Here the payload travels as tensors: one row-gather each for the per-image header
    (image id, width, height, ROI count), the boxes and every tensor field present on all images (first dim = ROIs).
    Fields that are not tensors are not transported; duplicated image ids keep the copy of the highest rank, like
    dict.update in the reference."""
    from ..structures.bounding_box import BoxList
    ids = sorted(predictions)
    dev = next((predictions[i].bbox.device for i in ids), torch.device("cpu"))
    head = torch.tensor([[i, predictions[i].size[0], predictions[i].size[1], len(predictions[i])] for i in ids],
                        dtype=torch.int64, device=dev).reshape(-1, 4)
    boxes = torch.cat([predictions[i].bbox for i in ids]) if ids else torch.zeros(0, 4, device=dev)
    # a field travels if every rank has it on every one of its images: agree on that with one small gather
    have = torch.tensor([[int(all(predictions[i].has_field(f) and torch.is_tensor(predictions[i].get_field(f)) for i in ids))
                          for f in fields]], dtype=torch.int64, device=dev)
    have = all_gather_rows(have).min(dim=0)[0].tolist() if len(fields) else []
    head_all, boxes_all = all_gather_rows(head), all_gather_rows(boxes.float())
    payload = {}
    for f, ok in zip(fields, have):
        if not ok:
            continue
        parts = [predictions[i].get_field(f) for i in ids]
        shape = next((tuple(p_.shape[1:]) for p_ in parts), None)
        dtype = next((p_.dtype for p_ in parts), None)
        # ranks without images learn the trailing shape AND the dtype from the others: gather both first (rank-independent
        # result) -- an empty payload of the wrong dtype would put mismatched byte sizes into all_gather
        meta = torch.tensor([[len(shape) if shape is not None else -1, _DTYPE_CODES.index(dtype) if dtype is not None else -1]
                             + list(shape or ()) + [0] * (8 - len(shape or ()))], dtype=torch.int64, device=dev)
        meta = all_gather_rows(meta)
        known = meta[meta[:, 0] >= 0]
        if len(known) == 0:
            continue
        if bool((known[:, 1:] != known[0, 1:]).any()):
            raise RuntimeError(f"gather_predictions: field {f!r} has different dtypes / trailing shapes on different ranks")
        shape = tuple(int(v) for v in known[0, 2:2 + int(known[0, 0])])
        dtype = _DTYPE_CODES[int(known[0, 1])]
        local = torch.cat(parts) if parts else torch.zeros((0,) + shape, dtype=dtype, device=dev)
        payload[f] = all_gather_rows(local)
    if not is_main_process():
        return None
    out, start = {}, 0
    for img_id, w, h, r in head_all.tolist():
        bl = BoxList(boxes_all[start:start + r], (w, h))
        for f, t in payload.items():
            bl.add_field(f, t[start:start + r])
        out[img_id] = bl
        start += r
    return [out[i] for i in sorted(out)]


def all_ranks_ok(ok, device=None):
    """True iff `ok` is true on EVERY rank (one tiny MIN all-reduce).  Call it before a step that contains a collective when a rank may have
    failed locally (an exception caught per rank): a rank that skips the collective would leave the others waiting forever."""
    if get_world_size() == 1:
        return bool(ok)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def sharded_inference(model, samples, fields=("scores", "labels", "disparity"), timing=None):
    """The sharded inference loop of one node (reference: engine/inference.py:24-72 -- every rank runs the model on its share of the
    images, then the per-image predictions are gathered on the main process).

    samples : a sequence of (image_id, lr_images, lr_result), the SAME on every rank (the reference reaches the same partition through
              its DistributedSampler); this rank takes the contiguous `shard_range` of it.
    model   : the detector (DispRCNN3D: forward(lr_images, lr_result) -> {"left": [BoxList], "right": [...]}); called once per sample of
              the shard with NO collective in between -- ROIs and images are independent units.
    Returns the left-view BoxLists of ALL samples ordered by image id on the main process, None on the others (gather_predictions: the
    [R,H,W] disparities travel as tensors, one row all_gather per field).  `timing` (a dict) receives compute_s / gather_s of this rank."""
    import time
    lo, hi = shard_range(len(samples))
    local = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        for image_id, lr_images, lr_result in samples[lo:hi]:
            out = model(lr_images, lr_result)
            left = out["left"] if isinstance(out, dict) else out
            local[int(image_id)] = left[0] if isinstance(left, (list, tuple)) else left
    if torch.cuda.is_available() and any(b.bbox.is_cuda for b in local.values()):
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    gathered = gather_predictions(local, fields)
    t2 = time.perf_counter()
    if timing is not None:
        timing.update(compute_s=t1 - t0, gather_s=t2 - t1, shard=(lo, hi))
    return gathered


def reduce_dict(d, average=True):
    """Reduce a dict of scalar tensors to rank 0 (reference comm.py:90-116, trainer.py:19-41)."""
    world = get_world_size()
    if world < 2:
        return d
    with torch.no_grad():
        names = sorted(d)
        vals = torch.stack([d[k] for k in names], 0)
        dist.reduce(vals, dst=0)
        if dist.get_rank() == 0 and average:
            vals = vals / world
        return dict(zip(names, vals))


class GradientSync:
    """Data-parallel gradient averaging for the train step (the reference wraps the model in DistributedDataParallel,
    tools/train_net.py:32-38; BatchNorm statistics stay per GPU there and here).

    The backward of the disparity path is one engine call that fills every ``.grad`` at once, so there is nothing to
    overlap bucket by bucket: the gradients are packed into ONE persistent fp32 buffer (PSMNet: 5.2 M parameters = 20.9 MB)
    and averaged with a single all-reduce -- on xGMI a ring all-reduce is per-link bound, and one 21 MB message amortises
    the per-collective latency that 200 small per-tensor reductions would pay.  Parameters without a gradient contribute zeros
    (every rank must reduce the same layout)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self._flat = None

    def _buffer(self):
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        if self._flat is None or self._flat.numel() != n or self._flat.device != dev:
            self._flat = torch.zeros(n, dtype=torch.float32, device=dev)
        return self._flat

    @torch.no_grad()
    def zero_grad(self):
        """Use instead of ``optimizer.zero_grad()``: every ``.grad`` becomes a zeroed VIEW of the flat buffer, so the backward pass
        accumulates straight into it and ``__call__`` all-reduces in place -- no per-parameter pack / unpack copies (2 x 260 small
        launches around one collective for PSMNet).  ``zero_grad(set_to_none=True)`` drops the views; the next
        ``__call__`` then falls back to packing."""
        flat = self._buffer()
        flat.zero_()
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n

    def _attached(self):
        flat, off = self._flat, 0
        if flat is None:
            return False
        base, esz = flat.data_ptr(), flat.element_size()
        for p in self.params:
            g = p.grad
            if g is None or g.dtype != flat.dtype or not g.is_contiguous() or g.data_ptr() != base + off * esz:
                return False
            off += p.numel()
        return True

    @torch.no_grad()
    def __call__(self):
        world = get_world_size()
        if world == 1 or not self.params:
            return
        if self._attached():                    # the gradients already live in the flat buffer (zero_grad above): one collective, no copies
            dist.all_reduce(self._flat)
            self._flat.div_(world)
            return
        flat = self._buffer()
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(flat)
        flat.div_(world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
