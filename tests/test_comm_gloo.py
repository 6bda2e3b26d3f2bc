"""world_size-2 gloo test (CPU) of the N>1 path: ROI sharding without a data-path collective + one tensor all_gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from disprcnn_amd.utils import comm


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_rois, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert comm.get_world_size() == world and comm.get_rank() == rank and comm.is_main_process() == (rank == 0)
        lo, hi = comm.shard_range(n_rois)
        # stand-in for the per-ROI work: every ROI i yields a [H,W] map filled with i (independent units, no exchange)
        local = torch.stack([torch.full((3, 4), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 3, 4)
        comm.synchronize()
        full = comm.all_gather_rows(local)
        red = comm.reduce_dict({"loss": torch.tensor(float(rank + 1))})
        # train step: per-rank gradients averaged by one flat all-reduce; a parameter that got no gradient on one rank
        # (e.g. an empty ROI shard) still takes part
        w = torch.nn.Parameter(torch.zeros(5)); b = torch.nn.Parameter(torch.zeros(2, 3)); frozen = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        w.grad = torch.full((5,), float(rank + 1))
        if rank == 0:
            b.grad = torch.full((2, 3), 4.0)
        comm.GradientSync([w, b, frozen])()
        assert torch.allclose(w.grad, torch.full((5,), 1.5)) and torch.allclose(b.grad, torch.full((2, 3), 2.0)) and frozen.grad is None
        # the copy-free form: zero_grad() makes every .grad a view of the flat buffer, autograd accumulates into it, one collective in place
        w2 = torch.nn.Parameter(torch.arange(5.0)); b2 = torch.nn.Parameter(torch.ones(2, 3))
        sync = comm.GradientSync([w2, b2, frozen])
        sync.zero_grad()
        ptrs = (w2.grad.data_ptr(), b2.grad.data_ptr())
        ((w2 * float(rank + 1)).sum() + (b2.sum() * 4.0 if rank == 0 else b2.sum() * 0.0)).backward()
        assert (w2.grad.data_ptr(), b2.grad.data_ptr()) == ptrs and sync._attached()         # accumulated in place
        sync()
        assert (w2.grad.data_ptr(), b2.grad.data_ptr()) == ptrs
        assert torch.allclose(w2.grad, torch.full((5,), 1.5)) and torch.allclose(b2.grad, torch.full((2, 3), 2.0))
        sync.zero_grad()
        assert float(w2.grad.abs().sum()) == 0.0 and w2.grad.data_ptr() == ptrs[0]
        q.put((rank, lo, hi, full[:, 0, 0].tolist(), float(red["loss"])))
    finally:
        dist.destroy_process_group()


def _run(n_rois):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_rois, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_shard_and_gather_two_ranks():
    res = _run(7)
    (r0, lo0, hi0, g0, l0), (r1, lo1, hi1, g1, l1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)                 # balanced contiguous shards
    assert g0 == g1 == [float(i) for i in range(7)]             # gathered in rank order, ragged sizes handled
    assert abs(l0 - 1.5) < 1e-6                                 # reduce_dict averages on rank 0


def test_empty_shard_is_handled():
    res = _run(1)                                               # rank 1 owns nothing: must still join the collectives
    assert res[0][3] == res[1][3] == [0.0]


def test_single_process_fallbacks():
    assert comm.get_world_size() == 1 and comm.get_rank() == 0 and comm.is_main_process()
    w = torch.nn.Parameter(torch.ones(3)); w.grad = torch.full((3,), 2.0)
    comm.GradientSync([w])()                                    # world size 1: gradients untouched, no collective
    assert torch.equal(w.grad, torch.full((3,), 2.0))
    comm.synchronize()
    t = torch.arange(6.).view(2, 3)
    assert torch.equal(comm.all_gather_rows(t), t)
    assert comm.shard_range(10, 2, 4) == (6, 8) and comm.shard_range(2, 3, 4) == (2, 2)


def _pred_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disprcnn_amd.structures import BoxList
        preds = {}
        for img in ([0, 3] if rank == 0 else [1]):                 # rank 1 holds one image, image 2 is missing everywhere
            r = img + 1
            bl = BoxList(torch.arange(r * 4, dtype=torch.float32).reshape(r, 4) + 100 * img, (320 + img, 96))
            bl.add_field("scores", torch.full((r,), 0.1 * img))
            bl.add_field("disparity", torch.full((r, 2, 3), float(img)))
            if rank == 0:
                bl.add_field("labels", torch.ones(r, dtype=torch.int64))      # not on rank 1: must not travel
            preds[img] = bl
        got = comm.gather_predictions(preds)
        if rank == 0:
            q.put([(b.size, len(b), b.bbox[0, 0].item(), sorted(b.fields()), b.get_field("disparity")[0, 0, 0].item(),
                    round(b.get_field("scores")[0].item(), 3)) for b in got])
        else:
            assert got is None
            q.put("none")
    finally:
        dist.destroy_process_group()


def test_gather_predictions_two_ranks():
    """Tensor gather of per-image predictions (the reference pickles them): ordered by image id on rank 0, variable ROI
    counts, a field missing on one rank is dropped, image sizes travel."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pred_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    main = next(r for r in res if r != "none")
    assert [m[0] for m in main] == [(320, 96), (321, 96), (323, 96)]
    assert [m[1] for m in main] == [1, 2, 4]
    assert [m[2] for m in main] == [0.0, 100.0, 300.0]
    assert all(m[3] == ["disparity", "scores"] for m in main)
    assert [m[4] for m in main] == [0.0, 1.0, 3.0] and [m[5] for m in main] == [0.0, 0.1, 0.3]


def _pred_empty_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disprcnn_amd.structures import BoxList
        preds = {}
        if rank == 0:                                              # fewer images than ranks: rank 1 holds nothing
            for img, r in ((0, 2), (1, 0), (2, 3)):                # image 1 has no ROIs
                bl = BoxList(torch.arange(r * 4, dtype=torch.float32).reshape(r, 4), (1242, 375))
                bl.add_field("scores", torch.linspace(0.5, 1.0, r))
                bl.add_field("labels", torch.arange(r, dtype=torch.int64) + 10 * img)
                bl.add_field("disparity", torch.full((r, 4, 4), float(img)))
                preds[img] = bl
        got = comm.gather_predictions(preds)
        if rank == 0:
            q.put([(len(b), b.get_field("labels").dtype == torch.int64, b.get_field("labels").tolist(), tuple(b.get_field("disparity").shape))
                   for b in got])
        else:
            assert got is None
            q.put("none")
    finally:
        dist.destroy_process_group()


def test_gather_predictions_rank_without_images_int64_field():
    """A rank that holds no image must send an empty payload of the AGREED dtype (int64 'labels'), not float32: mismatched
    byte sizes in all_gather are an error on gloo and undefined on RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pred_empty_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    main = next(r for r in res if r != "none")
    assert [m[0] for m in main] == [2, 0, 3] and all(m[1] for m in main)
    assert main[0][2] == [0, 1] and main[2][2] == [20, 21, 22]
    assert main[2][3] == (3, 4, 4)


# ------------------------------------------------------------------ the product's sharded inference loop
def _sharded_worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disprcnn_amd.structures.bounding_box import BoxList
        calls = []

        class StubDetector:
            """Stands in for DispRCNN3D on the CPU: image i with r = i % 3 ROIs gets r boxes and a [r,4,5] 'disparity' filled with i."""

            def __call__(self, lr_images, lr_result):
                i = int(lr_images["left"])
                calls.append(i)
                b = lr_result["left"][0]
                b.add_field("disparity", torch.full((len(b), 4, 5), float(i)))
                return {"left": [b], "right": lr_result["right"]}

        def sample(i):
            r = i % 3
            boxes = torch.tensor([[1.0, 2.0, 3.0 + j, 4.0 + i] for j in range(r)]).reshape(-1, 4)
            b = BoxList(boxes, (100 + i, 50))
            b.add_field("scores", torch.full((r,), 0.5 + 0.01 * i))
            b.add_field("labels", torch.ones(r, dtype=torch.int64))
            return (i, {"left": i, "right": i}, {"left": [b], "right": [b]})

        samples = [sample(i) for i in range(n_images)]
        timing = {}
        got = comm.sharded_inference(StubDetector(), samples, timing=timing)
        lo, hi = comm.shard_range(n_images)
        assert calls == list(range(lo, hi)) and timing["shard"] == (lo, hi)        # only this rank's shard was computed
        ok_all = comm.all_ranks_ok(True)
        ok_one = comm.all_ranks_ok(rank != 1)                                       # rank 1 "failed": every rank learns it
        if rank == 0:
            assert got is not None and len(got) == n_images
            for i, b in enumerate(got):
                assert len(b) == i % 3 and b.size == (100 + i, 50)
                assert torch.equal(b.get_field("disparity"), torch.full((i % 3, 4, 5), float(i)))
                assert torch.allclose(b.get_field("scores"), torch.full((i % 3,), 0.5 + 0.01 * i)) and b.get_field("labels").dtype == torch.int64
        else:
            assert got is None
        q.put((rank, ok_all, ok_one))
    finally:
        dist.destroy_process_group()


def test_sharded_inference_world2_gloo():
    """comm.sharded_inference: shard -> model per sample (no collective) -> tensor gather on rank 0, with a stub detector; ragged ROI
    counts incl. images without ROIs; and the all-ranks-ok flag that guards collective-bearing steps."""
    for n_images in (7, 1):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=120) for _ in ps)
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res == [(0, True, False), (1, True, False)]
