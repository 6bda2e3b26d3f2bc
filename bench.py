#!/usr/bin/env python3
"""bench.py -- instance-disparity hot path on MI355X.

One "step" = one pass of the hot path (a1 cost volume -> a3-a6 3D regressor -> a7 soft-argmin) over one batch of
synthetic ROI feature pairs per GPU.  Headline workload = BASELINE.json's metric shape, Config A:
112x112 ROI, 48 disparities -> cost volume [64,12,28,28] per ROI, entered at the feature boundary.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  ROIs are independent units: they are sharded across ranks with no data-path
collective (weak scaling: fixed ROIs per GPU); a barrier + max-over-ranks brackets the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOPS_PER_VOXEL_3D = 644544          # conv FLOPs (2*MAC) of dres0..classif3 per cost-volume voxel
PEAK_F32_TFLOPS = 157.3              # MI355X fp32 vector == fp32 MFMA peak (MI355X_MICROARCH.md)


def build_model(dev, maxdisp, mindisp, bn_case):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn_amd.utils import synth
    model = PSMNet(maxdisp, mindisp)
    sd = synth.synth_state_dict(model.state_dict())
    bn = os.path.join(ROOT, "tests", "golden", f"bn_stats_{bn_case}.npz")
    if os.path.exists(bn):
        synth.load_bn_stats(sd, bn)
    model.load_state_dict(sd, strict=True)
    return model.to(dev).eval(), sd


def cpu_baseline_config_a(sd, budget_s=12.0):
    """The CPU oracle (torch-CPU restatement, verified against the reference's outputs) on a bounded sample."""
    from oracle import psmnet_oracle as O
    from disprcnn_amd.utils import synth
    threads = torch.get_num_threads()
    fl, fr = synth.synth_features(4, 32, 28, 28, tag="cpu")
    with torch.no_grad():
        O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)          # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
            n += 4
            el = time.perf_counter() - t0
            if el >= budget_s:
                break
    return {"value": n / el, "unit": "ROI cost-volumes/s", "cores": threads, "kind": "port",
            "sample": f"{n} ROI pairs (batches of 4) of the same Config-A workload, {el:.1f} s wall, torch-CPU fp32 oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rois", type=int, default=256, help="ROI pairs per step per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the Config-B extra measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; used only for the barrier / max-reduce

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()

    from disprcnn_amd import engine as E
    from disprcnn_amd.utils import synth

    model, sd = build_model(dev, 48, 0, "A")
    N = args.rois
    fl, fr = synth.synth_features(N, 32, 28, 28, tag=f"bench{rank}")
    fl, fr = fl.to(dev), fr.to(dev)

    def step():
        return model.forward_from_features(fl, fr, (112, 112))

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream, same K steps, same inputs
    roofline = None
    extra = {}
    if rank == 0:
        E.TIMING = []
        with torch.no_grad():
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
        agg = {}
        for name, flops, e0, e1 in E.TIMING:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += flops
        E.TIMING = None
        dom = max(agg, key=lambda k: agg[k][1])
        calls, secs, flops = agg[dom]
        achieved = flops / secs / 1e12
        # HBM bytes per launch of the same kernel from the committed PMC pass of this command (profiles/collect.sh):
        # FETCH_SIZE (x2: gfx950 128-B request correction) + WRITE_SIZE; null if that profile is absent
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            key = [k for k in tj if k.replace(" ", "") == dom.replace(" ", "")]
            if key:
                traffic = round(tj[key[0]]["fetch_bytes_corrected"] + tj[key[0]]["write_bytes"])
        except (OSError, ValueError, KeyError):
            traffic = None
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, separate pass; profiles/r1_pmc.md)",
                    "calls_per_step": calls // args.steps, "avg_launch_us": round(secs / calls * 1e6, 2),
                    "algorithmic_flops_per_launch": flops / calls}
        if dom.startswith("wino3d"):
            # `achieved` prices the layer at the direct convolution's 2*27*Cin*Cout flops per voxel (the algorithmic figure of
            # DESIGN.md section 3); Winograd F(2,3)^3 executes 64/216 of those multiplies on the matrix cores, so `frac` may
            # exceed 1 -- `executed_frac` is what the MFMA pipe actually sustains against its peak
            roofline["executed"] = round(achieved * 64 / 216, 2)
            roofline["executed_frac"] = round(achieved * 64 / 216 / PEAK_F32_TFLOPS, 4)
            roofline["note"] = "achieved = direct-convolution flops / time; the kernel is Winograd F(2x2x2,3x3x3): 64/216 of them are executed"
        extra["kernels"] = {k: {"calls_per_step": v[0] // args.steps, "avg_us": round(v[1] / v[0] * 1e6, 2),
                                "tflops": round(v[2] / v[1] / 1e12, 2)} for k, v in agg.items()}
        step_flops = FLOPS_PER_VOXEL_3D * 12 * 28 * 28 * N
        extra["regressor_tflops_whole_step"] = round(step_flops * args.steps * world / elapsed / 1e12 / world, 2)

        # ---- extra: Config B (224x224, D=96, full PSMNet incl. the 2D feature CNN), 16 ROI pairs per step
        if not args.no_extra:
            try:
                mB, _ = build_model(dev, 48, -48, "B")
                l, r = synth.synth_images(16, 224, 224, tag="benchB")
                l, r = l.to(dev), r.to(dev)
                with torch.no_grad():
                    for _ in range(2):
                        mB((l, r))
                    torch.cuda.synchronize()
                    tb = time.perf_counter()
                    for _ in range(5):
                        mB((l, r))
                    torch.cuda.synchronize()
                    tb = (time.perf_counter() - tb) / 5
                extra["config_b_full_psmnet"] = {"roi_pairs_per_s": round(16 / tb, 1), "ms_per_16_roi_image": round(tb * 1e3, 2),
                                                 "stereo_pairs_per_s_16roi": round(1 / tb, 2),
                                                 "workload": "16 ROI crops 224x224, D=96 (-48..48): 2D CNN + cost volume + 3D + soft-argmin"}
                # ---- extra: BASELINE configs[3] shape (64 ROIs/image at 224x224x96) -- in fp32: no fp16 path is built (the
                # reference has no fp16 oracle), so this is the stress shape at the reference's own precision
                l64, r64 = synth.synth_images(64, 224, 224, tag="benchB64")
                l64, r64 = l64.to(dev), r64.to(dev)
                with torch.no_grad():
                    mB((l64, r64))
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    for _ in range(3):
                        mB((l64, r64))
                    torch.cuda.synchronize()
                    ts = (time.perf_counter() - ts) / 3
                extra["stress_64roi_224x224x96_f32"] = {"roi_pairs_per_s": round(64 / ts, 1), "ms_per_64_roi_image": round(ts * 1e3, 2),
                                                        "regressor_tflops": round(FLOPS_PER_VOXEL_3D * 24 * 56 * 56 * 64 / ts / 1e12, 1)}
                del l64, r64
                # ---- extra: BASELINE configs[1] -- one stereo pair 2x3x375x1242 through ResNet-50-FPN (2D stage trunk) plus the
                # disparity stage on 16 ROIs/image (device-side ROI pairing + ROIAlign crops + full PSMNet at 224^2 / D=96)
                from types import SimpleNamespace as NS
                from disprcnn_amd.modeling.backbone import build_backbone
                from disprcnn_amd.modeling.detector.disprcnn3d import DispRCNN3D, default_cfg
                from disprcnn_amd.structures import BoxList, ImageList
                bb = build_backbone(NS(MODEL=NS(BACKBONE=NS(CONV_BODY="R-50-FPN"), RESNETS=NS(BACKBONE_OUT_CHANNELS=256, RES2_OUT_CHANNELS=256))))
                bsd = synth.synth_backbone_state(bb.state_dict())
                bnf = os.path.join(ROOT, "tests", "golden", "bn_stats_backbone.npz")
                if os.path.exists(bnf):
                    synth.load_bn_stats(bsd, bnf)
                bb.load_state_dict(bsd)
                bb = bb.to(dev).eval()
                det = DispRCNN3D(default_cfg(48, -48, 224))
                det.dispnet = mB
                det = det.to(dev).eval()
                Wi, Hi = 1242, 375
                pair = synth.hash_uniform("benchpair", (2, 3, Hi, Wi), 0.0, 1.0).to(dev)
                u = synth.hash_uniform("benchboxes", (16, 4), 0.0, 1.0)
                x1 = 20 + u[:, 0] * (Wi - 400); y1 = 10 + u[:, 1] * (Hi - 240)
                lb = torch.stack([x1, y1, x1 + 40 + u[:, 2] * 260, y1 + 30 + u[:, 3] * 170], 1)
                rb = lb.clone(); rb[:, [0, 2]] -= 2 + 78 * u[:, 0:1]
                rb[:, [0, 2]] = rb[:, [0, 2]].clamp(min=0)

                def pair_step():
                    feats = bb(pair)
                    out = det({"left": ImageList(pair[:1], [(Hi, Wi)]), "right": ImageList(pair[1:], [(Hi, Wi)])},
                              {"left": [BoxList(lb, (Wi, Hi))], "right": [BoxList(rb, (Wi, Hi))]})
                    return feats, out
                with torch.no_grad():
                    for _ in range(2):
                        pair_step()
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                    for _ in range(5):
                        pair_step()
                    torch.cuda.synchronize()
                    tp = (time.perf_counter() - tp) / 5
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    for _ in range(5):
                        bb(pair)
                    torch.cuda.synchronize(); tbb = (time.perf_counter() - t1) / 5
                extra["kitti_pair_r50fpn_plus_16roi"] = {
                    "stereo_pairs_per_s": round(1.0 / tp, 2), "ms_per_pair": round(tp * 1e3, 2), "backbone_ms": round(tbb * 1e3, 2),
                    "backbone_tflops": round(250.3e9 / tbb / 1e12, 2),
                    "workload": "BASELINE configs[1]: R-50-FPN on 2x3x375x1242 (250.3 GFLOP/pair, SURVEY a12) + 16 ROIs: pairing, ROIAlign crops, PSMNet 224^2 D=96"}
                del bb, det
                # ---- extra: one TRAIN step of the disparity stage (reference: PSMNet.train() + PSMLoss, trainer.do_train): forward with
                # per-GPU batch-stat BatchNorm, 3-head smooth-L1 loss, full backward (dgrad + MFMA wgrad) on the HIP engine, gradient
                # sync (GradientSync: a no-op at world size 1), optimizer step; 64 ROI pairs of Config A from the feature boundary, and 8 ROI crops of
                # Config B through the 2D CNN as well
                from disprcnn_amd.utils.loss_utils import PSMLoss
                from disprcnn_amd.utils.comm import GradientSync
                tr = {}
                for tag, mdl, nroi in (("config_a_from_features_64roi", model, 64), ("config_b_full_psmnet_8roi", mB, 8)):
                    mdl.train()
                    sync = GradientSync(mdl.parameters())
                    if tag.startswith("config_a"):
                        fl_, fr_ = synth.synth_features(nroi, 32, 28, 28, tag="trainA")
                        fl_, fr_ = fl_.to(dev), fr_.to(dev)
                        tgt = synth.hash_uniform("trainA:t", (nroi, 112, 112), 0.0, 47.0).to(dev)
                        fwd = lambda: mdl.forward_from_features(fl_, fr_, (112, 112))
                        fl3 = FLOPS_PER_VOXEL_3D * 12 * 28 * 28 * nroi
                    else:
                        li, ri = synth.synth_images(nroi, 224, 224, tag="trainB")
                        li, ri = li.to(dev), ri.to(dev)
                        tgt = synth.hash_uniform("trainB:t", (nroi, 224, 224), -47.0, 47.0).to(dev)
                        fwd = lambda: mdl({"left": li, "right": ri})
                        fl3 = FLOPS_PER_VOXEL_3D * 24 * 56 * 56 * nroi
                    msk = torch.ones_like(tgt, dtype=torch.uint8)
                    crit = PSMLoss()
                    opt = torch.optim.SGD(mdl.parameters(), lr=1e-7, momentum=0.9)   # the parameters change every step: weights are re-packed

                    def train_step():
                        opt.zero_grad(set_to_none=True)
                        loss = crit(fwd(), {"disparity": tgt, "mask": msk})
                        loss.backward()
                        sync()
                        opt.step()
                        return loss
                    for _ in range(2):
                        train_step()
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    for _ in range(3):
                        train_step()
                    torch.cuda.synchronize(); tt = (time.perf_counter() - t1) / 3
                    tr[tag] = {"ms_per_step": round(tt * 1e3, 2), "roi_pairs_per_s": round(nroi / tt, 1),
                               "regressor_tflops_fwd_bwd": round(3 * fl3 / tt / 1e12, 2)}
                    mdl.eval()
                tr["workload"] = "forward (batch-stat BN) + PSMLoss + backward + gradient sync + SGD step; regressor FLOPs counted as 3x forward"
                extra["train_step"] = tr
                del mB
                # ---- extra: post-processing: 16 images 375x1242, 16 ROI maps 224x224 each -> full-image disparity maps
                from disprcnn_amd import ops as _ops
                gpp = torch.Generator().manual_seed(0)
                nimg, nr, ih, iw = 16, 16, 375, 1242
                x1 = torch.rand(nimg * nr, generator=gpp) * (iw - 260); y1 = torch.rand(nimg * nr, generator=gpp) * (ih - 180)
                lbp = torch.stack([x1, y1, x1 + 60 + torch.rand(nimg * nr, generator=gpp) * 190, y1 + 40 + torch.rand(nimg * nr, generator=gpp) * 130], 1)
                rbp = lbp.clone(); rbp[:, 0] = (lbp[:, 0] - 30).clamp(min=0); rbp[:, 2] = lbp[:, 2] - 25
                dpp = (torch.rand(nimg * nr, 224, 224, generator=gpp) * 96 - 48).to(dev)
                b6 = _ops.integer_roi_boxes(lbp.to(dev), rbp.to(dev))
                for _ in range(3):
                    _ops.disparity_paste(dpp, b6, [nr] * nimg, ih, iw)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(20):
                    _ops.disparity_paste(dpp, b6, [nr] * nimg, ih, iw)
                torch.cuda.synchronize(); tp = (time.perf_counter() - t1) / 20
                pbytes = nimg * ih * iw * 4 + dpp.numel() * 4
                extra["post_process_16img_x16roi"] = {"us_per_call": round(tp * 1e6, 1), "images_per_s": round(nimg / tp, 1),
                                                      "algorithmic_GB_per_s": round(pbytes / tp / 1e9, 1),
                                                      "workload": "drc_disparity_paste_fwd: 256 ROI maps 224^2 -> 16 maps 375x1242 (one launch; bytes = maps read once + outputs written once)"}
            except Exception as ex:  # report, never hide
                extra["config_b_full_psmnet"] = extra.get("config_b_full_psmnet") or {"error": repr(ex)}
                extra["extra_error"] = repr(ex)

    cpu = None
    if rank == 0 and not args.no_cpu:
        cpu = cpu_baseline_config_a(sd)

    if rank == 0:
        total_rois = N * args.steps * world
        line = {
            "metric": "ROI cost-volumes/sec (112x112x48)", "value": round(total_rois / elapsed, 1), "unit": "ROI cost-volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Config A: per ROI pair, features [32,28,28]x2 -> concat cost volume [64,12,28,28] -> "
                                   "3D stacked-hourglass regressor -> trilinear x4 + softmax + soft-argmin -> disparity [112,112]",
                       "rois_per_step_per_gpu": N, "maxdisp": 48, "mindisp": 0, "parallelism": f"roi-shard x{world} (no collective)",
                       "weights": "closed-form synthetic (disprcnn_amd.utils.synth), BN stats calibrated fixture"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
