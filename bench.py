#!/usr/bin/env python3
"""bench.py -- instance-disparity hot path on MI355X.

One "step" = one pass of the hot path (a1 cost volume -> a3-a6 3D regressor -> a7 soft-argmin) over one batch of
synthetic ROI feature pairs per GPU.  Headline workload = BASELINE.json's metric shape, Config A:
112x112 ROI, 48 disparities -> cost volume [64,12,28,28] per ROI, entered at the feature boundary.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N ...        (no WORLD_SIZE in the env: re-launches itself under torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  ROIs are independent units: they are sharded across ranks with no data-path
collective (weak scaling: fixed ROIs per GPU); a barrier + max-over-ranks brackets the timed region.  For world > 1 the
only collective-bearing extra (the train step: one flat gradient all-reduce) runs on EVERY rank; the rank-0-only extras
contain no collective.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DEFAULT_ROIS = 1024                  # ROI pairs per step and GPU: 64 images x 16 ROIs (BASELINE configs[1] has 16 ROIs per image)
FLOPS_PER_VOXEL_3D = 644544          # conv FLOPs (2*MAC) of dres0..classif3 per cost-volume voxel
FLOPS_2D_PER_IMAGE = 22192734208     # feature_extraction conv FLOPs per 224x224 image
FLOPS_BACKBONE_PAIR = 250.3e9        # R-50-FPN on one 2x3x375x1242 stereo pair
PEAK_F16_TFLOPS = 2500.0             # MI355X dense f16/bf16 MFMA peak (MI355X_MICROARCH.md); the split-f16 kernels spend 3 products per fp32 product
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


def cpu_baseline_config_a(sd, batches=(1, 16, 64), warmups=2, runs=5):
    """The CPU oracle (torch-CPU restatement, verified against the reference's outputs) timed as SURVEY 8d prescribes: fp32, no_grad,
    every host thread torch has (stated), for N in {1, 16, 64} ROI pairs: 2 warm-ups, median of 5 runs.  `value` = the best batch's rate."""
    import statistics

    from oracle import psmnet_oracle as O
    from disprcnn_amd.utils import synth
    threads = torch.get_num_threads()
    per, spent, total = {}, 0.0, 0
    with torch.no_grad():
        for nb in batches:
            fl, fr = synth.synth_features(nb, 32, 28, 28, tag="cpu")
            for _ in range(warmups if nb < 64 else 1):               # (64 pairs take seconds per run: one warm-up there)
                O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
            ts = []
            for _ in range(runs):
                t0 = time.perf_counter()
                O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            per[str(nb)] = {"roi_pairs_per_s": round(nb / med, 2), "median_s": round(med, 4), "runs": runs}
            spent += sum(ts); total += nb * runs
    best = max(per, key=lambda k: per[k]["roi_pairs_per_s"])
    return {"value": per[best]["roi_pairs_per_s"], "unit": "ROI cost-volumes/s", "cores": threads, "kind": "port", "per_batch": per,
            "sample": (f"Config-A workload, torch-CPU fp32 oracle, batches of {', '.join(map(str, batches))} ROI pairs: 2 warm-ups (1 at 64), median of "
                       f"{runs} runs each; value = the best batch ({best}); {total} ROI pairs in {spent:.1f} s of timed CPU work")}


def timed_steps(step, steps, warmup, world, sync):
    """W untimed warm-up steps, then exactly K steps between (barrier + device sync) pairs; MAX over ranks."""
    out = None
    for _ in range(warmup):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    return out, elapsed


def max_over_ranks(elapsed, world, dev):
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


DTYPE = ("f32 (every convolution of the regressor, the three 32->1 heads included: split-f16 -- each fp32 value carried as hi + lo fp16, 3 f16-MFMA "
         "products per fp32 product, f32 accumulate; fp32-class error, tests/test_hip_s16.py.  Head gather and soft-argmin: f32 VALU)")


def _gpu_sensors(index=0):
    """{power_w, sclk_mhz, temp_c} of the GPU from the amdgpu hwmon / sysfs files (no subprocess: this is sampled inside the sustained loop);
    a field is None where the file is absent (an ordinary user on the box may not see all of them)."""
    import glob
    out = {"power_w": None, "sclk_mhz": None, "temp_c": None}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
    if not cards:
        return out
    hw = cards[min(index, len(cards) - 1)]

    def rd(name, scale):
        try:
            return round(int(open(os.path.join(hw, name)).read().strip()) * scale, 1)
        except (OSError, ValueError):
            return None
    out["power_w"] = rd("power1_average", 1e-6) or rd("power1_input", 1e-6)
    out["sclk_mhz"] = rd("freq1_input", 1e-6)
    out["temp_c"] = rd("temp1_input", 1e-3)
    return out


def sustained_run(step, rois_per_step, seconds, short_rate):
    """the headline window is 20 steps = 0.4 s; a power- or temperature-limited chip may not hold that rate.  The same step
    looped for >= `seconds` (every step synchronised by the range-guard read, as in the timed region), ROI/s per one-second bucket with the
    sensor readings sampled at each bucket's end; `tail` = the mean rate of the last third of the buckets."""
    buckets, sensors = [], []
    t_start = time.perf_counter()
    t_b, n_b = t_start, 0
    while True:
        step()
        torch.cuda.synchronize()
        n_b += 1
        now = time.perf_counter()
        if now - t_b >= 1.0:
            buckets.append(round(n_b * rois_per_step / (now - t_b), 1))
            sensors.append(_gpu_sensors())
            t_b, n_b = now, 0
            if now - t_start >= seconds:
                break
    third = max(1, len(buckets) // 3)
    tail = sum(buckets[-third:]) / third
    return {"seconds": round(time.perf_counter() - t_start, 2), "roi_pairs_per_s_per_second_bucket": buckets,
            "tail_roi_pairs_per_s": round(tail, 1), "short_window_roi_pairs_per_s": round(short_rate, 1),
            "tail_over_short_window": round(tail / short_rate, 4),
            "power_w": [s_["power_w"] for s_ in sensors], "sclk_mhz": [s_["sclk_mhz"] for s_ in sensors], "temp_c": [s_["temp_c"] for s_ in sensors],
            "rule": "if the tail is more than 3 % below the K-step figure, `value` / `ms_per_step` report the sustained tail (value_source says so)"}


def headline(total_rois, elapsed, args, world, N, roofline, cpu, extra):
    return {
        "metric": "ROI cost-volumes/sec (112x112x48)", "value": round(total_rois / elapsed, 1), "unit": "ROI cost-volumes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "Config A: per ROI pair, features [32,28,28]x2 -> concat cost volume [64,12,28,28] (folded into the first "
                               "3D layer's loads, never written) -> 3D stacked-hourglass regressor -> trilinear x4 + softmax + soft-argmin -> disparity [112,112]",
                   "rois_per_step_per_gpu": N, "maxdisp": 48, "mindisp": 0, "parallelism": f"roi-shard x{world} (no collective)",
                   "weights": "closed-form synthetic (disprcnn_amd.utils.synth), BN stats calibrated fixture"},
        "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
    }


def dry_run_cpu(args, world, rank):
    """The launcher / rank / barrier / max-over-ranks / collective-on-every-rank / rank-0-prints control flow of this file on
    the gloo backend with a stand-in step (a sleep): what tests/test_bench_flow.py drives at world size 2.  No kernel runs
    and no throughput is claimed -- the line is marked as a dry run."""
    from disprcnn_amd.utils.comm import GradientSync
    dev = torch.device("cpu")
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    _, elapsed = timed_steps(lambda: time.sleep(0.002 * (rank + 1)), args.steps, args.warmup, world, lambda: None)
    elapsed = max_over_ranks(elapsed, world, dev)
    # the one collective-bearing extra (train step) is entered by every rank
    w = torch.nn.Parameter(torch.zeros(1000))
    w.grad = torch.full((1000,), float(rank + 1))
    sync = GradientSync([w])
    t0 = time.perf_counter()
    sync()
    allreduce_ms = (time.perf_counter() - t0) * 1e3
    ok = bool(torch.allclose(w.grad, torch.full((1000,), (world + 1) / 2.0)))
    if rank == 0:
        line = headline(args.rois * args.steps * world, elapsed, args, world, args.rois, None, None,
                        {"dry_run_cpu": True, "grad_sync_ok": ok, "grad_allreduce_ms": round(allreduce_ms, 3)})
        line["data"] = "none (dry run of the control flow on CPU/gloo: no kernels executed, value is not a measurement)"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rois", type=int, default=DEFAULT_ROIS, help="ROI pairs per step per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--sustain-seconds", type=float, default=10.0, help="rank 0, one GPU: loop the headline step this long for extra.sustained (0: skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (Config B, stress, KITTI pair, train step, post-processing)")
    ap.add_argument("--dry-run-cpu", action="store_true", help=argparse.SUPPRESS)   # tests/test_bench_flow.py: control flow on gloo, no kernels
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run (RCCL rendezvous on
        # 127.0.0.1), the ranks print the single JSON line
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_cpu:
        return dry_run_cpu(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; the data path has no collective (barrier / max-reduce only)

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
        out, elapsed = timed_steps(step, args.steps, args.warmup, world, torch.cuda.synchronize)
    assert torch.isfinite(out).all()
    elapsed = max_over_ranks(elapsed, world, dev)

    # ---- the same step held for >= 10 s (rank 0, one GPU): does the K-step rate survive power / thermal steady state?
    extra = {}
    value_override = None
    if rank == 0 and world == 1 and args.sustain_seconds > 0:
        with torch.no_grad():
            sus = sustained_run(step, N, args.sustain_seconds, N * args.steps / elapsed)
        extra["sustained"] = sus
        if sus["tail_over_short_window"] < 0.97:
            value_override = sus["tail_roi_pairs_per_s"]

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream, same K steps, same inputs
    roofline = None
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
        traffic, traffic_src = None, "no committed PMC pass"
        from disprcnn_amd.csrc.build import source_digest
        cur_sha = source_digest()
        profs = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json") and f.count("_") == 1), reverse=True)
        for prof in profs:                      # newest round first (r6 > r5c > r5 > ...); only the headline command's files (<tag>_traffic.json)
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", prof)))
                key = [k for k in tj if k != "__meta__" and k.replace(" ", "") == dom.replace(" ", "")]
                if not key:
                    continue
                meta = tj.get("__meta__") or {}
                if meta.get("csrc_sha") != cur_sha:
                    #  a PMC pass of OTHER kernel sources says nothing about this run: null, and say which profile went stale
                    traffic_src = (f"stale: profiles/{prof} is from commit {str(meta.get('commit', 'unknown'))[:12]} (kernel sources {meta.get('csrc_sha')}), "
                                   f"this run's sources are {cur_sha}; re-collect with profiles/collect_all.sh")
                    break
                traffic = round(tj[key[0]]["fetch_bytes_corrected"] + tj[key[0]]["write_bytes"])
                traffic_src = f"profiles/{prof.replace('_traffic.json', '_pmc.md')}, commit {str(meta.get('commit'))[:12]}, kernel sources {cur_sha}"
                break
            except (OSError, ValueError, KeyError):
                continue
        # `achieved` = flops the kernel EXECUTES on the matrix cores per second.  For the direct kernels that is the algorithmic
        # 2*27*Cin*Cout per voxel; the Winograd F(2x2x2,3x3x3) kernel executes 64 multiplies per 2x2x2 tile and
        # (cin,cout) pair where the direct form needs 216, so its executed flops are 64/216 of the direct-convolution figure --
        # that is what is compared with the MFMA peak (frac <= 1).  The direct-convolution-equivalent rate (the layer's
        # algorithmic flops / time, which may exceed the peak) is reported separately and never as `frac`.
        exec_ratio = 64.0 / 216.0 if dom.startswith("wino3d") else 1.0
        s16_dom = dom.startswith("convs16")
        # split-f16 kernel (convs16.hip): every fp32 product is three f16 MFMA products, so the roofline it is priced against is the f16
        # MFMA peak / 3 in fp32-equivalent flops (frac = executed f16 flops / 2.5 PF, the same number)
        peak = PEAK_F16_TFLOPS / 3.0 if s16_dom else PEAK_F32_TFLOPS
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved * exec_ratio, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved * exec_ratio / peak, 4), "traffic": traffic,
                    "traffic_unit": f"HBM bytes per launch (rocprofv3 PMC, separate pass; {traffic_src})",
                    "calls_per_step": calls // args.steps, "avg_launch_us": round(secs / calls * 1e6, 2),
                    "executed_flops_per_launch": flops / calls * exec_ratio,
                    "direct_conv_equivalent_tflops": round(achieved, 2),
                    "direct_conv_equivalent_flops_per_launch": flops / calls,
                    "note": (("achieved = the layer's algorithmic fp32 flops (SURVEY 8d: 2*27*Cin*Cout per voxel) / launch time; the kernel spends three "
                              "v_mfma_f32_32x32x16_f16 products per fp32 product (hi*hi + lo*hi + hi*lo, fp32 accumulate), so peak = 2500 TFLOP/s dense f16 "
                              "MFMA / 3; 12.5 % of the issued MFMA columns (lanes 28..31 of a 28-voxel row) are idle on top.  A '...,true>' instantiation is the "
                              "32->32 layer with the 32->1 head fused behind it: its flops are those of both layers") if s16_dom else
                             ("achieved/frac = executed MFMA flops (Winograd: 64/216 of the direct convolution's) vs the fp32 MFMA peak; "
                              "direct_conv_equivalent_* = SURVEY 8d's algorithmic conv flops / time, not a roofline fraction"))}
        if s16_dom:
            roofline["executed_f16_mfma_tflops"] = round(3 * achieved, 1)
        extra["kernels"] = {k: {"calls_per_step": v[0] // args.steps, "avg_us": round(v[1] / v[0] * 1e6, 2),
                                "tflops": round(v[2] / v[1] / 1e12, 2)} for k, v in agg.items()}
        step_flops = FLOPS_PER_VOXEL_3D * 12 * 28 * 28 * N
        extra["regressor_tflops_whole_step"] = round(step_flops * args.steps * world / elapsed / 1e12 / world, 2)
        # batch sensitivity of the same workload (round 1's headline used 256 ROI pairs per step): tile-group rounds and per-launch
        # tails amortise with the batch; 288 GB of HBM hold far more than 1024 ROIs' activations (1.6 GB)
        if not args.no_extra:
            bs = {}
            for nb in (16, 64, 256, 512):
                if nb >= N:
                    continue
                with torch.no_grad():
                    tb_ = _time(lambda: model.forward_from_features(fl[:nb], fr[:nb], (112, 112)), 3, 5)
                plans = (model._rt._ws.get(("3ds16", nb, 12, 28, 28)) or model._rt._ws[("3d", nb, 12, 28, 28)])["p"]   # the launch heuristics depend on the batch: name what ran
                bs[str(nb)] = {"roi_pairs_per_s": round(nb / tb_, 1), "ms_per_step": round(tb_ * 1e3, 3),
                               "kernels": {k: plans[k].kname for k in ("dres1.0", "hg1.conv1", "hg1.conv2", "hg1.conv4", "hg1.conv5")}}
            bs["note"] = "batches of <= 96 units are replayed from a captured HIP graph (PSMNet.graph_eval = 'auto'): the eager step is host-bound there"
            extra["batch_sensitivity_rois_per_step"] = bs

    # ---- extras.  The train step is the only one with a collective (one flat gradient all-reduce): EVERY rank enters it.
    # The others have none and run on rank 0 at world size 1 only (the scaling runs stay short; their numbers do not depend on N).
    if rank == 0:
        extra["world"] = world_info(dev, world)
    if not args.no_extra:
        try:
            tr = train_step_extra(dev, model, world)
            if rank == 0:
                extra["train_step"] = tr
        except Exception as ex:  # report, never hide (the step itself agrees on an ok flag before its collective: no rank is left behind)
            extra["train_step"] = {"error": repr(ex)}
        try:
            si = sharded_inference_extra(dev, world)      # the product's inference loop: shard -> DispRCNN3D -> tensor gather (every rank)
            if rank == 0:
                extra["sharded_inference"] = si
        except Exception as ex:
            extra["sharded_inference"] = {"error": repr(ex)}
        if rank == 0 and world == 1:
            try:
                rank0_extras(dev, extra)
            except Exception as ex:
                extra["extra_error"] = repr(ex)
        elif rank == 0:
            extra["skipped_for_world_gt_1"] = ["config_b_full_psmnet", "stress_64roi_224x224x96", "kitti_pair_r50fpn_plus_16roi", "post_process_16img_x16roi"]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline_config_a(sd)

    if rank == 0:
        line = headline(N * args.steps * world, elapsed, args, world, N, roofline, cpu, extra)
        line["value_source"] = f"{args.steps} timed steps (barrier + synchronize on both sides, max over ranks)"
        if value_override is not None:
            line["value_k_steps"], line["ms_per_step_k_steps"] = line["value"], line["ms_per_step"]
            line["value"], line["ms_per_step"] = value_override, round(N / value_override * 1e3, 3)
            line["value_source"] = (f"sustained tail of a {extra['sustained']['seconds']} s loop of the same step (more than 3 % below the {args.steps}-step "
                                    f"figure, kept as value_k_steps)")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def world_info(dev, world):
    """What the driver's scaling run can check: ranks, the device of rank 0, the collective library behind backend "nccl"."""
    info = {"ranks": world, "device": torch.cuda.get_device_name(dev), "backend": "nccl (RCCL)" if world > 1 else "none (single rank)"}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as ex:          # noqa: BLE001
        info["rccl_version"] = repr(ex)[:80]
    if world > 1:
        names = [None] * world
        dist.all_gather_object(names, torch.cuda.get_device_name(dev))
        info["devices"] = names
    return info


def sharded_inference_extra(dev, world):
    """disprcnn_amd.utils.comm.sharded_inference on every rank: 4 synthetic KITTI-sized images x 16 ROIs per rank, DispRCNN3D (device-side
    pairing + ROIAlign crops + PSMNet 224^2, D=96) on this rank's shard, then ONE tensor gather of the [R,224,224] disparities on rank 0
    (reference: engine/inference.py:53-72 pickles them).  Reports the gather time next to the compute time."""
    from disprcnn_amd.modeling.detector.disprcnn3d import DispRCNN3D, default_cfg
    from disprcnn_amd.structures import BoxList, ImageList
    from disprcnn_amd.utils import synth
    from disprcnn_amd.utils.comm import sharded_inference
    mB, _ = build_model(dev, 48, -48, "B")
    det = DispRCNN3D(default_cfg(48, -48, 224))
    det.dispnet = mB
    det = det.to(dev).eval()
    Wi, Hi, per_rank, nroi = 1242, 375, 4, 16
    samples = []
    for i in range(per_rank * world):
        pair = synth.hash_uniform(f"shard:{i}", (2, 3, Hi, Wi), 0.0, 1.0).to(dev)
        u = synth.hash_uniform(f"shardboxes:{i}", (nroi, 4), 0.0, 1.0)
        x1 = 20 + u[:, 0] * (Wi - 400); y1 = 10 + u[:, 1] * (Hi - 240)
        lb = torch.stack([x1, y1, x1 + 40 + u[:, 2] * 260, y1 + 30 + u[:, 3] * 170], 1)
        rb = lb.clone(); rb[:, [0, 2]] -= 2 + 78 * u[:, 0:1]
        rb[:, [0, 2]] = rb[:, [0, 2]].clamp(min=0)
        bl, br = BoxList(lb.to(dev), (Wi, Hi)), BoxList(rb.to(dev), (Wi, Hi))
        for b in (bl, br):
            b.add_field("scores", torch.ones(nroi, device=dev)); b.add_field("labels", torch.ones(nroi, dtype=torch.int64, device=dev))
        samples.append((i, {"left": ImageList(pair[:1], [(Hi, Wi)]), "right": ImageList(pair[1:], [(Hi, Wi)])}, {"left": [bl], "right": [br]}))
    sharded_inference(det, samples)                   # warm-up (plans, workspaces)
    timing = {}
    got = sharded_inference(det, samples, timing=timing)
    out = {"images": len(samples), "images_per_rank": per_rank, "rois_per_image": nroi,
           "compute_ms_rank0": round(timing["compute_s"] * 1e3, 2), "gather_ms_rank0": round(timing["gather_s"] * 1e3, 3),
           "gathered_bytes": int(len(samples) * nroi * 224 * 224 * 4),
           "workload": "comm.sharded_inference: shard_range -> DispRCNN3D per image (no collective) -> gather_predictions (tensor all_gather)"}
    if got is not None:
        out["gathered_images"] = len(got)
        out["disparity_shape"] = list(got[0].get_field("disparity").shape)
    del det, mB
    return out


def _time(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def train_step_extra(dev, model_a, world):
    """One TRAIN step of the disparity stage (reference: PSMNet.train() + PSMLoss, trainer.do_train): forward with per-GPU
    batch-stat BatchNorm, 3-head smooth-L1 loss, full backward (dgrad + MFMA wgrad) on the HIP engine, gradient sync (ONE flat
    20.9 MB all-reduce over RCCL/xGMI; a no-op at world size 1), optimizer step.  64 ROI pairs of Config A from the feature
    boundary, and 8 ROI crops of Config B through the 2D CNN as well.  Runs on EVERY rank (the all-reduce must be matched);
    the all-reduce time is measured with events around the sync."""
    from disprcnn_amd.utils import synth
    from disprcnn_amd.utils.loss_utils import PSMLoss
    from disprcnn_amd.utils.comm import GradientSync, all_ranks_ok
    mB, _ = build_model(dev, 48, -48, "B")
    tr = {}
    for tag, mdl, nroi in (("config_a_from_features_64roi", model_a, 64), ("config_b_full_psmnet_8roi", mB, 8)):
        mdl.train()
        sync = GradientSync(mdl.parameters())
        if tag.startswith("config_a"):
            fl_, fr_ = synth.synth_features(nroi, 32, 28, 28, tag="trainA")
            fl_, fr_ = fl_.to(dev), fr_.to(dev)
            tgt = synth.hash_uniform("trainA:t", (nroi, 112, 112), 0.0, 47.0).to(dev)
            fwd = lambda: mdl.forward_from_features(fl_, fr_, (112, 112))      # noqa: E731
            fl3 = FLOPS_PER_VOXEL_3D * 12 * 28 * 28 * nroi
        else:
            li, ri = synth.synth_images(nroi, 224, 224, tag="trainB")
            li, ri = li.to(dev), ri.to(dev)
            tgt = synth.hash_uniform("trainB:t", (nroi, 224, 224), -47.0, 47.0).to(dev)
            fwd = lambda: mdl({"left": li, "right": ri})                       # noqa: E731
            fl3 = FLOPS_PER_VOXEL_3D * 24 * 56 * 56 * nroi
        msk = torch.ones_like(tgt, dtype=torch.uint8)
        crit = PSMLoss()
        opt = torch.optim.SGD(mdl.parameters(), lr=1e-7, momentum=0.9)   # the parameters change every step: weights are re-packed
        sync_ms = []

        failed = []

        def train_step():
            # the forward / backward have no collective; the gradient sync has one.  A rank whose local part raised must not leave the
            # others waiting in the all-reduce: every rank agrees on an ok flag first (world 1: a local bool) and all skip the sync together
            loss, err = None, None
            try:
                sync.zero_grad()                 # .grad = zeroed views of the flat all-reduce buffer: the backward fills it directly
                loss = crit(fwd(), {"disparity": tgt, "mask": msk})
                loss.backward()
            except Exception as ex:              # noqa: BLE001
                err = ex
            if not all_ranks_ok(err is None, dev):
                failed.append(repr(err) if err is not None else "another rank failed")
                return loss
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sync(); e1.record()
            sync_ms.append((e0, e1))
            opt.step()
            return loss
        tt = _time(train_step, 3, 5)             # (the eager step is host-bound -- ~2,800 launches for Config B -- and noisy from box to box)
        if failed:
            tr[tag] = {"error": failed[0][:300]}
            mdl.eval()
            continue
        ar = sum(a_.elapsed_time(b_) for a_, b_ in sync_ms[-5:]) / 5
        tr[tag] = {"ms_per_step": round(tt * 1e3, 2), "roi_pairs_per_s_per_gpu": round(nroi / tt, 1),
                   "regressor_tflops_fwd_bwd": round(3 * fl3 / tt / 1e12, 2),
                   "grad_allreduce_ms": round(ar, 3) if world > 1 else 0.0}
        if world == 1:
            # the same step replayed from a HIP graph (utils/graph.py): the eager step is bounded by ~2,800 host-side launches
            from disprcnn_amd.utils.graph import GraphedStep

            def graph_step():
                opt.zero_grad(set_to_none=True)
                loss = crit(fwd(), {"disparity": tgt, "mask": msk})
                loss.backward()
                opt.step()
                return loss
            try:
                gs = GraphedStep(graph_step, warmup=2)
                tg = _time(gs, 2, 5)
                tr[tag]["hip_graph_ms_per_step"] = round(tg * 1e3, 2)
                tr[tag]["hip_graph_roi_pairs_per_s_per_gpu"] = round(nroi / tg, 1)
                del gs
            except Exception as e:                                   # noqa: BLE001 -- report, never hide the eager number
                tr[tag]["hip_graph_error"] = repr(e)[:200]
        mdl.eval()
    tr["workload"] = ("forward (batch-stat BN) + PSMLoss + backward + gradient sync (one flat fp32 all-reduce of 5.2 M parameters, "
                      f"world {world}) + SGD step; per-GPU batch; regressor FLOPs counted as 3x forward")
    del mB
    return tr


def rank0_extras(dev, extra):
    from disprcnn_amd import engine as E
    from disprcnn_amd.utils import synth
    # ---- a1 on its own ("a1: HBM bandwidth"): the MATERIALISED concat volume (what the train path builds; the eval path folds it
    # into the first layer's loads and never writes it).  Algorithmic bytes per ROI = 4 * (2*C*H'W' + 2*C*D'H'W') = 2,609,152 at Config A.
    n_cv = 512
    fl_cv, fr_cv = synth.synth_features(n_cv, 32, 28, 28, tag="benchcv")
    fl_cv, fr_cv = fl_cv.to(dev), fr_cv.to(dev)
    vol = E.Blocked(n_cv, 64, 12, 28, 28, 1, 1, 1, dev)
    t_cv = _time(lambda: E.cost_volume_blocked(fl_cv, fr_cv, vol, 0, 12, 0), 2, 10)
    by_cv = 4 * (2 * 32 * 28 * 28 + 2 * 32 * 12 * 28 * 28) * n_cv
    extra["cost_volume_standalone"] = {"GB_per_s_algorithmic": round(by_cv / t_cv / 1e9, 1), "us_per_launch": round(t_cv * 1e6, 1), "roi_pairs": n_cv,
                                       "frac_of_hbm_peak_8TBps": round(by_cv / t_cv / 8e12, 3), "roi_volumes_per_s": round(n_cv / t_cv, 1),
                                       "workload": "drc_cost_volume_blocked_fwd: dense features [N,32,28,28] x2 -> blocked fp32 volume [N,64,12,28,28] (zero halo not written)"}
    del vol, fl_cv, fr_cv
    # ---- Config B (224x224, D=96, full PSMNet incl. the 2D feature CNN), 16 ROI pairs per step
    mB, _ = build_model(dev, 48, -48, "B")
    l, r = synth.synth_images(16, 224, 224, tag="benchB")
    l, r = l.to(dev), r.to(dev)
    with torch.no_grad():
        tb = _time(lambda: mB((l, r)), 3, 5)
    fl_b = (FLOPS_PER_VOXEL_3D * 24 * 56 * 56 + 2 * FLOPS_2D_PER_IMAGE) * 16
    extra["config_b_full_psmnet"] = {"roi_pairs_per_s": round(16 / tb, 1), "ms_per_16_roi_image": round(tb * 1e3, 2),
                                     "stereo_pairs_per_s_16roi": round(1 / tb, 2),
                                     "direct_conv_equivalent_tflops": round(fl_b / tb / 1e12, 1),
                                     "workload": "16 ROI crops 224x224, D=96 (-48..48): 2D CNN + cost volume + 3D + soft-argmin"}
    # ---- BASELINE configs[3] shape (64 ROIs/image at 224x224x96).  (a) the default path: the regressor in split-f16 arithmetic (fp16 operands
    # on the f16 matrix cores, fp32-class error: 2.6e-4 px mean against the CPU fp32 oracle over all 64 ROIs, tests/test_hip_f16.py), the 2D CNN's stride-1 3x3 layers likewise (convs16r.hip)
    l64, r64 = synth.synth_images(64, 224, 224, tag="benchB64")
    l64, r64 = l64.to(dev), r64.to(dev)

    def parts(fn):
        """One instrumented pass: time and conv flops of the regressor's launches (3D kernels) and of the 2D CNN's."""
        E.TIMING = []
        with torch.no_grad():
            fn()
        torch.cuda.synchronize()
        is2d = lambda k: k.startswith(("convs16r", "wino2d", "conv2d", "pointwise", "stemconv"))      # noqa: E731 -- (convs16r: the split-f16 2D kernel)
        reg = [(f, a.elapsed_time(b) * 1e-3) for k, f, a, b in E.TIMING
               if not is2d(k) and k.startswith(("convs16", "conv16", "wino3d", "tapdirect", "downdirect_kernel<7", "downdirect_kernel<4", "deconv"))]
        cnn = [(f, a.elapsed_time(b) * 1e-3) for k, f, a, b in E.TIMING if is2d(k)]
        E.TIMING = None
        return sum(t for _, t in reg), sum(f for f, _ in reg), sum(t for _, t in cnn), sum(f for f, _ in cnn)
    with torch.no_grad():
        ts = _time(lambda: mB((l64, r64)), 3, 3)
    t_reg, f_reg, t_cnn, f_cnn = parts(lambda: mB((l64, r64)))
    extra["stress_64roi_224x224x96"] = {
        "roi_pairs_per_s": round(64 / ts, 1), "ms_per_64_roi_image": round(ts * 1e3, 2),
        "dtype": ("regressor and the stride-1 3x3 layers of the 2D CNN (89 % of its FLOPs): split-f16 (3 f16-MFMA products per fp32 product, f32 accumulate), "
                  "fp32-class error; the rest of the 2D CNN (stem, stride-2, 1x1, SPP, lastconv): f32 MFMA"),
        "regressor_ms": round(t_reg * 1e3, 2), "regressor_tflops_fp32_equivalent": round(f_reg / t_reg / 1e12, 1),
        "regressor_frac_of_f16_mfma_peak": round(3 * f_reg / t_reg / 1e12 / PEAK_F16_TFLOPS, 4),
        "cnn2d_ms": round(t_cnn * 1e3, 2), "cnn2d_direct_conv_equivalent_tflops": round(f_cnn / t_cnn / 1e12, 1),
        "note": "regressor_ms / cnn2d_ms are sums of per-launch HIP-event times of one instrumented pass (convolution launches only)"}
    # ---- (b) the same shape with the fp16-STORAGE regressor (half the activation bytes; one f16 product per fp32 product): fp16 cost volume + 3D
    # regressor with fp32 accumulation, fp32-class 2D CNN; error vs the CPU fp32 oracle measured over all 64 ROIs in tests/test_hip_f16.py
    # (5.1e-2 px mean on the sharp synthetic weights, 5.0e-3 on the tempered set)
    mB.regressor_storage = "f16"
    with torch.no_grad():
        out16 = mB((l64, r64))
        t16 = _time(lambda: mB((l64, r64)), 3, 3)
    t_reg16, f_reg16, _, _ = parts(lambda: mB((l64, r64)))
    mB.regressor_storage = "f32"
    with torch.no_grad():
        ref64 = mB((l64, r64))
        err16 = (out16 - ref64).abs().mean().item()
    del ref64
    # HBM bytes the fp16-storage regressor must move per ROI: every conv's input + output (+ residual) once, 2 bytes per value (279.1 MB
    # of fp32 activation traffic per ROI at Config B, halved) -- its kernels are priced against HBM as well as against the MFMA peak
    extra["stress_64roi_224x224x96_f16_storage"] = {"roi_pairs_per_s": round(64 / t16, 1), "ms_per_64_roi_image": round(t16 * 1e3, 2),
                                                    "regressor_ms": round(t_reg16 * 1e3, 2),
                                                    "regressor_direct_conv_equivalent_tflops": round(f_reg16 / t_reg16 / 1e12, 1),
                                                    "regressor_frac_of_f16_mfma_peak": round(f_reg16 / t_reg16 / 1e12 / PEAK_F16_TFLOPS, 4),
                                                    "regressor_algorithmic_hbm_GB_per_s": round(64 * 279.1e6 / 2 / t_reg16 / 1e9, 1),
                                                    "regressor_frac_of_hbm_peak_8TBps": round(64 * 279.1e6 / 2 / t_reg16 / 8e12, 3),
                                                    "mean_abs_err_px_vs_default_path": round(err16, 4),
                                                    "dtype": "f16 storage / f32 accumulate (v_mfma_f32_16x16x32_f16) for cost volume + 3D regressor; 2D CNN at fp32-class error (split-f16 / f32 MFMA, as in the default path)"}
    # (the opt-in all-fp16 mode, PSMNet.feature_storage = "f16", is not reported here: its error vs the fp32 path -- 0.34 px, tests/test_hip_f16.py --
    # is outside the bound a throughput figure may be quoted under)
    del l64, r64, out16
    # ---- BASELINE configs[1] -- one stereo pair 2x3x375x1242 through ResNet-50-FPN (2D stage trunk) plus the
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

    def pair_once():
        feats = bb(pair)
        out = det({"left": ImageList(pair[:1], [(Hi, Wi)]), "right": ImageList(pair[1:], [(Hi, Wi)])},
                  {"left": [BoxList(lb, (Wi, Hi))], "right": [BoxList(rb, (Wi, Hi))]})
        return feats, out

    def pair_step():        # trunk + disparity stage under ONE range guard (engine.one_guard: one read of the guard word per pair instead of two)
        return E.one_guard(pair_once, dev, what="KITTI pair (trunk + DispRCNN3D)")
    with torch.no_grad():
        tp = _time(pair_step, 3, 5)
        tbb = _time(lambda: bb(pair), 2, 5)
        # flops the pair's launches EXECUTE on the matrix cores: each plan's algorithmic conv flops x its kernel's ratio (Winograd
        # F(2x2x2,3x3x3) runs 64 multiplies where the direct form needs 216, F(2x2,3x3) 16 of 36)
        E.TIMING = []
        pair_step()
        torch.cuda.synchronize()
        # per launch: the flops it EXECUTES (Winograd F(2x2x2,3x3x3) runs 64 multiplies where the direct form needs 216, F(2x2,3x3) 16 of 36) and
        # the peak of the pipe it runs on: fp32 MFMA 157.3 TF, or -- the split-f16 kernels, three f16 products per fp32 product -- 2500 / 3 TF
        ratio = lambda k: 64.0 / 216.0 if k.startswith("wino3d") else 16.0 / 36.0 if k.startswith("wino2d") else 1.0      # noqa: E731
        peak_of = lambda k: PEAK_F16_TFLOPS / 3.0 if k.startswith("convs16") else PEAK_F32_TFLOPS                       # noqa: E731
        fl_exec = sum(f * ratio(k) for k, f, _, _ in E.TIMING)
        t_at_peak = sum(f * ratio(k) / (peak_of(k) * 1e12) for k, f, _, _ in E.TIMING)      # seconds the pair's conv launches would take at their pipes' peaks
        fl_timed = sum(f for _, f, _, _ in E.TIMING)
        E.TIMING = None
    fl_pair = FLOPS_BACKBONE_PAIR + fl_b
    extra["kitti_pair_r50fpn_plus_16roi"] = {
        "stereo_pairs_per_s": round(1.0 / tp, 2), "ms_per_pair": round(tp * 1e3, 2), "backbone_ms": round(tbb * 1e3, 2),
        "backbone_tflops": round(FLOPS_BACKBONE_PAIR / tbb / 1e12, 2),
        "roofline": {"bound": "mfma", "unit": "TFLOP/s",
                     "frac": round(t_at_peak / tp, 4),
                     "frac_is": "sum over the pair's conv launches of (executed flops / the peak of the pipe the kernel runs on) / wall time: fp32 MFMA "
                                "157.3 TF (trunk, 2D CNN; Winograd launches counted at their executed 16/36 or 64/216), split-f16 kernels 2500/3 TF",
                     "direct_conv_equivalent_tflops": round(fl_pair / tp / 1e12, 2),
                     "direct_equiv_over_f32_peak": round(fl_pair / tp / 1e12 / PEAK_F32_TFLOPS, 4),
                     "executed_flops_per_pair": fl_exec, "conv_flops_of_the_timed_launches": fl_timed,
                     "flops_per_pair": fl_pair,
                     "note": ("direct_conv_equivalent_* = SURVEY 8a/8d's algorithmic conv flops (backbone 250.3 G + 16 x (2 x 22.19 G 2D CNN + 48.51 G "
                              "regressor)) / wall time: a rate, not a roofline fraction; per-kernel MFMA-busy and HBM bytes: profiles/r5_pair_backbone_*.md, "
                              "r5_configB_*.md")},
        "workload": "BASELINE configs[1]: R-50-FPN on 2x3x375x1242 (250.3 GFLOP/pair, SURVEY a12) + 16 ROIs: pairing, ROIAlign crops, PSMNet 224^2 D=96"}
    del bb, det, mB
    # ---- the 2D stage in front of the path: DispRCNN = R-50-FPN trunk + Stereo RPN + stereo box head + mask head on the pair
    try:
        from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
        d2 = DispRCNN(default_cfg_2d("R-50-FPN"))
        sd2 = d2.state_dict()
        heads2 = synth.synth_det_state({k: v for k, v in sd2.items() if not k.startswith("backbone.")}, gain={
            "rpn." + k if k.startswith("head.") else "roi_heads." + k: v for k, v in synth.DET_GAIN.items()})
        d2.load_state_dict({**{"backbone." + k: v for k, v in bsd.items()}, **heads2}, strict=True)
        d2 = d2.to(dev).eval()
        with torch.no_grad():
            out2 = d2({"left": pair[:1], "right": pair[1:]})
            t2d = _time(lambda: d2({"left": pair[:1], "right": pair[1:]}), 2, 5)
        extra["stereo_2d_stage_r50fpn_pair"] = {
            "ms_per_pair": round(t2d * 1e3, 2), "pairs_per_s": round(1.0 / t2d, 2), "heads_ms": round((t2d - tbb) * 1e3, 2),
            "detections": int(len(out2["left"][0])),
            "workload": ("DispRCNN on 2x3x375x1242 with synthetic weights: R-50-FPN trunk, Stereo RPN over 5 levels (PRE_NMS 6000, POST_NMS 300), "
                         "stereo box head (2 x 7x7 ROIAlign, FC 25088-2048-2048), NMS, mask head (14x14 ROIAlign, 4 x conv3x3, deconv, logits)")}
        del d2, out2
    except Exception as ex:                                   # noqa: BLE001 -- reported, never hides the other extras
        extra["stereo_2d_stage_error"] = repr(ex)[:300]
    # ---- post-processing: 16 images 375x1242, 16 ROI maps 224x224 each -> full-image disparity maps
    from disprcnn_amd import ops as _ops
    gpp = torch.Generator().manual_seed(0)
    nimg, nr, ih, iw = 16, 16, 375, 1242
    x1 = torch.rand(nimg * nr, generator=gpp) * (iw - 260); y1 = torch.rand(nimg * nr, generator=gpp) * (ih - 180)
    lbp = torch.stack([x1, y1, x1 + 60 + torch.rand(nimg * nr, generator=gpp) * 190, y1 + 40 + torch.rand(nimg * nr, generator=gpp) * 130], 1)
    rbp = lbp.clone(); rbp[:, 0] = (lbp[:, 0] - 30).clamp(min=0); rbp[:, 2] = lbp[:, 2] - 25
    dpp = (torch.rand(nimg * nr, 224, 224, generator=gpp) * 96 - 48).to(dev)
    b6 = _ops.integer_roi_boxes(lbp.to(dev), rbp.to(dev))
    tp = _time(lambda: _ops.disparity_paste(dpp, b6, [nr] * nimg, ih, iw), 3, 20)
    pbytes = nimg * ih * iw * 4 + dpp.numel() * 4
    extra["post_process_16img_x16roi"] = {"us_per_call": round(tp * 1e6, 1), "images_per_s": round(nimg / tp, 1),
                                          "algorithmic_GB_per_s": round(pbytes / tp / 1e9, 1),
                                          "workload": "drc_disparity_paste_fwd: 256 ROI maps 224^2 -> 16 maps 375x1242 (one launch; bytes = maps read once + outputs written once)"}


if __name__ == "__main__":
    main()
