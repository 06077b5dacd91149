#!/usr/bin/env python
"""bench.py - YOLOv3-416 COCO-80 end-to-end (conv + decode + NMS) images/sec on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run)

A step = one pass of the hot path over one batch of 64 synthetic 416x416 images per GPU that are
already resident in HBM: yolov3.forward (bf16 MFMA conv stack) -> predict -> conf*prob -> per-class NMS
(gpu_nms defaults 50 / 0.5 / 0.5; the three post-processing steps run fused, yolo_detect_post).  Images are independent, so ranks share nothing
(weak scaling, no collective in the timed region).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
F32_PEAK_TFLOPS = 157.3
# HBM-side bytes per launch of a kernel instantiation, from the committed PMC passes (profiles/, see DESIGN.md
# section 5): 2 x FETCH_SIZE + WRITE_SIZE averaged over the launches of one bench step (FETCH_SIZE doubled as
# MI355X_MICROARCH.md "HBM" prescribes for gfx950).  Keyed by (kernel, batch, size); None when not measured.
TRAFFIC_BYTES_PER_LAUNCH = {
    # profiles/r1_bench_pmc_traffic_v5_lanes1.txt (one launch per layer over the full batch, as the roofline is measured):
    # halo kernel, two patch buffers: fetch 138.70 MB (2 x FETCH_SIZE) + write 35.69 MB
    ("yolo::conv3x3_halo_kernel<true, 128, 128, 4, 2, false, 32>", 64, 416): 174.39e6,
    ("yolo::conv3x3_halo_kernel<true, 128, 128, 4, 2, true, 32>", 64, 416): 245.34e6,
    ("yolo::conv_igemm_kernel<true, 128, 128, 4, 2, 2>", 64, 416): 119.59e6,
}


def conv_flops_per_image(info, H, W):
    """2*MAC per conv: output resolution follows the strides along the plan."""
    # resolution of every conv's OUTPUT: replay the architecture (stride-2 convs halve; the two
    # upsample branches return to the route resolution; heads keep theirs).
    flops = []
    h, w = H, W
    res_at = {}
    for i, (cin, cout, k, s) in enumerate(info):
        if i == 59:            # inter1 conv consumes conv 56's output (13x13)
            h, w = H // 32, W // 32
        elif i == 60:          # yolo_block on concat1 (26x26)
            h, w = H // 16, W // 16
        elif i == 67:
            h, w = H // 16, W // 16
        elif i == 68:
            h, w = H // 8, W // 8
        h, w = h // s, w // s
        flops.append(2.0 * h * w * cout * cin * k * k)
    return flops


def cpu_baseline(n_images, threads):
    """The oracle (CPU restatement of the reference: TF is not installable offline) on host cores."""
    from oracle import yolo_oracle as O
    torch.set_num_threads(threads)
    specs = O.yolov3_conv_specs(80)
    params = O.split_darknet_stream(specs, O.synth_darknet_stream(specs, seed=1234))
    rng = np.random.default_rng(0)
    img = rng.random((1, 416, 416, 3), dtype=np.float32)
    O.detect(params, O.YOLO_ANCHORS, img, max_boxes=50, score_thresh=0.5, iou_thresh=0.5)     # warm
    t0 = time.perf_counter()
    for i in range(n_images):
        img = rng.random((1, 416, 416, 3), dtype=np.float32)
        O.detect(params, O.YOLO_ANCHORS, img, max_boxes=50, score_thresh=0.5, iou_thresh=0.5)
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d images, batch 1, 416x416, oracle forward+predict+gpu_nms(50,0.5,0.5), torch-CPU fp32 convs "
                      "(CPU restatement of the reference; TensorFlow not installable offline)" % n_images}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=32, help="images of the CPU baseline sample (about 10 s of host work)")
    ap.add_argument("--layer-table", default="", help="write the per-layer timing table to this file")
    ap.add_argument("--lanes", type=int, default=0, help="sub-batches run concurrently on separate streams inside the library "
                                                         "(YOLO_LANES); 0 = library default (2 from batch 12 up, 3 from batch 48 up)")
    ap.add_argument("--h2d", action="store_true", help="also report the rate with the batch copied from pinned host memory "
                                                       "inside every step (extra key, never `value`)")
    args = ap.parse_args()

    if args.lanes > 0:
        os.environ["YOLO_LANES"] = str(args.lanes)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d bench.py ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))

    from yolo_amd.model import yolov3
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    from yolo_amd.weights import synth_darknet_stream

    B, S = args.batch, args.size
    stream = synth_darknet_stream(80, seed=1234)
    net = yolov3(80, DEFAULT_ANCHORS, dtype=args.dtype).load_weight_stream(stream)
    rng = np.random.default_rng(1 + rank)
    images = torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda()

    def step():
        # forward -> predict -> conf*prob -> per-class NMS (gpu_nms defaults), the last three as one library call
        return net.detect(images, 50, 0.5, 0.5)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ndet = int(out[4].sum().item())

    # ---- roofline of the dominant kernel: per-layer HIP events recorded by the library on the launch stream
    # (yolo_net_forward_timed), layers grouped by the kernel INSTANTIATION the library launched for them ----
    reps = 5
    acc = None
    for _ in range(reps):
        _, ms, info = net.forward_timed(images)
        acc = ms.astype(np.float64) if acc is None else acc + ms
    layer_ms = acc / reps
    kernels = net.layer_kernels
    flops = conv_flops_per_image(info, S, S)
    groups = {}
    for i, name in enumerate(kernels):
        g = groups.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0})
        g["launches"] += 1
        g["ms"] += float(layer_ms[i])
        g["flops"] += flops[i] * B
    dom_name = max(groups, key=lambda n: groups[n]["ms"])
    dom = groups[dom_name]
    peak = BF16_PEAK_TFLOPS if args.dtype == "bf16" else F32_PEAK_TFLOPS
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    traffic = TRAFFIC_BYTES_PER_LAUNCH.get((dom_name, B, S))
    lanes = int(os.environ.get("YOLO_LANES", "0")) or (3 if B >= 48 else 2 if B >= 12 else 1)
    roofline = {"bound": "mfma", "kernel": dom_name,
                "measured": "HIP events around every layer of one launch per layer over the full batch (yolo_net_forward_timed); "
                            "the timed steps run the same kernels as %d concurrent lane(s) of ~%d images" % (lanes, B // lanes),
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic, "launches_per_step": dom["launches"],
                "avg_launch_ms": round(dom["ms"] / dom["launches"], 5),
                "flops_per_launch_avg": dom["flops"] / dom["launches"],
                "share_of_conv_stack_ms": round(dom["ms"] / float(layer_ms.sum()), 3),
                "conv_stack_ms": round(float(layer_ms.sum()), 4),
                "conv_stack_tflops": round(sum(flops) * B / (float(layer_ms.sum()) * 1e-3) / 1e12, 2),
                "kernels": [{"kernel": n, "launches": g["launches"], "ms": round(g["ms"], 4),
                             "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 1)}
                            for n, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])]}
    if args.layer_table and rank == 0:
        with open(args.layer_table, "w") as f:
            f.write("# layer cin cout k stride ms GFLOP TFLOP/s   (batch %d, %dx%d, %s)\n" % (B, S, S, args.dtype))
            for i, ((cin, cout, k, s), ms_i) in enumerate(zip(info, layer_ms)):
                f.write("%2d %4d %4d %d %d %8.4f %9.3f %8.2f  %s\n" % (i, cin, cout, k, s, ms_i, flops[i] * B / 1e9,
                                                                       flops[i] * B / (ms_i * 1e-3) / 1e12, kernels[i]))

    # ---- post-processing alone (decode + conf*prob + score mask + per-class NMS, fused: yolo_detect_post) on the feature
    # maps of the last forward: HBM-bound by SURVEY.md 8(d); algorithmic bytes per image = decode 7.24 MB (read 3 fms,
    # write boxes + confs + probs) + NMS 3.58 MB (read boxes + scores) at 416 - what the separate reference calls move
    fms = net.forward(images)
    for _ in range(3):
        net.postprocess(fms, 50, 0.5, 0.5)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        net.postprocess(fms, 50, 0.5, 0.5)
    ev1.record()
    torch.cuda.synchronize()
    post_ms = ev0.elapsed_time(ev1) / 20
    n_boxes = 3 * ((S // 32) ** 2 + (S // 16) ** 2 + (S // 8) ** 2)
    post_bytes = B * (2.0 * n_boxes * 85 * 4 + n_boxes * 84 * 4)
    fused_bytes = B * (n_boxes * 85 * 4 + n_boxes * 16)            # what the fused call must move: read fms, write boxes
    post = {"bound": "hbm", "ms": round(post_ms, 4), "peak": 8000.0, "unit": "GB/s",
            "achieved": round(post_bytes / (post_ms * 1e-3) / 1e9, 1), "frac": round(post_bytes / (post_ms * 1e-3) / 8e12, 4),
            "achieved_fused_bytes": round(fused_bytes / (post_ms * 1e-3) / 1e9, 1),
            "what": "yolo_detect_post on %d images (%d boxes x 80 classes each, thresholds 50 / 0.5 / 0.5), HIP events on the launch stream; "
                    "`achieved` prices the algorithmic bytes of the reference's separate predict + gpu_nms calls (10.82 MB/img), "
                    "`achieved_fused_bytes` the 3.79 MB/img the fused call really has to move" % (B, n_boxes)}

    h2d = None
    if args.h2d:
        from yolo_amd.preprocess import preprocess
        host = images.cpu().pin_memory()
        host_u8 = torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).pin_memory()

        def timed(fn):
            for _ in range(2):
                fn()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            barrier()
            return (time.perf_counter() - t0) / args.steps

        dth = timed(lambda: net.detect(host.to("cuda", non_blocking=True), 50, 0.5, 0.5))
        dtu = timed(lambda: net.detect(preprocess(host_u8.to("cuda", non_blocking=True), (S, S)), 50, 0.5, 0.5))
        from yolo_amd.pipeline import DetectPipeline
        pipe = DetectPipeline(net, (S, S), 50, 0.5, 0.5)
        for _ in pipe.run([host_u8] * 3):
            pass
        barrier()
        t0 = time.perf_counter()
        for _ in pipe.run([host_u8] * args.steps):
            pass
        barrier()
        dtp = (time.perf_counter() - t0) / args.steps
        h2d = {"images_per_sec": round(B / dth, 2), "ms_per_step": round(dth * 1e3, 4),
               "what": "same step with the %d MB f32 batch copied from pinned host memory each step (serial copy + compute)"
                       % (images.numel() * 4 // 1000000),
               "uint8_upload": {"images_per_sec": round(B / dtu, 2), "ms_per_step": round(dtu * 1e3, 4),
                                "what": "%d MB uint8 BGR batch copied from pinned host memory, then yolo_preprocess "
                                        "(resize / BGR2RGB / /255 on the GPU) + the step" % (host_u8.numel() // 1000000)},
               "uint8_pipelined": {"images_per_sec": round(B / dtp, 2), "ms_per_step": round(dtp * 1e3, 4),
                                   "what": "the same with yolo_amd.pipeline.DetectPipeline: the copy of batch i+1 runs on a copy "
                                           "stream under preprocess + detect of batch i"}}

    if rank == 0:
        res = {
            "metric": "YOLOv3-416 images/sec end-to-end (conv+decode+NMS)",
            "value": round(world * B * args.steps / dt, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "YOLOv3-%d COCO-80, batch %d per GPU, %s conv stack + fp32 decode + per-class NMS "
                                   "(max_boxes 50, score 0.5, iou 0.5), synthetic images and seeded random-init Darknet-order weights"
                                   % (S, B, args.dtype),
                       "global_batch": world * B, "detections_last_step": ndet, "parallelism": "image-sharded dp%d" % world,
                       "lanes_per_gpu": lanes},
            "roofline": roofline,
        }
        res["post_processing"] = post
        if h2d is not None:
            res["pcie_inclusive"] = h2d
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_images, min(16, os.cpu_count() or 1))
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
