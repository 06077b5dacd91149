#!/usr/bin/env python
"""bench.py - YOLOv3-416 COCO-80 end-to-end (conv + decode + NMS) images/sec on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run)

A step = one pass of the hot path over one batch of 64 synthetic 416x416 images per GPU that are
already resident in HBM: yolov3.forward (bf16 MFMA conv stack) -> predict -> conf*prob -> per-class NMS
(gpu_nms defaults 50 / 0.5 / 0.5; the three post-processing steps run fused, yolo_detect_post).  Images are independent, so ranks share nothing
(weak scaling, no collective in the timed region).  Rank 0 prints ONE JSON line.

Beside the headline value the line carries (all measured in this run):
  roofline          dominant conv kernel against the dense bf16 MFMA peak (HIP events on the launch stream)
  fp32              the same step in the reference's own precision (parity path), short pass
  sharded_512       BASELINE config 3: a 512-image batch cut with shard_bounds over the ranks, local steps of 64
  post_processing   yolo_detect_post alone against HBM;  decode: API-level yolo_decode (predict + scores) against HBM
  cpu_baseline      the oracle (CPU restatement of the reference) on the host cores, bounded sample
"""
import argparse
import hashlib
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
HBM_PEAK_GBS = 8000.0
TRAFFIC_TABLE = os.path.join(ROOT, "profiles", "traffic_table.json")


def kernel_sources_sha():
    """Hash of the kernel sources: the PMC traffic table is only valid for the sources it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "yolo_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def traffic_for(kernel, B, S):
    """HBM-side bytes per launch of `kernel` from the PMC passes scripts/gpu_profile.sh wrote (2 x FETCH_SIZE + WRITE_SIZE,
    MI355X_MICROARCH.md "HBM"); None (with the reason) when the table is missing or was measured on other sources."""
    if not os.path.exists(TRAFFIC_TABLE):
        return None, "no profiles/traffic_table.json"
    t = json.load(open(TRAFFIC_TABLE))
    if t.get("sources_sha") != kernel_sources_sha():
        return None, "traffic table measured on other kernel sources (%s): re-run scripts/gpu_profile.sh" % t.get("sources_sha")
    if (t.get("batch"), t.get("size")) != (B, S):
        return None, "traffic table measured at batch %s / %s" % (t.get("batch"), t.get("size"))
    for name, v in t.get("kernels", {}).items():
        if kernel in name:
            return v["total_bytes"], "profiles/traffic_table.json (%s)" % t.get("command", "")
    return None, "kernel not in the traffic table"


def post_processing_counter_bytes(B, S):
    """Sum of the PMC bytes per launch of the three post-processing kernels (decode_sparse / nms_select / nms_gather)."""
    tot = 0.0
    for k in ("decode_sparse_kernel", "nms_select_kernel", "nms_gather_kernel"):
        b, src = traffic_for(k, B, S)
        if b is None:
            return None, src
        tot += b
    return tot, src


def conv_flops_per_image(info, H, W):
    """2*MAC per conv: output resolution follows the strides along the plan."""
    flops = []
    h, w = H, W
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
    """The oracle (CPU restatement of the reference: TF is not installable offline) on host cores: batch 1 with the
    stages timed apart (conv stack / decode / per-class NMS), then a batch-8 figure."""
    from oracle import yolo_oracle as O
    torch.set_num_threads(threads)
    specs = O.yolov3_conv_specs(80)
    params = O.split_darknet_stream(specs, O.synth_darknet_stream(specs, seed=1234))
    rng = np.random.default_rng(0)

    def one(img, acc):
        t0 = time.perf_counter()
        fms = O.forward(params, img)
        t1 = time.perf_counter()
        b, c, p = O.predict(fms, O.YOLO_ANCHORS, img.shape[1:3])
        s = (c * p).astype(np.float32)
        t2 = time.perf_counter()
        for i in range(img.shape[0]):
            O.gpu_nms(b[i], s[i], 80, 50, 0.5, 0.5)
        t3 = time.perf_counter()
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2

    one(rng.random((1, 416, 416, 3), dtype=np.float32), [0.0, 0.0, 0.0])      # warm
    st1 = [0.0, 0.0, 0.0]
    for _ in range(n_images):
        one(rng.random((1, 416, 416, 3), dtype=np.float32), st1)
    dt1 = sum(st1)
    st8 = [0.0, 0.0, 0.0]
    n8 = max(1, n_images // 16)
    for _ in range(n8):
        one(rng.random((8, 416, 416, 3), dtype=np.float32), st8)
    dt8 = sum(st8)
    return {"value": round(n_images / dt1, 3), "unit": "images/sec", "cores": threads, "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": "%d images at batch 1 + %d at batch 8, 416x416, oracle forward + predict + conf*prob + gpu_nms(50,0.5,0.5), "
                      "torch-CPU fp32 convs on %d threads (CPU restatement of the reference; TensorFlow not installable offline)"
                      % (n_images, 8 * n8, threads),
            "batch1_ms_per_image": {"conv": round(st1[0] / n_images * 1e3, 2), "decode": round(st1[1] / n_images * 1e3, 2),
                                    "nms": round(st1[2] / n_images * 1e3, 2)},
            "batch8": {"images_per_sec": round(8 * n8 / dt8, 3),
                       "ms_per_image": {"conv": round(st8[0] / (8 * n8) * 1e3, 2), "decode": round(st8[1] / (8 * n8) * 1e3, 2),
                                        "nms": round(st8[2] / (8 * n8) * 1e3, 2)}}}


def dominant_kernel(net, images, dtype, B, S, lanes, reps=5):
    """Per-layer HIP events recorded by the library on the launch stream (yolo_net_forward_timed), layers grouped by the
    kernel INSTANTIATION the library launched for them; the group with the most time is the dominant kernel."""
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
    peak = BF16_PEAK_TFLOPS if dtype == "bf16" else F32_PEAK_TFLOPS
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    traffic, traffic_src = traffic_for(dom_name, B, S) if dtype == "bf16" else (None, "not measured for fp32")
    roofline = {"bound": "mfma", "kernel": dom_name,
                "measured": "HIP events around every layer of one launch per layer over the full batch (yolo_net_forward_timed); "
                            "the timed steps run the same kernels as %d concurrent lane(s) of ~%d images" % (lanes, B // lanes),
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": dom["launches"],
                "avg_launch_ms": round(dom["ms"] / dom["launches"], 5),
                "flops_per_launch_avg": dom["flops"] / dom["launches"],
                "share_of_conv_stack_ms": round(dom["ms"] / float(layer_ms.sum()), 3),
                "conv_stack_ms": round(float(layer_ms.sum()), 4),
                "conv_stack_tflops": round(sum(flops) * B / (float(layer_ms.sum()) * 1e-3) / 1e12, 2),
                "kernels": [{"kernel": n, "launches": g["launches"], "ms": round(g["ms"], 4),
                             "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 1)}
                            for n, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])]}
    return roofline, layer_ms, info, flops, kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--global-batch", type=int, default=512,
                    help="BASELINE config 3 (extra record `sharded_512`): images of ONE batch cut over the ranks with shard_bounds, "
                         "each rank runs its shard in local steps of --batch; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / sharded / decode side records (profiling runs)")
    ap.add_argument("--cpu-images", type=int, default=16, help="batch-1 images of the CPU baseline sample (+ the same number at batch 8)")
    ap.add_argument("--layer-table", default="", help="write the per-layer timing table to this file")
    ap.add_argument("--lanes", type=int, default=0, help="sub-batches run concurrently on separate streams inside the library "
                                                         "(YOLO_LANES); 0 = library default (2 from 16 images of 416 x 416 up, 3 from 56 up)")
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
    # YOLO_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks then share
    # devices and the three scalar reductions go through host tensors); the driver's runs use RCCL, one rank per GPU
    backend = os.environ.get("YOLO_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)

    from yolo_amd.model import yolov3
    from yolo_amd.shard import shard_bounds
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    from yolo_amd.weights import synth_darknet_stream

    B, S = args.batch, args.size
    stream = synth_darknet_stream(80, seed=1234)
    net = yolov3(80, DEFAULT_ANCHORS, dtype=args.dtype).load_weight_stream(stream)
    rng = np.random.default_rng(1 + rank)
    images = torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup, spread=None):
        """W untimed + exactly K timed calls between barrier + synchronize; MAX over ranks.  `spread` (a list) receives the
        device time of every timed step of this rank: events recorded on the launch stream between the steps, no extra
        synchronisation (the steps stay queued back to back exactly as without them)."""
        out = None
        for _ in range(warmup):
            out = fn()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if spread is not None else None
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            if evs is not None:
                evs[i].record()
            out = fn()
        if evs is not None:
            evs[steps].record()
        barrier()
        dt = time.perf_counter() - t0
        if evs is not None:
            spread.extend(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    # forward -> predict -> conf*prob -> per-class NMS (gpu_nms defaults), the last three fused, as one library call
    step_ms = []
    dt, out = timed(lambda: net.detect(images, 50, 0.5, 0.5), args.steps, args.warmup, spread=step_ms)
    ndet = int(out[4].sum().item())
    work = B * S * S // (416 * 416)                        # the library's rule (net.hip lane_count), including the clamps
    lanes = int(os.environ.get("YOLO_LANES", "0")) or (2 if work >= 90 else 3 if work >= 56 else 2 if work >= 26 else 1)
    lanes = max(1, min(lanes, 8, B))

    roofline, layer_ms, info, flops, kernels = dominant_kernel(net, images, args.dtype, B, S, lanes)
    if args.layer_table and rank == 0:
        with open(args.layer_table, "w") as f:
            f.write("# layer cin cout k stride ms GFLOP TFLOP/s   (batch %d, %dx%d, %s)\n" % (B, S, S, args.dtype))
            for i, ((cin, cout, k, s), ms_i) in enumerate(zip(info, layer_ms)):
                f.write("%2d %4d %4d %d %d %8.4f %9.3f %8.2f  %s\n" % (i, cin, cout, k, s, ms_i, flops[i] * B / 1e9,
                                                                       flops[i] * B / (ms_i * 1e-3) / 1e12, kernels[i]))

    def event_ms(fn, reps=20, warm=3):
        for _ in range(warm):
            fn()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps

    # ---- post-processing alone (decode + conf*prob + score mask + per-class NMS, fused: yolo_detect_post) on the feature
    # maps of the last forward.  HBM-bound by SURVEY.md 8(d); `frac` prices the bytes the fused call has to move (read the
    # three feature maps, write the decoded boxes); the bytes the reference's separate predict + gpu_nms calls would move
    # (7.24 + 3.58 MB/img) are a named extra, not a roofline fraction.
    fms = net.forward(images)
    n_boxes = 3 * ((S // 32) ** 2 + (S // 16) ** 2 + (S // 8) ** 2)
    post_ms = event_ms(lambda: net.postprocess(fms, 50, 0.5, 0.5))
    fused_bytes = B * (n_boxes * 85 * 4 + n_boxes * 16)
    ref_bytes = B * (2.0 * n_boxes * 85 * 4 + n_boxes * 84 * 4)
    counter_bytes, counter_src = post_processing_counter_bytes(B, S)
    moved = counter_bytes if counter_bytes is not None else fused_bytes
    post = {"bound": "hbm", "ms": round(post_ms, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": round(moved / (post_ms * 1e-3) / 1e9, 1), "frac": round(moved / (post_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "counter_bytes": counter_bytes, "counter_source": counter_src, "algorithmic_bytes": fused_bytes,
            "algorithmic_rate": round(fused_bytes / (post_ms * 1e-3) / 1e9, 1),
            "reference_bytes_rate": round(ref_bytes / (post_ms * 1e-3) / 1e9, 1),
            "what": "yolo_detect_post on %d images (%d boxes x 80 classes each, thresholds 50 / 0.5 / 0.5), events on the launch stream; "
                    "`achieved` / `frac` = the bytes the PMC counters saw the three kernels move per call (`counter_bytes`: the box-level "
                    "conf gate skips the class logits of most boxes) over the time, or the algorithmic bytes when the counter table does "
                    "not fit the sources; `algorithmic_bytes` = the %.2f MB/img a fused call without the gate would move; "
                    "`reference_bytes_rate` prices the 10.82 MB/img of the reference's separate predict + gpu_nms calls and is not a "
                    "roofline fraction; the stage is three dependent launches + a memset and latency-bound, not HBM-bound"
                    % (B, n_boxes, fused_bytes / B / 1e6)}

    side = {}
    if not args.no_extras:
        # ---- API-level predict (+ scores): yolo_decode, algorithmic bytes SURVEY.md 8(d) = 7.24 + 3.41 MB/img at 416
        dec_ms = event_ms(lambda: net.predict(fms, return_scores=True))
        dec_bytes = B * (2.0 * n_boxes * 85 * 4 + n_boxes * 80 * 4)
        side["decode"] = {"bound": "hbm", "ms": round(dec_ms, 4), "achieved": round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dec_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "what": "yolo_decode = predict + conf*prob on %d images: reads 3 feature maps, writes boxes, confs, probs, scores "
                                  "(%.2f MB/img algorithmic), events on the launch stream; includes the output allocation of the mirror" %
                                  (B, dec_bytes / B / 1e6)}

        # ---- BASELINE config 3: ONE batch of --global-batch images, contiguous shards (shard_bounds), local steps of --batch
        if args.global_batch > 0:
            G = args.global_batch
            lo, hi = shard_bounds(G, rank, world)
            n_local = hi - lo
            shard = torch.empty((n_local, S, S, 3), dtype=torch.float32, device="cuda")
            for j in range(0, n_local, 64):            # per-image seeds: the global batch does not depend on the split
                k = min(64, n_local - j)
                blk = np.stack([np.random.default_rng([2, lo + j + i]).random((S, S, 3), dtype=np.float32) for i in range(k)])
                shard[j:j + k] = torch.from_numpy(blk).cuda()

            def shard_pass():
                outs = []
                for j in range(0, n_local, B):
                    outs.append(net.detect(shard[j:j + B], 50, 0.5, 0.5))
                return outs
            reps = max(2, min(4, args.steps))
            dts, outs = timed(shard_pass, reps, 1)
            nd = sum(int(o[4].sum().item()) for o in outs)
            if dist is not None:
                t = torch.tensor([nd], dtype=torch.int64, device=red_dev)
                dist.all_reduce(t)
                nd = int(t.item())
            side["sharded_512"] = {"global_batch": G, "n_gpus": world, "images_per_sec": round(G * reps / dts, 2),
                                   "ms_per_global_batch": round(dts / reps * 1e3, 3), "local_images_rank0": n_local,
                                   "local_steps_rank0": (n_local + B - 1) // B, "scaling": "strong", "detections": nd,
                                   "what": "BASELINE config 3: shard_bounds(%d, rank, %d), every rank runs its contiguous shard in local "
                                           "steps of %d, no collective in the timed region; efficiency vs the 1-GPU figure of this key is "
                                           "the driver's to compute" % (G, world, B)}
            del shard

        # ---- the reference's own precision (fp32 parity path), short pass: value + dominant kernel against the f32 matrix peak
        if args.dtype == "bf16":
            del fms
            net32 = yolov3(80, DEFAULT_ANCHORS, dtype="fp32").load_weight_stream(stream)
            dt32, _ = timed(lambda: net32.detect(images, 50, 0.5, 0.5), 4, 1)
            r32, _, _, _, _ = dominant_kernel(net32, images, "fp32", B, S, lanes, reps=2)
            side["fp32"] = {"images_per_sec": round(world * B * 4 / dt32, 2), "ms_per_step": round(dt32 / 4 * 1e3, 3), "steps": 4,
                            "dominant_kernel": r32["kernel"], "achieved": r32["achieved"], "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": r32["frac"], "conv_stack_ms": r32["conv_stack_ms"], "conv_stack_tflops": r32["conv_stack_tflops"],
                            "what": "same step with the conv stack on exact-f32 MFMAs (v_mfma_f32_32x32x2_f32): the parity path"}
            del net32

    # ---- latency at batch 1 (the reference's own published figure is a batch-1 latency): default plan vs the opt-in
    # split-K small-batch plan (yolo_net_set_small_batch_plan), same detect call, events on the launch stream
    if not args.no_extras and world == 1 and args.dtype == "bf16":
        one = images[:1].contiguous()
        net_sb = yolov3(80, DEFAULT_ANCHORS, dtype="bf16", small_batch_plan=True).load_weight_stream(stream)
        d_ms = event_ms(lambda: net.detect(one, 50, 0.5, 0.5), reps=100, warm=10)
        s_ms = event_ms(lambda: net_sb.detect(one, 50, 0.5, 0.5), reps=100, warm=10)
        sf_ms = event_ms(lambda: net_sb.forward(one), reps=100, warm=10)
        side["small_batch"] = {"batch": 1, "default_plan_ms": round(d_ms, 4), "small_batch_plan_ms": round(s_ms, 4),
                               "small_batch_plan_forward_ms": round(sf_ms, 4), "images_per_sec": round(1e3 / s_ms, 1),
                               "what": "one %dx%d image per detect call, 100 calls back to back; small_batch_plan: launches with fewer than "
                                       "128 tiles split their K loop over up to 512 workgroups (f32 partial tiles + an ordered reduce launch); "
                                       "opt-in per handle, results within the bf16 tolerance of the default plan, not bit-identical" % (S, S)}
        del net_sb

    h2d = None
    if args.h2d:
        from yolo_amd.preprocess import preprocess
        host = images.cpu().pin_memory()
        host_u8 = torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).pin_memory()
        dth = timed(lambda: net.detect(host.to("cuda", non_blocking=True), 50, 0.5, 0.5), args.steps, 2)[0] / args.steps
        dtu = timed(lambda: net.detect(preprocess(host_u8.to("cuda", non_blocking=True), (S, S)), 50, 0.5, 0.5), args.steps, 2)[0] / args.steps
        from yolo_amd.pipeline import DetectPipeline
        pipe = DetectPipeline(net, (S, S), 50, 0.5, 0.5)

        def piped(n):
            for _ in pipe.run([host_u8] * n):
                pass
        piped(3)
        barrier()
        t0 = time.perf_counter()
        piped(args.steps)
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
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "step_ms_min": round(min(step_ms), 4), "step_ms_median": round(float(np.median(step_ms)), 4), "step_ms_max": round(max(step_ms), 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "YOLOv3-%d COCO-80, batch %d per GPU, %s conv stack + fp32 decode + per-class NMS "
                                   "(max_boxes 50, score 0.5, iou 0.5), synthetic images and seeded random-init Darknet-order weights"
                                   % (S, B, args.dtype),
                       "global_batch": world * B, "detections_last_step": ndet, "parallelism": "image-sharded dp%d" % world,
                       "lanes_per_gpu": lanes},
            "roofline": roofline,
        }
        res["post_processing"] = post
        res.update(side)
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
