#!/usr/bin/env python
"""Secondary configurations of SURVEY.md 8(d), one JSON line each (not the driver's bench line):

  config 4: YOLOv2-416 VOC-20, batch 64: build_networks (bf16 conv stack) + decode + class-agnostic NMS (0.3 / 0.5)
  config 5: NMS stress, 608 x 608 decode shape: 8 images x 22743 boxes x 80 classes, score_thresh 0.001, iou 0.5, 50 per class,
            plain and "clustered" (200 seed boxes x jitter N(0, 4 px)) box sets; and the whole chain (forward + predict + NMS)
            at 608 x 608, batch 8, with that threshold

python scripts/bench_secondary.py [--steps 20] [--warmup 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def synth_v2_variables(seed=4321):
    """He-initialised weights, BN statistics near identity, in the variable order yolo2_net_create takes."""
    from yolo_amd.yolo_v2.yolo.yolo_v2 import conv_shapes
    rng = np.random.default_rng(seed)
    parts = []
    for _, (k, _, ci, co), bn in conv_shapes():
        parts.append(rng.normal(0, np.sqrt(2.0 / (k * k * ci)), (k, k, ci, co)).astype(np.float32).ravel())
        parts.append(rng.normal(0, 0.1, co).astype(np.float32))                      # biases
        if bn:
            parts.append(rng.uniform(0.5, 1.5, co).astype(np.float32) * 0.9)         # scale
            parts.append(rng.normal(0, 0.1, co).astype(np.float32))                  # shift
            parts.append(rng.normal(0, 0.1, co).astype(np.float32))                  # rolling_mean
            parts.append(rng.uniform(0.5, 1.5, co).astype(np.float32))               # rolling_variance
    return np.concatenate(parts)


def config4(steps, warmup, B=64):
    from yolo_amd.yolo_v2.Detector import Detector
    from yolo_amd.yolo_v2.yolo.yolo_v2 import yolo_v2
    net = yolo_v2(False, dtype="bf16").load_variables(synth_v2_variables())
    det = Detector(net)
    x = torch.from_numpy(np.random.default_rng(3).random((B, 416, 416, 3), dtype=np.float32) * 2 - 1).cuda()

    def step():
        boxes, probs = det.decode(net.build_networks(x))
        return det.nms(boxes, probs)
    dt = timed(step, steps, warmup)
    fwd = timed(lambda: net.build_networks(x), steps, warmup)
    counts = step()[0]
    return {"config": 4, "metric": "YOLOv2-416 VOC-20 images/sec end-to-end (conv + decode + class-agnostic NMS)",
            "value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 4), "batch": B, "dtype": "bf16",
            "conv_stack_ms": round(fwd * 1e3, 4), "conv_stack_tflops": round(29.360e9 * B / fwd / 1e12, 1),
            "detections_last_step": int(counts.sum().item()), "data": "synthetic"}


def config5(steps, warmup, B=8):
    from yolo_amd.model import yolov3
    from yolo_amd.utils import nms_utils
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    rng = np.random.default_rng(5)
    C = 80
    net = yolov3(C, DEFAULT_ANCHORS)                                # decode only: no weights needed
    fms = [torch.from_numpy(rng.standard_normal((B, g, g, 3 * (5 + C))).astype(np.float32)).cuda() for g in (19, 38, 76)]
    net.img_size = (608, 608)
    boxes, _, _, scores = net.predict(fms, return_scores=True)
    n = int(boxes.shape[1])
    out = []
    seeds = rng.random((200, 2)) * 560
    wh = rng.random((200, 2)) * 150 + 8
    base = np.concatenate([seeds, seeds + wh], 1)
    clustered = torch.from_numpy((base[rng.integers(0, 200, (B, n))] + rng.normal(0, 4, (B, n, 4))).astype(np.float32)).cuda()
    for name, bx in (("decoded", boxes), ("clustered", clustered)):
        dt = timed(lambda: nms_utils.gpu_nms_batched(bx, scores, C, 50, 0.001, 0.5), steps, warmup)
        cnt = nms_utils.gpu_nms_batched(bx, scores, C, 50, 0.001, 0.5)[4]
        alg = B * n * (4 + C) * 4.0                                 # read boxes + scores once
        out.append({"config": 5, "variant": name, "metric": "per-class NMS, %d images x %d boxes x %d classes, score_thresh 0.001" % (B, n, C),
                    "value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 4),
                    "hbm": {"achieved": round(alg / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / dt / 8e12, 4)},
                    "candidates_per_image": n * C, "kept_last_step": int(cnt.sum().item())})
    dt = timed(lambda: net.predict(fms, return_scores=True), steps, warmup)
    alg = B * (2.0 * n * 85 * 4 + n * 80 * 4)
    out.append({"config": 5, "variant": "decode", "metric": "predict (+ scores) at 608: %d images" % B, "value": round(B / dt, 1),
                "unit": "images/sec", "ms_per_step": round(dt * 1e3, 4),
                "hbm": {"achieved": round(alg / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / dt / 8e12, 4)}})
    return out


def config5_e2e(steps, warmup, B=8):
    """Config 5 widened: the whole chain at 608 x 608 with the eval-style threshold (model.py:22: the size comes from the
    input; eval.py:93-95 passes score_thresh through gpu_nms): forward + predict + conf*prob + per-class NMS in one call."""
    from yolo_amd.model import yolov3
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    from yolo_amd.weights import synth_darknet_stream
    net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(synth_darknet_stream(80, seed=1234))
    x = torch.from_numpy(np.random.default_rng(5).random((B, 608, 608, 3), dtype=np.float32)).cuda()
    dt = timed(lambda: net.detect(x, 50, 0.001, 0.5), steps, warmup)
    fwd = timed(lambda: net.forward(x), steps, warmup)
    cnt = net.detect(x, 50, 0.001, 0.5)[4]
    return {"config": 5, "variant": "end_to_end_608", "metric": "YOLOv3-608 images/sec end-to-end (conv + decode + NMS), score_thresh 0.001, "
            "max_boxes 50, batch %d" % B, "value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 4), "dtype": "bf16",
            "conv_stack_ms": round(fwd * 1e3, 4), "conv_stack_tflops": round(140.692e9 * B / fwd / 1e12, 1),
            "kept_last_step": int(cnt.sum().item()), "data": "synthetic"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    print(json.dumps(config4(a.steps, a.warmup)))
    for line in config5(a.steps, a.warmup):
        print(json.dumps(line))
    print(json.dumps(config5_e2e(a.steps, a.warmup)))
