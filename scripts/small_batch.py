"""Latency of one forward / one detect at small batch, default plan vs the split-K small-batch plan
(python scripts/small_batch.py [B ...])."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream
Bs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
ws = synth_darknet_stream(80, seed=1234)
nets = {"default": yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(ws),
        "split-K": yolov3(80, DEFAULT_ANCHORS, dtype="bf16", small_batch_plan=True).load_weight_stream(ws)}

def t(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best

for B in Bs:
    x = torch.rand(B, 416, 416, 3, device="cuda")
    row = []
    for name, net in nets.items():
        row.append("%s forward %.3f ms detect %.3f ms" % (name, t(lambda: net.forward(x)), t(lambda: net.detect(x))))
    print("batch %2d: " % B + " | ".join(row), flush=True)
