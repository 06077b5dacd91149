"""Per-layer times of one forward at small batch (python scripts/b1_layers.py [B] [split])."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16", small_batch_plan=len(sys.argv) > 2).load_weight_stream(synth_darknet_stream(80, seed=1234))
x = torch.rand(B, 416, 416, 3, device="cuda")
acc = None
for _ in range(6):
    _, ms, info = net.forward_timed(x)
    acc = ms.astype(np.float64) if acc is None else acc + ms
ms = acc / 6
groups = {}
for i, (m, (cin, cout, k, s)) in enumerate(zip(ms, info)):
    print("%2d %4d->%4d k%d s%d %7.1f us  %s" % (i, cin, cout, k, s, m * 1e3, net.layer_kernels[i][:60]))
print("total %.3f ms" % ms.sum())
