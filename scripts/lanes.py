"""Experiment: images/s of detect() against the number of lanes (YOLO_LANES) and the batch size."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream

net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(synth_darknet_stream(80, seed=1234))
rng = np.random.default_rng(1)
for B in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "64").split(",")]:
    images = torch.from_numpy(rng.random((B, 416, 416, 3), dtype=np.float32)).cuda()
    ref = None
    for lanes in (1, 2, 3, 4):
        if lanes > B: break
        os.environ["YOLO_LANES"] = str(lanes)
        for _ in range(5): out = net.detect(images, 50, 0.5, 0.5)
        torch.cuda.synchronize()
        K = 20
        t0 = time.perf_counter()
        for _ in range(K): out = net.detect(images, 50, 0.5, 0.5)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        fms = [f.clone() for f in net.forward(images)]
        torch.cuda.synchronize()
        if ref is None: ref = fms
        same = all(torch.equal(a, b) for a, b in zip(ref, fms))
        print("B %3d lanes %d: %.3f ms/step  %.0f img/s  bitwise==lanes1: %s" % (B, lanes, dt * 1e3, B / dt, same), flush=True)
