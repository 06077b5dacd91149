"""Experiment: replay detect() from a captured HIP graph (torch.cuda.CUDAGraph) against eager launches."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream

net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(synth_darknet_stream(80, seed=1234))
rng = np.random.default_rng(1)
for B in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "1,4,16,64").split(",")]:
    images = torch.from_numpy(rng.random((B, 416, 416, 3), dtype=np.float32)).cuda()
    def timeit(fn, K=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K
    eager = timeit(lambda: net.detect(images, 50, 0.5, 0.5))
    ref = [t.clone() for t in net.detect(images, 50, 0.5, 0.5)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net.detect(images, 50, 0.5, 0.5)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = net.detect(images, 50, 0.5, 0.5)
    graphed = timeit(g.replay)
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(ref[4:], out[4:]))
    print("B %3d eager %.3f ms (%.0f img/s)  graph %.3f ms (%.0f img/s)  counts equal: %s" % (B, eager * 1e3, B / eager, graphed * 1e3, B / graphed, same), flush=True)
