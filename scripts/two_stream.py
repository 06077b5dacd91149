"""Experiment: does running the batch as S independent sub-batches on S HIP streams hide tile-quantisation tails?"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream

B = 64
ws = synth_darknet_stream(80, seed=1234)
rng = np.random.default_rng(1)
images = torch.from_numpy(rng.random((B, 416, 416, 3), dtype=np.float32)).cuda()
for S in (1, 2, 4):
    nets = [yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(ws) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = list(images.chunk(S))
    def step():
        for n, s, p in zip(nets, streams, parts):
            with torch.cuda.stream(s):
                n.detect(p, 50, 0.5, 0.5)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("streams %d: %.3f ms/step  %.0f img/s" % (S, dt * 1e3, B / dt), flush=True)
