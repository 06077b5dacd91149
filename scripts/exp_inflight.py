"""Experiment: consecutive batches issued on alternating (stream, handle) pairs so that batch i+1 overlaps the tail of batch i.
python scripts/exp_inflight.py [--steps 30] ; prints ms per step for 1 / 2 batches in flight at 1, 2, 3 lanes."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--inflight", default="1,2,3")
ap.add_argument("--lanes", default="1,2,3")
args = ap.parse_args()
B = args.batch
stream = synth_darknet_stream(80, seed=1234)
nmax = max(int(v) for v in args.inflight.split(","))
nets = [yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(stream) for _ in range(nmax)]
streams = [torch.cuda.Stream() for _ in range(nmax)]
images = torch.from_numpy(np.random.default_rng(1).random((B, 416, 416, 3), dtype=np.float32)).cuda()
torch.cuda.synchronize()

def run(nfl, steps):
    outs = []
    for i in range(steps):
        k = i % nfl
        with torch.cuda.stream(streams[k]):
            outs.append(nets[k].detect(images, 50, 0.5, 0.5))
    return outs

for lanes in [int(v) for v in args.lanes.split(",")]:
    os.environ["YOLO_LANES"] = str(lanes)
    for nfl in [int(v) for v in args.inflight.split(",")]:
        run(nfl, 6)
        torch.cuda.synchronize()
        res = []
        for rep in range(3):
            t0 = time.perf_counter()
            outs = run(nfl, args.steps)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / args.steps * 1e3)
        nd = int(outs[-1][4].sum().item())
        print("lanes %d inflight %d: ms/step %s  det %d" % (lanes, nfl, " ".join("%.3f" % r for r in res), nd), flush=True)
