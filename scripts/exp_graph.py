"""Does replaying the whole detect step as a HIP graph (torch.cuda.CUDAGraph) beat launching it?  python scripts/exp_graph.py [B]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(synth_darknet_stream(80, seed=1234))
x = torch.rand(B, 416, 416, 3, device="cuda")
for _ in range(10):
    out = net.detect(x, 50, 0.5, 0.5)
torch.cuda.synchronize()

def timeit(fn, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for r in range(3):
    print("launched: %.4f ms/step" % timeit(lambda: net.detect(x, 50, 0.5, 0.5)), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        net.detect(x, 50, 0.5, 0.5)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="relaxed"):
    gout = net.detect(x, 50, 0.5, 0.5)
for r in range(3):
    print("graph replay: %.4f ms/step" % timeit(g.replay), flush=True)
ref = net.detect(x, 50, 0.5, 0.5)
g.replay(); torch.cuda.synchronize()
print("replay equals launch:", all(torch.equal(a, b) for a, b in zip(gout, ref)))
