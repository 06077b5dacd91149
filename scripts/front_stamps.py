"""Phase breakdown of front_kernel (library built with `make STAMPS=1`): cycles wave 0 of workgroup 0 spends per phase, per tile."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["YOLO_HIP_LIB"] = os.path.join(ROOT, "yolo_amd", "lib", "libyolo_hip_stamps.so")
import numpy as np, torch
from yolo_amd import _lib
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
os.environ["YOLO_LANES"] = "1"
net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(synth_darknet_stream(80, seed=1234))
x = torch.rand(B, 416, 416, 3, device="cuda")
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
buf = np.zeros(8, np.uint64)
L = _lib.lib()
L.yolo_debug_read_front_stamps.argtypes = [ctypes.c_void_p]
assert L.yolo_debug_read_front_stamps(buf.ctypes.data) == 0
names = ["next patch: load issue", "stem groups", "barrier (stem done)", "conv MFMAs + act", "barrier (stem reads done)", "staging + barrier",
         "patch -> LDS + out stores", "last barrier"]
ntiles = B * 13 * 26
grid = min(512, ntiles)
per = -(-ntiles // grid)
tot = float(buf.sum())
print("batch %d: %d tiles on %d workgroups (%d tiles for workgroup 0), %.0f cycles per tile" % (B, ntiles, grid, per, tot / per))
for n, v in zip(names, buf):
    print("  %-28s %8.0f cycles / tile  (%4.1f %%)" % (n, float(v) / per, 100.0 * float(v) / tot))
