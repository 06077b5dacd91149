"""YOLOv2-416 batch 64 forwards for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 scripts/v2_trace.py"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_secondary import synth_v2_variables
from yolo_amd.yolo_v2.yolo.yolo_v2 import yolo_v2
net = yolo_v2(False, dtype="bf16").load_variables(synth_v2_variables())
x = torch.from_numpy(np.random.default_rng(3).random((64, 416, 416, 3), dtype=np.float32) * 2 - 1).cuda()
os.environ["YOLO_LANES"] = os.environ.get("YOLO_LANES", "1")
for _ in range(8):
    net.build_networks(x)
torch.cuda.synchronize()
