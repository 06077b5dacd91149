"""Batch-1 forwards of the small-batch plan for a kernel trace (rocprofv3 --kernel-trace --output-format csv -- python3 scripts/b1_trace.py [B] [default]),
and the reader of that trace (python3 scripts/b1_trace.py --read <kernel_trace.csv>): the kernels of the last forward with start offset, duration and the
gap to the previous kernel's end."""
import sys, os, csv
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = max(i for i, r in enumerate(rows) if "front_kernel" in r["Kernel_Name"])
    t0 = int(rows[last]["Start_Timestamp"]); prev_end = t0; busy = 0
    for r in rows[last:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%8.1f us  dur %6.1f  gap %5.1f  grid %8s wg %4s lds %6s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
              r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:70]))
        busy += e - s; prev_end = e
    print("span %.1f us, kernels busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.weights import synth_darknet_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16", small_batch_plan=len(sys.argv) <= 2).load_weight_stream(synth_darknet_stream(80, seed=1234))
x = torch.rand(B, 416, 416, 3, device="cuda")
for _ in range(30):
    net.forward(x)
torch.cuda.synchronize()
