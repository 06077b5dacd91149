import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
from yolo_amd.utils.nms_utils import gpu_nms_batched
from yolo_amd.weights import synth_darknet_stream
net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(synth_darknet_stream(80, seed=1234))
images = torch.from_numpy(np.random.default_rng(1).random((64, 416, 416, 3), dtype=np.float32)).cuda()
fms = net.forward(images)
a = net.postprocess(fms, 50, 0.5, 0.5)
boxes, _, _, scores = net.predict(fms, return_scores=True)
b = gpu_nms_batched(boxes, scores, 80, 50, 0.5, 0.5)
print("fused", int(a[4].sum()), "unfused", int(b[4].sum()), "equal counts", torch.equal(a[4], b[4]))
fms2 = net.forward(images)
print("forward deterministic:", all(torch.equal(x, y) for x, y in zip(fms, fms2)))
print("fm1 checksum", float(fms[0].double().sum()), float(fms[2].double().abs().sum()))
