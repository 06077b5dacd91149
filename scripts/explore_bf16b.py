import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import yolo_oracle as O
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
def iou(a, b):
    x0 = np.maximum(a[..., 0], b[..., 0]); y0 = np.maximum(a[..., 1], b[..., 1]); x1 = np.minimum(a[..., 2], b[..., 2]); y1 = np.minimum(a[..., 3], b[..., 3])
    inter = np.clip(x1 - x0, 0, None) * np.clip(y1 - y0, 0, None)
    ua = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    return inter / np.maximum(ua, 1e-9)
specs = O.yolov3_conv_specs(80)
img = np.random.default_rng(1).random((2, 416, 416, 3), dtype=np.float32)
for hg in (0.55, 1.0, 1.5, 2.5):
    stream = O.synth_darknet_stream(specs, seed=1234, head_gain=hg)
    params = O.split_darknet_stream(specs, stream)
    net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(stream)
    fms = O.forward(params, img)
    rb, rc, rp = O.predict(fms, O.YOLO_ANCHORS, (416, 416))
    rs = (rc * rp).astype(np.float32)
    gb, gc, gp, gs = [t.cpu().numpy() for t in net.predict(net.forward(img), return_scores=True)]
    d = np.abs(gs - rs)
    tot = hit = same = 0
    for b in range(2):
        ro = O.gpu_nms(rb[b], rs[b], 80, 50, 0.5, 0.5, return_indices=True)
        go = O.gpu_nms(gb[b], gs[b], 80, 50, 0.5, 0.5, return_indices=True)
        def share(A, B):
            h = 0
            for k in range(len(A[2])):
                cand = np.nonzero(B[2] == A[2][k])[0]
                if len(cand) == 0: continue
                io = iou(A[0][k][None, :], B[0][cand]); j = int(np.argmax(io))
                if io[j] >= 0.9 and abs(float(B[1][cand[j]]) - float(A[1][k])) <= 0.03: h += 1
            return h
        tot += len(ro[2]) + len(go[2]); hit += share(ro, go) + share(go, ro)
        same += len(set(zip(ro[3].tolist(), ro[2].tolist())) & set(zip(go[3].tolist(), go[2].tolist())))
    print("head_gain %.2f: max|dscore| %.4f, candidates >= 0.5: %d, in [0.48,0.52]: %d, detections %d, matched@0.9 %.4f, same index %d" % (
        hg, d.max(), int((rs >= 0.5).sum()), int(((rs >= 0.48) & (rs <= 0.52)).sum()), tot, hit / max(1, tot), same), flush=True)
