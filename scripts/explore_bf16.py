import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import yolo_oracle as O
from yolo_amd.model import yolov3
from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
specs = O.yolov3_conv_specs(80)
stream = O.synth_darknet_stream(specs, seed=1234)
params = O.split_darknet_stream(specs, stream)
net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16").load_weight_stream(stream)
img = np.random.default_rng(1).random((2, 416, 416, 3), dtype=np.float32)
fms = O.forward(params, img)
rb, rc, rp = O.predict(fms, O.YOLO_ANCHORS, (416, 416))
rs = (rc * rp).astype(np.float32)
gb, gc, gp, gs = [t.cpu().numpy() for t in net.predict(net.forward(img), return_scores=True)]
d = np.abs(gs - rs)
print("scores: max |d| %.4f  mean %.5f; share > 0.01: %.5f" % (d.max(), d.mean(), (d > 0.01).mean()))
def iou(a, b):
    x0 = np.maximum(a[..., 0], b[..., 0]); y0 = np.maximum(a[..., 1], b[..., 1]); x1 = np.minimum(a[..., 2], b[..., 2]); y1 = np.minimum(a[..., 3], b[..., 3])
    inter = np.clip(x1 - x0, 0, None) * np.clip(y1 - y0, 0, None)
    ua = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    return inter / np.maximum(ua, 1e-9)
bi = iou(gb, rb)
print("boxes: IoU min %.4f, share < 0.9: %.5f, share < 0.95: %.5f" % (bi.min(), (bi < 0.9).mean(), (bi < 0.95).mean()))
for b in range(2):
    s = np.sort(rs[b][(rs[b] > 0.3) & (rs[b] < 0.7)])
    print("image", b, "oracle scores in (0.3,0.7):", len(s), " >=0.5:", int((rs[b] >= 0.5).sum()))
    gaps = np.diff(s)
    for lo, hi in ((0.45, 0.55), (0.4, 0.6), (0.3, 0.7)):
        sel = (s[:-1] >= lo) & (s[1:] <= hi)
        i = np.argmax(np.where(sel, gaps, 0))
        print("   widest gap in (%.2f,%.2f): %.5f at %.5f..%.5f" % (lo, hi, gaps[i], s[i], s[i + 1]))
    # detection-level at several thresholds
    for thr in (0.5, 0.6, 0.7):
        ro = O.gpu_nms(rb[b], rs[b], 80, 50, thr, 0.5, return_indices=True)
        go = O.gpu_nms(gb[b], gs[b], 80, 50, thr, 0.5, return_indices=True)
        rset = set(zip(ro[3].tolist(), ro[2].tolist())); gset = set(zip(go[3].tolist(), go[2].tolist()))
        print("   thr %.2f: oracle %d det, bf16 %d det, same (index,class): %d" % (thr, len(rset), len(gset), len(rset & gset)))
print("---- gap thresholds, 4 images")
img = np.random.default_rng(1).random((4, 416, 416, 3), dtype=np.float32)
fms = O.forward(params, img)
rb, rc, rp = O.predict(fms, O.YOLO_ANCHORS, (416, 416))
rs = (rc * rp).astype(np.float32)
gb, gc, gp, gs = [t.cpu().numpy() for t in net.predict(net.forward(img), return_scores=True)]
tot = hit = 0
for b in range(4):
    s = np.sort(rs[b][(rs[b] >= 0.56) & (rs[b] <= 0.64)])
    gaps = np.diff(s); i = int(np.argmax(gaps)); thr = float((s[i] + s[i + 1]) / 2)
    ro = O.gpu_nms(rb[b], rs[b], 80, 50, thr, 0.5, return_indices=True)
    go = O.gpu_nms(gb[b], gs[b], 80, 50, thr, 0.5, return_indices=True)
    def share(A, B):
        h = 0
        for k in range(len(A[2])):
            cand = np.nonzero(B[2] == A[2][k])[0]
            if len(cand) == 0: continue
            io = iou(A[0][k][None, :], B[0][cand]); j = int(np.argmax(io))
            if io[j] >= 0.9 and abs(float(B[1][cand[j]]) - float(A[1][k])) <= 0.03: h += 1
        return h
    h1, h2 = share(ro, go), share(go, ro)
    print("image %d thr %.5f (gap %.5f): oracle %d bf16 %d matched %d / %d" % (b, thr, gaps[i], len(ro[2]), len(go[2]), h1, h2))
    tot += len(ro[2]) + len(go[2]); hit += h1 + h2
print("pooled share %.4f of %d" % (hit / tot, tot))
