"""CPU restatement of the FenHua/yolo `yolo_v3/` inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product (`yolo_amd/`) never does; it fails loudly when the HIP
library is missing.

Every function cites the reference file:line (relative to the reference checkout) it restates.
Arithmetic that the reference delegates to TensorFlow 1.x (not vendored in the reference tree,
`yolo_v3/README.md:15` "tensorflow >= 1.8.0") is restated from the published op semantics:

* conv / FusedBatchNorm(inference) / resize_nearest_neighbor(align_corners=True)   -> PARITY PINNED
  only against the reference's own PyTorch lesson `yolo-v3/002` (same Darknet-53 semantics) through
  `tests/golden/g1_*.npz`.
* `tf.image.non_max_suppression` (TF 1.x `non_max_suppression_op.cc`)               -> PARITY UNPINNED:
  the reference holds no vector for it.  Normative choice here: score descending, ties by ascending
  box index, suppress iff IoU > threshold (strict), boxes with area <= 0 never overlap.
* `py_nms` / `cpu_nms` are pinned by `tests/golden/g3_*.npz` (outputs of the reference numpy code).

Conv arithmetic uses torch-CPU fp32 `F.conv2d`; everything else is numpy float32.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------------------------
# Network description (reference: yolo_v3/utils/layer_utils.py:23-65, yolo_v3/model.py:37-65)
# --------------------------------------------------------------------------------------------


def yolov3_conv_specs(class_num=80):
    """List of the 75 convs in call order == Darknet `.weights` order.

    Each item: dict(cin, cout, k, stride, bn).  Order follows `darknet53_body`
    (layer_utils.py:31-53) then the head assembly in `yolov3.forward` (model.py:41-64).
    """
    specs = []

    def conv(cin, cout, k, s=1, bn=True, res_tail=False):
        specs.append(dict(cin=cin, cout=cout, k=k, stride=s, bn=bn, res_tail=res_tail))
        return cout

    def res(c, f):                                   # layer_utils.py:24-29
        conv(c, f, 1)
        conv(f, 2 * f, 3, res_tail=True)             # its output is added to the shortcut (:28)
        return 2 * f

    c = conv(3, 32, 3)                               # layer_utils.py:31
    c = conv(c, 64, 3, 2)                            # :32
    c = res(c, 32)                                   # :34
    c = conv(c, 128, 3, 2)                           # :35
    for _ in range(2):
        c = res(c, 64)                               # :37-38
    c = conv(c, 256, 3, 2)                           # :39
    for _ in range(8):
        c = res(c, 128)                              # :41-42
    c = conv(c, 512, 3, 2)                           # :44
    for _ in range(8):
        c = res(c, 256)                              # :46-47
    c = conv(c, 1024, 3, 2)                          # :49
    for _ in range(4):
        c = res(c, 512)                              # :51-52

    def yolo_block(cin, f):                          # layer_utils.py:56-65
        conv(cin, f, 1)
        conv(f, 2 * f, 3)
        conv(2 * f, f, 1)
        conv(f, 2 * f, 3)
        conv(2 * f, f, 1)
        conv(f, 2 * f, 3)

    nout = 3 * (5 + class_num)
    yolo_block(1024, 512)                            # model.py:41
    conv(1024, nout, 1, bn=False)                    # model.py:42-44
    conv(512, 256, 1)                                # model.py:47
    yolo_block(256 + 512, 256)                       # model.py:49-51
    conv(512, nout, 1, bn=False)                     # model.py:52-54
    conv(256, 128, 1)                                # model.py:57
    yolo_block(128 + 256, 128)                       # model.py:59-61
    conv(256, nout, 1, bn=False)                     # model.py:62-64
    return specs


def darknet_stream_len(specs):
    n = 0
    for s in specs:
        n += (4 * s["cout"] if s["bn"] else s["cout"]) + s["cout"] * s["cin"] * s["k"] ** 2
    return n


def synth_darknet_stream(specs, seed=1234, res_gamma=0.2, head_gain=0.55, gamma_scale=0.9):
    """Seeded synthetic Darknet-order float32 stream.

    Per conv: [beta, gamma, mean, var] (or [bias]) then kernel [Cout, Cin, kh, kw]
    (yolo_v3/utils/misc_utils.py:45-96; yolo-v3/002/darknet.py:234-264).
    conf-logit bias -4.0 on head channels 4, 4+(5+C), 4+2(5+C).
    Deviation from SURVEY 8(d): gamma of the conv feeding each residual add is scaled by `res_gamma`
    and head kernels by `head_gain`; with gamma~U(0.5,1.5) everywhere the 23 residual adds double the
    variance each (|fm| ~ 6e4, every exp() saturates), which makes decode/NMS inputs meaningless.
    """
    rng = np.random.default_rng(seed)
    parts = []
    for s in specs:
        co, ci, k = s["cout"], s["cin"], s["k"]
        if s["bn"]:
            beta = rng.normal(0, 0.1, co)
            gamma = rng.uniform(0.5, 1.5, co) * (res_gamma if s.get("res_tail") else gamma_scale)
            mean = rng.normal(0, 0.1, co)
            var = rng.uniform(0.5, 1.5, co)
            parts += [beta, gamma, mean, var]
        else:
            bias = rng.normal(0, 0.1, co)
            na = co // 3
            bias[4::na] = -4.0
            parts.append(bias)
        w = rng.normal(0, np.sqrt(2.0 / (k * k * ci)), (co, ci, k, k))
        if not s["bn"]:
            w *= head_gain
        parts.append(w.ravel())
    return np.concatenate(parts).astype(F32)


def write_darknet_weights(path, stream, header=(0, 2, 0, 0, 0)):
    """5 x int32 header + float32 body (misc_utils.py:52-55)."""
    with open(path, "wb") as fp:
        np.asarray(header, np.int32).tofile(fp)
        np.asarray(stream, F32).tofile(fp)


def read_darknet_weights(path):
    with open(path, "rb") as fp:
        np.fromfile(fp, dtype=np.int32, count=5)              # misc_utils.py:54
        return np.fromfile(fp, dtype=np.float32)              # misc_utils.py:55


def split_darknet_stream(specs, stream):
    """Walk the stream like misc_utils.py:59-95 -> per conv dict(w HWIO, and bn params or bias)."""
    out, ptr = [], 0
    for s in specs:
        co, ci, k = s["cout"], s["cin"], s["k"]
        d = {}
        if s["bn"]:
            for name in ("beta", "gamma", "mean", "var"):      # misc_utils.py:67-74
                d[name] = stream[ptr:ptr + co].copy()
                ptr += co
        else:
            d["bias"] = stream[ptr:ptr + co].copy()            # misc_utils.py:77-85
            ptr += co
        n = co * ci * k * k
        w = stream[ptr:ptr + n].reshape(co, ci, k, k)          # misc_utils.py:89-90
        d["w"] = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))   # HWIO, misc_utils.py:92
        ptr += n
        out.append(d)
    assert ptr == len(stream), (ptr, len(stream))
    return out


# --------------------------------------------------------------------------------------------
# Layers (NHWC float32 numpy in / out)
# --------------------------------------------------------------------------------------------


def leaky_relu(x, alpha=0.1):
    """model.py:8-12: f1*x + f2*|x| with f1, f2 Python doubles cast to the tensor dtype."""
    f1 = F32(0.5 * (1 + alpha))
    f2 = F32(0.5 * (1 - alpha))
    return (f1 * x + f2 * np.abs(x)).astype(F32)


def conv2d(x, p, spec):
    """layer_utils.py:9-21 under the arg_scope of model.py:32-36.

    stride 1: SAME; stride 2: explicit pad (k-1)//2 before / rest after then VALID.
    BN convs: conv(no bias) -> (y-mean)*gamma/sqrt(var+1e-5)+beta -> leaky 0.1.
    Non-BN convs (detection heads, model.py:42-44): conv + bias, no activation.
    """
    import torch
    import torch.nn.functional as TF

    k, s = spec["k"], spec["stride"]
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(np.transpose(p["w"], (3, 2, 0, 1))))
    pad_total = k - 1
    pb = pad_total // 2
    pe = pad_total - pb
    xt = TF.pad(xt, (pb, pe, pb, pe))        # SAME for odd k stride 1 == same symmetric pad
    y = TF.conv2d(xt, wt, None, stride=s).permute(0, 2, 3, 1).contiguous().numpy()
    if spec["bn"]:
        inv = (p["gamma"] / np.sqrt(p["var"] + F32(1e-5))).astype(F32)
        y = (y - p["mean"]) * inv + p["beta"]
        y = leaky_relu(y.astype(F32))
    else:
        y = y + p["bias"]
    return y.astype(F32)


def upsample_layer(x, out_shape):
    """layer_utils.py:67-72: resize_nearest_neighbor(align_corners=True).

    TF1 kernel: scale = (in-1)/(out-1) when align_corners and out > 1;
    src = min(round(dst*scale), in-1).
    """
    nh, nw = out_shape[1], out_shape[2]
    ih, iw = x.shape[1], x.shape[2]

    def idx(n_in, n_out):
        scale = F32((n_in - 1) / (n_out - 1)) if n_out > 1 else F32(0)
        d = np.arange(n_out, dtype=F32)
        return np.minimum(np.floor(d * scale + F32(0.5)).astype(np.int64), n_in - 1)

    return x[:, idx(ih, nh)][:, :, idx(iw, nw)]


def forward(params, x, class_num=80, taps=None):
    """yolov3.forward (model.py:21-67).  x: [B,H,W,3] float32 RGB in [0,1].

    `taps`, when a dict, receives intermediate tensors (route_1/2/3, concat1/2).
    Returns (feature_map_1, feature_map_2, feature_map_3), NHWC.
    """
    specs = yolov3_conv_specs(class_num)
    it = iter(range(len(specs)))

    def conv(t):
        i = next(it)
        return conv2d(t, params[i], specs[i])

    def res(t):                                      # layer_utils.py:24-29
        sc = t
        t = conv(t)
        t = conv(t)
        return t + sc

    net = conv(x)
    net = conv(net)
    net = res(net)
    net = conv(net)
    for _ in range(2):
        net = res(net)
    net = conv(net)
    for _ in range(8):
        net = res(net)
    route_1 = net
    net = conv(net)
    for _ in range(8):
        net = res(net)
    route_2 = net
    net = conv(net)
    for _ in range(4):
        net = res(net)
    route_3 = net

    def yolo_block(t):                               # layer_utils.py:56-65
        for _ in range(5):
            t = conv(t)
        route = t
        t = conv(t)
        return route, t

    inter1, net = yolo_block(route_3)                # model.py:41
    fm1 = conv(net)                                  # model.py:42-44
    inter1 = conv(inter1)                            # model.py:47
    inter1 = upsample_layer(inter1, route_2.shape)   # model.py:48
    concat1 = np.concatenate([inter1, route_2], axis=3)     # model.py:49
    inter2, net = yolo_block(concat1)                # model.py:51
    fm2 = conv(net)                                  # model.py:52-54
    inter2 = conv(inter2)                            # model.py:57
    inter2 = upsample_layer(inter2, route_1.shape)   # model.py:58
    concat2 = np.concatenate([inter2, route_1], axis=3)     # model.py:59
    _, net = yolo_block(concat2)                     # model.py:61
    fm3 = conv(net)                                  # model.py:62-64
    if taps is not None:
        taps.update(route_1=route_1, route_2=route_2, route_3=route_3,
                    concat1=concat1, concat2=concat2)
    return fm1, fm2, fm3


# --------------------------------------------------------------------------------------------
# Decode (model.py:69-153)
# --------------------------------------------------------------------------------------------


def _sigmoid(x):
    x = x.astype(F32)
    return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)


def reorg_layer(feature_map, anchors, img_size, class_num=80):
    """yolov3.reorg_layer (model.py:69-108).  img_size = (H, W) recorded by forward (model.py:22)."""
    gh, gw = feature_map.shape[1:3]                                   # model.py:76
    ratio = np.asarray([img_size[0] / gh, img_size[1] / gw], F32)     # model.py:78 (h, w)
    rescaled = np.asarray([(a[0] / ratio[1], a[1] / ratio[0]) for a in anchors], F32)   # :80
    fm = feature_map.reshape(-1, gh, gw, 3, 5 + class_num)            # :81
    centers, sizes = fm[..., 0:2], fm[..., 2:4]                       # :87
    conf_logits, prob_logits = fm[..., 4:5], fm[..., 5:]
    centers = _sigmoid(centers)                                       # :88
    gx, gy = np.meshgrid(np.arange(gw, dtype=np.int32), np.arange(gh, dtype=np.int32))  # :90-92
    x_y_offset = np.concatenate([gx.reshape(-1, 1), gy.reshape(-1, 1)], axis=-1)        # :93-95
    x_y_offset = x_y_offset.reshape(gh, gw, 1, 2).astype(F32)         # :97
    centers = (centers + x_y_offset) * ratio[::-1]                    # :99-101
    sizes = np.clip(np.exp(sizes.astype(F32)), F32(1e-9), F32(50)) * rescaled   # :103
    sizes = sizes * ratio[::-1]                                       # :105
    boxes = np.concatenate([centers, sizes], axis=-1).astype(F32)     # :107
    return x_y_offset, boxes, conf_logits, prob_logits


def predict(feature_maps, anchors, img_size, class_num=80):
    """yolov3.predict (model.py:110-153) -> boxes [B,n,4] (x0,y0,x1,y1), confs [B,n,1], probs [B,n,C]."""
    fm1, fm2, fm3 = feature_maps
    pairs = [(fm1, anchors[6:9]), (fm2, anchors[3:6]), (fm3, anchors[0:3])]      # :116-118
    bl, cl, pl = [], [], []
    for fm, anc in pairs:
        xy, boxes, conf_logits, prob_logits = reorg_layer(fm, anc, img_size, class_num)
        gh, gw = xy.shape[:2]
        bl.append(boxes.reshape(-1, gh * gw * 3, 4))                             # :123
        cl.append(_sigmoid(conf_logits.reshape(-1, gh * gw * 3, 1)))             # :124,134
        pl.append(_sigmoid(prob_logits.reshape(-1, gh * gw * 3, class_num)))     # :125,135
    boxes = np.concatenate(bl, axis=1)                                           # :142
    confs = np.concatenate(cl, axis=1)
    probs = np.concatenate(pl, axis=1)
    cx, cy, w, h = [boxes[..., i:i + 1] for i in range(4)]                       # :147
    two = F32(2)
    boxes = np.concatenate([cx - w / two, cy - h / two, cx + w / two, cy + h / two], axis=-1)  # :148-152
    return boxes.astype(F32), confs.astype(F32), probs.astype(F32)


# --------------------------------------------------------------------------------------------
# NMS
# --------------------------------------------------------------------------------------------


def tf_non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """`tf.image.non_max_suppression` (TF 1.x), called at nms_utils.py:27-30.  PARITY UNPINNED.

    Pure-numpy float32 restatement (small cases; the C twin in nms_oracle.c is used at size).
    Returns selected indices (int32) in selection order.
    """
    boxes = np.asarray(boxes, F32).reshape(-1, 4)
    scores = np.asarray(scores, F32).reshape(-1)
    thr = F32(iou_threshold)
    n = len(scores)
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))   # desc score, asc index
    lo0 = np.minimum(boxes[:, 0], boxes[:, 2])
    lo1 = np.minimum(boxes[:, 1], boxes[:, 3])
    hi0 = np.maximum(boxes[:, 0], boxes[:, 2])
    hi1 = np.maximum(boxes[:, 1], boxes[:, 3])
    area = ((hi0 - lo0) * (hi1 - lo1)).astype(F32)
    sel = []
    for i in order:
        if len(sel) >= min(max_output_size, n):
            break
        ok = True
        for j in sel:
            if area[i] <= 0 or area[j] <= 0:
                continue
            d0 = max(F32(min(hi0[i], hi0[j]) - max(lo0[i], lo0[j])), F32(0))
            d1 = max(F32(min(hi1[i], hi1[j]) - max(lo1[i], lo1[j])), F32(0))
            inter = F32(d0 * d1)
            iou = F32(inter / F32(F32(area[i] + area[j]) - inter))
            if iou > thr:
                ok = False
                break
        if ok:
            sel.append(i)
    return np.asarray(sel, np.int32)


_NMS_LIB = None


def _nms_lib():
    """Compile (once) and load oracle/nms_oracle.c -> oracle/_build/libnms_oracle.so."""
    global _NMS_LIB
    if _NMS_LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libnms_oracle.so")
        src = os.path.join(here, "nms_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build_c_oracle()
        lib = ctypes.CDLL(so)
        lib.oracle_class_nms.restype = ctypes.c_int
        lib.oracle_class_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _NMS_LIB = lib
    return _NMS_LIB


def build_c_oracle():
    here = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(os.path.join(here, "_build"), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                           "-o", os.path.join(here, "_build", "libnms_oracle.so"),
                           os.path.join(here, "nms_oracle.c")])


def class_nms_c(boxes, scores, c, max_boxes, score_thresh, iou_thresh, mode):
    """C oracle for one class: returns original box indices in selection order.

    mode 0 = gpu_nms semantics (nms_utils.py:22-33, TF op); mode 1 = cpu_nms/py_nms (nms_utils.py:40-90).
    """
    boxes = np.ascontiguousarray(boxes, F32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, F32)
    n, C = scores.shape
    out = np.empty(max(min(max_boxes, n), 1), np.int32)
    cnt = np.zeros(1, np.int32)
    rc = _nms_lib().oracle_class_nms(boxes.ctypes.data, scores.ctypes.data, n, C, c, max_boxes,
                                     score_thresh, iou_thresh, mode, out.ctypes.data, cnt.ctypes.data)
    assert rc == 0
    return out[:cnt[0]].copy()


def gpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5, use_c=True,
            return_indices=False):
    """nms_utils.py:7-37.  Batch dim is flattened (:20-21) exactly like the reference.

    Returns (boxes [m,4] f32, score [m] f32, label [m] i32) (+ original flat indices when asked).
    """
    boxes = np.asarray(boxes, F32).reshape(-1, 4)                     # :20
    score = np.asarray(scores, F32).reshape(-1, num_classes)          # :21
    thr = F32(score_thresh)
    bl, sl, ll, il = [], [], [], []
    for c in range(num_classes):                                      # :23
        if use_c:
            idx = class_nms_c(boxes, score, c, max_boxes, score_thresh, iou_thresh, 0)
        else:
            m = np.nonzero(score[:, c] >= thr)[0]                     # :22,25-26
            k = tf_non_max_suppression(boxes[m], score[m, c], max_boxes, iou_thresh)   # :27-30
            idx = m[k]
        bl.append(boxes[idx])                                         # :32
        sl.append(score[idx, c])                                      # :33
        ll.append(np.full(len(idx), c, np.int32))                     # :31
        il.append(np.asarray(idx, np.int32))
    res = (np.concatenate(bl, 0).reshape(-1, 4), np.concatenate(sl, 0), np.concatenate(ll, 0))   # :34-36
    if return_indices:
        return res + (np.concatenate(il, 0),)
    return res


def py_nms(boxes, scores, max_boxes=50, iou_thresh=0.5):
    """nms_utils.py:40-64 restated with an explicit tie rule.

    The reference orders with `scores.argsort()[::-1]` (:49); numpy's default sort is unstable, so the
    order of exactly-equal scores is implementation-defined there.  Normative rule here: stable
    ascending sort then reversal, i.e. equal scores come out in DESCENDING index order.
    """
    boxes = np.asarray(boxes, F32)
    scores = np.asarray(scores, F32)
    assert boxes.shape[1] == 4 and len(scores.shape) == 1            # :42
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)                                    # :48
    order = np.argsort(scores, kind="stable")[::-1]                  # :49 (+ tie rule)
    keep = []
    thr = F32(iou_thresh)
    one = F32(1)
    while order.size > 0:                                            # :51
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        xx1 = np.maximum(x1[i], x1[rest])                            # :54-57
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(F32(0), xx2 - xx1 + one)                      # :58
        h = np.maximum(F32(0), yy2 - yy1 + one)                      # :59
        inter = w * h                                                # :60
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)           # :61
        inds = np.where(ovr <= thr)[0]                               # :62
        order = order[inds + 1]                                      # :63
    return keep[:max_boxes]                                          # :64


def cpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
    """nms_utils.py:67-90."""
    boxes = np.asarray(boxes, F32).reshape(-1, 4)                    # :69
    scores = np.asarray(scores, F32).reshape(-1, num_classes)        # :70
    pb, ps, pl = [], [], []
    for i in range(num_classes):                                     # :73
        indices = np.where(scores[:, i] >= F32(score_thresh))        # :74
        fb = boxes[indices]
        fs = scores[:, i][indices]
        if len(fb) == 0:                                             # :77-78
            continue
        k = py_nms(fb, fs, max_boxes=max_boxes, iou_thresh=iou_thresh)   # :80-81
        pb.append(fb[k])
        ps.append(fs[k])
        pl.append(np.ones(len(k), dtype="int32") * i)                # :84
    if len(pb) == 0:                                                 # :85-86
        return None, None, None
    return np.concatenate(pb, 0), np.concatenate(ps, 0), np.concatenate(pl, 0)   # :87-90


# --------------------------------------------------------------------------------------------
# Driver counterpart (test_single_image.py:31-45) on already-preprocessed input
# --------------------------------------------------------------------------------------------


def detect(params, anchors, img, class_num=80, max_boxes=30, score_thresh=0.4, iou_thresh=0.5):
    """test_single_image.py:31-40: forward -> predict -> conf*prob -> gpu_nms.  img: [1,H,W,3]."""
    fms = forward(params, img, class_num)
    boxes, confs, probs = predict(fms, anchors, img.shape[1:3], class_num)
    scores = confs * probs                                           # :38
    return gpu_nms(boxes, scores, class_num, max_boxes, score_thresh, iou_thresh)   # :39


YOLO_ANCHORS = np.asarray([10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326],
                          F32).reshape(-1, 2)      # yolo_v3/data/yolo_anchors.txt:1
