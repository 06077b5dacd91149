"""CPU restatement of the reference YOLOv2 inference path (config 4).  TEST INFRASTRUCTURE ONLY.

Network: reference yolo_v2/yolo/yolo_v2.py:32-113 (build_networks / conv_layer / pooling_layer / reorg);
decode + class-agnostic NMS: reference yolo_v2/Detector.py:51-115; constants yolo_v2/yolo/config.py:8-25.
TensorFlow ops (tf.nn.conv2d SAME, tf.nn.batch_normalization, tf.nn.max_pool) are restated from their
published semantics (parity unpinned for the conv stack: the reference holds no vector and no weights);
`calc_output` is pinned by tests/golden/g4_v2_decode.npz (outputs of the reference numpy code).
"""
import numpy as np

F32 = np.float32

CLASSES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
           'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']   # config.py:8-11
ANCHOR = [0.57273, 1.87446, 3.33843, 7.88282, 9.77052, 0.677385, 2.06253, 5.47434, 3.52778, 9.16828]      # config.py:13
IMAGE_SIZE, CELL_SIZE, BOX_PER_CELL, ALPHA, THRESHOLD = 416, 13, 5, 0.1, 0.3                               # config.py:16-25


def conv_specs(num_class=20, box_per_cell=5):
    """(name, k, cin, cout, batch_norm) in variable-creation order (yolo_v2.py:34-68)."""
    s = [("0_conv", 3, 3, 32), ("2_conv", 3, 32, 64), ("4_conv", 3, 64, 128), ("5_conv", 1, 128, 64),
         ("6_conv", 3, 64, 128), ("8_conv", 3, 128, 256), ("9_conv", 1, 256, 128), ("10_conv", 3, 128, 256),
         ("12_conv", 3, 256, 512), ("13_conv", 1, 512, 256), ("14_conv", 3, 256, 512), ("15_conv", 1, 512, 256),
         ("16_conv", 3, 256, 512), ("18_conv", 3, 512, 1024), ("19_conv", 1, 1024, 512), ("20_conv", 3, 512, 1024),
         ("21_conv", 1, 1024, 512), ("22_conv", 3, 512, 1024), ("23_conv", 3, 1024, 1024), ("24_conv", 3, 1024, 1024),
         ("26_conv", 1, 512, 64), ("29_conv", 3, 1280, 1024)]
    out = [dict(name=n, k=k, cin=ci, cout=co, bn=True) for n, k, ci, co in s]
    out.append(dict(name="30_conv", k=1, cin=1024, cout=box_per_cell * (num_class + 5), bn=False))
    return out


def synth_stream(specs, seed=4321):
    """Seeded variables in creation order per conv (yolo_v2.py:75-83): weight [k,k,cin,cout], biases,
    then (bn) scale, shift, rolling_mean, rolling_variance."""
    rng = np.random.default_rng(seed)
    parts = []
    for s in specs:
        k, ci, co = s["k"], s["cin"], s["cout"]
        gain = 2.0 if s["bn"] else 0.3
        parts.append(rng.normal(0, np.sqrt(gain / (k * k * ci)), (k, k, ci, co)).ravel())
        parts.append(rng.normal(0, 0.1, co))
        if s["bn"]:
            parts += [rng.uniform(0.5, 1.5, co) * 0.85, rng.normal(0, 0.1, co), rng.normal(0, 0.1, co), rng.uniform(0.5, 1.5, co)]
    return np.concatenate(parts).astype(F32)


def split_stream(specs, stream):
    out, ptr = [], 0
    for s in specs:
        k, ci, co = s["k"], s["cin"], s["cout"]
        d = {}
        n = k * k * ci * co
        d["weight"] = stream[ptr:ptr + n].reshape(k, k, ci, co); ptr += n
        names = ("biases", "scale", "shift", "mean", "variance") if s["bn"] else ("biases",)
        for nm in names:
            d[nm] = stream[ptr:ptr + co]; ptr += co
        out.append(d)
    assert ptr == len(stream)
    return out


def conv_layer(x, p, spec):
    """yolo_v2.py:73-89: conv SAME (no bias) -> batch_normalization(mean, var, shift, scale, 1e-5) -> + biases
    -> max(alpha*x, x); batch_norm=False: conv + biases."""
    import torch
    import torch.nn.functional as TF
    k = spec["k"]
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(np.transpose(p["weight"], (3, 2, 0, 1))))
    y = TF.conv2d(xt, wt, None, stride=1, padding=(k - 1) // 2).permute(0, 2, 3, 1).contiguous().numpy()   # :77
    if spec["bn"]:
        inv = (p["scale"] / np.sqrt(p["variance"] + F32(1e-5))).astype(F32)
        y = (y - p["mean"]) * inv + p["shift"]           # :84
        y = y + p["biases"]                              # :85
        y = np.maximum(F32(ALPHA) * y, y)                # :86
    else:
        y = y + p["biases"]                              # :88
    return y.astype(F32)


def pooling_layer(x):
    """yolo_v2.py:91-94: max_pool 2x2 stride 2 SAME (even sizes: plain 2x2 blocks)."""
    B, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    return x.reshape(B, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))


def reorg(x):
    """yolo_v2.py:97-113."""
    return np.concatenate([x[:, ::2, ::2, :], x[:, ::2, 1::2, :], x[:, 1::2, ::2, :], x[:, 1::2, 1::2, :]], axis=3)


def build_networks(params, x, specs=None):
    """yolo_v2.py:32-70.  x: [B,416,416,3] in [-1,1] -> logits [B,13,13,125]."""
    specs = specs or conv_specs()
    it = iter(range(len(specs)))

    def conv(t):
        i = next(it)
        return conv_layer(t, params[i], specs[i])

    net = pooling_layer(conv(x))
    net = pooling_layer(conv(net))
    net = conv(conv(conv(net)))
    net = pooling_layer(net)
    net = conv(conv(conv(net)))
    net = pooling_layer(net)
    net = conv(conv(conv(conv(net))))
    net16 = conv(net)
    net = pooling_layer(net16)
    for _ in range(6):
        net = conv(net)
    net24 = conv(net)
    net = reorg(conv(net16))
    net = np.concatenate([net, net24], 3)
    net = conv(net)
    return conv(net)


def get_boxes(boxes, cell_size=CELL_SIZE, box_per_cell=BOX_PER_CELL, anchor=ANCHOR):
    """Detector.py:91-104 (float32 sigmoid/exp, promoted to float64 by the int64 offset / list anchors)."""
    offset = np.transpose(np.reshape(np.array([np.arange(cell_size)] * cell_size * box_per_cell),
                                     [box_per_cell, cell_size, cell_size]), (1, 2, 0))
    boxes1 = np.stack([(1.0 / (1.0 + np.exp(-1.0 * boxes[:, :, :, 0])) + offset) / cell_size,
                       (1.0 / (1.0 + np.exp(-1.0 * boxes[:, :, :, 1])) + np.transpose(offset, (1, 0, 2))) / cell_size,
                       np.exp(boxes[:, :, :, 2]) * np.reshape(anchor[:5], [1, 1, 5]) / cell_size,
                       np.exp(boxes[:, :, :, 3]) * np.reshape(anchor[5:], [1, 1, 5]) / cell_size])
    return np.transpose(boxes1, (1, 2, 3, 0))


def calc_iou(b1, b2):
    """Detector.py:107-115 (float64)."""
    width = min(b1[0] + 0.5 * b1[2], b2[0] + 0.5 * b2[2]) - max(b1[0] - 0.5 * b1[2], b2[0] - 0.5 * b2[2])
    height = min(b1[1] + 0.5 * b1[3], b2[1] + 0.5 * b2[3]) - max(b1[1] - 0.5 * b1[3], b2[1] - 0.5 * b2[3])
    inter = 0 if (width <= 0 or height <= 0) else width * height
    return inter / (b1[2] * b1[3] + b2[2] * b2[3] - inter)


def decode(output, num_classes=20, cell_size=CELL_SIZE, box_per_cell=BOX_PER_CELL, image_size=IMAGE_SIZE):
    """Detector.py:53-62 for ONE image: -> boxes f64 [S,S,B,4] (cx,cy,w,h px), probs f32 [S,S,B,C]."""
    output = np.reshape(np.asarray(output, F32), [cell_size, cell_size, box_per_cell, 5 + num_classes])
    boxes = get_boxes(output[:, :, :, :4], cell_size, box_per_cell) * image_size            # :54-55
    confidence = 1.0 / (1.0 + np.exp(-1.0 * output[:, :, :, 4]))                              # :56-57
    confidence = np.tile(np.expand_dims(confidence, 3), (1, 1, 1, num_classes))              # :59
    classes = output[:, :, :, 5:]
    classes = np.exp(classes) / np.tile(np.expand_dims(np.sum(np.exp(classes), axis=3), axis=3), (1, 1, 1, num_classes))  # :61
    return boxes, (classes * confidence)                                                      # :62


def nms_from_decoded(boxes, probs, threshold=THRESHOLD, iou_thresh=0.5):
    """Detector.py:63-83 on decoded tensors.  Tie rule (the reference's argsort()[::-1] is unstable):
    stable ascending sort reversed = equal probs in DESCENDING candidate order.
    -> (class ids int, boxes f64 [m,4], probs f32 [m], candidate flat index [m])."""
    filter_probs = np.array(probs >= F32(threshold), dtype='bool')                            # :63
    fi = np.nonzero(filter_probs)                                                             # :64
    box_filter = boxes[fi[0], fi[1], fi[2]]                                                   # :65
    probs_filter = probs[filter_probs].copy()                                                 # :66
    classes_num = np.argmax(filter_probs, axis=3)[fi[0], fi[1], fi[2]]                        # :67
    flat = np.ravel_multi_index(fi, probs.shape)
    sort_num = np.argsort(probs_filter, kind="stable")[::-1]                                  # :68
    box_filter, probs_filter, classes_num, flat = box_filter[sort_num], probs_filter[sort_num], classes_num[sort_num], flat[sort_num]
    for i in range(len(probs_filter)):                                                        # :72-78
        if probs_filter[i] == 0:
            continue
        for j in range(i + 1, len(probs_filter)):
            if calc_iou(box_filter[i], box_filter[j]) > iou_thresh:
                probs_filter[j] = 0.0
    keep = probs_filter > 0                                                                   # :79
    return classes_num[keep], box_filter[keep], probs_filter[keep], flat[keep]


def calc_output(output, num_classes=20):
    """Detector.py:51-89 -> list of [class_name, cx, cy, w, h, prob]."""
    boxes, probs = decode(output, num_classes)
    cls, bx, pr, _ = nms_from_decoded(boxes, probs)
    return [[CLASSES[cls[i]], bx[i][0], bx[i][1], bx[i][2], bx[i][3], pr[i]] for i in range(len(pr))]
