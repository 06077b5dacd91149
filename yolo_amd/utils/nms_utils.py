# coding: utf-8
"""gpu_nms / py_nms / cpu_nms with the signatures of reference yolo_v3/utils/nms_utils.py, all
executed by the HIP NMS kernels (yolo_nms in include/yolo_hip.h).  `gpu_nms_batched` is the
per-image form the throughput path uses (no host synchronisation)."""
from __future__ import division, print_function

import numpy as np
import torch

from .. import _lib, _tensor


def _run(boxes, scores, B, n, C, max_boxes, score_thresh, iou_thresh, mode):
    L = _lib.lib()
    cap = C * max_boxes
    ob = _tensor.empty((B, cap, 4))
    osc = _tensor.empty((B, cap))
    ol = _tensor.empty((B, cap), torch.int32)
    oi = _tensor.empty((B, cap), torch.int32)
    oc = _tensor.empty((B,), torch.int32)
    need = L.yolo_nms_workspace_bytes(B, n, C, max_boxes)
    if need == 0:
        raise ValueError("nms: bad sizes B=%d n=%d C=%d max_boxes=%d" % (B, n, C, max_boxes))
    ws = _tensor.workspace(need)
    _lib.check(L.yolo_nms(boxes.data_ptr(), scores.data_ptr(), B, n, C, int(max_boxes), float(score_thresh),
                          float(iou_thresh), mode, ob.data_ptr(), osc.data_ptr(), ol.data_ptr(), oi.data_ptr(),
                          oc.data_ptr(), ws.data_ptr(), ws.numel(), _tensor.stream_ptr()))
    return ob, osc, ol, oi, oc


def gpu_nms_batched(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
    """Per-image NMS for a batch: boxes [B,n,4], scores [B,n,C] ->
    (boxes [B,C*max_boxes,4], scores [B,C*max_boxes], labels int32, keep-index int32, counts int32 [B]).
    Row r of image i is valid iff r < counts[i].  Equals B independent batch-1 gpu_nms calls."""
    b = _tensor.as_dev_f32(boxes)
    s = _tensor.as_dev_f32(scores)
    if b.dim() != 3 or b.shape[2] != 4 or s.dim() != 3 or s.shape[2] != num_classes or s.shape[:2] != b.shape[:2]:
        raise ValueError("gpu_nms_batched: boxes [B,n,4] and scores [B,n,%d] expected, got %s / %s"
                         % (num_classes, tuple(b.shape), tuple(s.shape)))
    return _run(b, s, int(b.shape[0]), int(b.shape[1]), num_classes, max_boxes, score_thresh, iou_thresh,
                _lib.YOLO_NMS_TF)


def gpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
    """nms_utils.py:7-37.  boxes [1,n,4], scores [1,n,C] -> (boxes [m,4], score [m], label [m] int32).

    Like the reference the batch dimension is flattened (nms_utils.py:20-21), so N > 1 mixes images;
    use gpu_nms_batched for per-image results."""
    b = _tensor.as_dev_f32(boxes).reshape(-1, 4)
    s = _tensor.as_dev_f32(scores).reshape(-1, num_classes)
    n = int(b.shape[0])
    if n == 0 or int(s.shape[0]) != n:
        if n == 0 and s.shape[0] == 0:
            return (_tensor.empty((0, 4)), _tensor.empty((0,)), _tensor.empty((0,), torch.int32))
        raise ValueError("gpu_nms: %d boxes but %d score rows" % (n, int(s.shape[0])))
    ob, osc, ol, _, oc = _run(b, s, 1, n, num_classes, max_boxes, score_thresh, iou_thresh, _lib.YOLO_NMS_TF)
    m = int(oc.item())                       # variable-length result: the one host round trip (== sess.run)
    return ob[0, :m], osc[0, :m], ol[0, :m]


def py_nms(boxes, scores, max_boxes=50, iou_thresh=0.5):
    """nms_utils.py:40-64: boxes [-1,4], scores [-1] -> python list of kept indices."""
    boxes = np.asarray(boxes)
    scores = np.asarray(scores)
    assert boxes.shape[1] == 4 and len(scores.shape) == 1            # nms_utils.py:42
    n = int(scores.shape[0])
    if n == 0:
        return []
    b = _tensor.as_dev_f32(boxes)
    s = _tensor.as_dev_f32(scores).reshape(n, 1)
    mb = min(int(max_boxes), n)
    if mb <= 0:
        return []
    _, _, _, oi, oc = _run(b, s, 1, n, 1, mb, float("-inf"), iou_thresh, _lib.YOLO_NMS_PY)
    return [int(v) for v in oi[0, :int(oc.item())].cpu().numpy()]


def cpu_nms(boxes, scores, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
    """nms_utils.py:67-90: numpy in, numpy out; (None, None, None) when nothing passes."""
    boxes = np.asarray(boxes).reshape(-1, 4)                         # nms_utils.py:69
    scores = np.asarray(scores).reshape(-1, num_classes)             # nms_utils.py:70
    n = int(boxes.shape[0])
    if n == 0:
        return None, None, None
    ob, osc, ol, _, oc = _run(_tensor.as_dev_f32(boxes), _tensor.as_dev_f32(scores), 1, n, num_classes,
                              max_boxes, score_thresh, iou_thresh, _lib.YOLO_NMS_PY)
    m = int(oc.item())
    if m == 0:
        return None, None, None                                      # nms_utils.py:85-86
    return ob[0, :m].cpu().numpy(), osc[0, :m].cpu().numpy(), ol[0, :m].cpu().numpy()
