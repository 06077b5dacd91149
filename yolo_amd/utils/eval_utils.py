# coding: utf-8
"""Recall / precision bookkeeping with the interface of the reference's yolo_v3/utils/eval_utils.py:7-162
(the step right after the hot path).  The per-image NMS runs on the GPU (`cpu_nms` = the numpy
semantics of nms_utils.py:40-90 as a kernel, `gpu_nms` = the TF-op semantics); `evaluate_on_gpu` sends ALL images
through one batched NMS call instead of one sess.run per image.  The counting itself is host-side Python like the
reference's and is pinned by vectors generated from the reference's own evaluate_on_cpu (tests/golden/g5_eval.npz)."""
from __future__ import division, print_function

from collections import Counter

import numpy as np

from . import nms_utils


def calc_iou(pred_boxes, true_boxes):
    """eval_utils.py:7-21: corner boxes [N,4] x [V,4] -> IoU [N,V] (no +1, denominator + 1e-10)."""
    p = np.asarray(pred_boxes)[:, None, :]
    t = np.asarray(true_boxes)[None, :, :]
    wh = np.maximum(np.minimum(p[..., 2:], t[..., 2:]) - np.maximum(p[..., :2], t[..., :2]), 0.)
    inter = wh[..., 0] * wh[..., 1]
    p_wh = p[..., 2:] - p[..., :2]
    t_wh = t[..., 2:] - t[..., :2]
    return inter / (p_wh[..., 0] * p_wh[..., 1] + t_wh[..., 0] * t_wh[..., 1] - inter + 1e-10)


def _ground_truth(y_true, i):
    """Labels and corner boxes of image i from the three y_true maps [N,g,g,3,5+C] (eval_utils.py:33-49).
    Like the reference this needs at least one labelled box per image (IndexError otherwise)."""
    labels, boxes = [], []
    for fm in y_true:
        cell = np.asarray(fm[i])
        probs = cell[..., 5:]
        has_obj = probs.sum(axis=-1) > 0
        labels += np.argmax(probs[has_obj], axis=-1).tolist()
        boxes += cell[..., 0:4][has_obj].tolist()                      # (x_center, y_center, w, h), python floats
    boxes = np.array(boxes)
    if boxes.ndim != 2:
        raise IndexError("image %d has no labelled box: too many indices for array (as in the reference, "
                         "eval_utils.py:47)" % i)
    corners = np.empty_like(boxes)
    corners[:, 0:2] = boxes[:, 0:2] - boxes[:, 2:4] / 2.
    corners[:, 2:4] = corners[:, 0:2] + boxes[:, 2:4]
    return labels, corners


def _count_image(pred_boxes, pred_confs, pred_labels, true_boxes, true_labels, iou_thresh, pred_count, tp_count):
    """eval_utils.py:63-83: every prediction counts for its class; a prediction claims its best-IoU ground truth
    when IoU > iou_thresh and the labels agree; a ground truth counts once (the most confident claimant keeps it)."""
    iou = calc_iou(pred_boxes, true_boxes)
    best = np.argmax(iou, axis=-1)
    held = {}                                                         # ground-truth index -> confidence holding it
    for k, g in enumerate(best.tolist()):
        lab = int(pred_labels[k])
        pred_count[lab] += 1
        if iou[k, g] > iou_thresh and true_labels[g] == lab:
            if g not in held or pred_confs[k] > held[g]:
                held[g] = pred_confs[k]
    for g in held:
        tp_count[true_labels[g]] += 1


def _finish(tp_count, true_count, pred_count, calc_now):
    if not calc_now:
        return tp_count, true_count, pred_count
    tp = sum(tp_count.values())
    return tp / (sum(true_count.values()) + 1e-6), tp / (sum(pred_count.values()) + 1e-6)     # recall, precision


def _evaluate(per_image_nms, y_pred, y_true, num_classes, calc_now, iou_thresh):
    num_images = y_true[0].shape[0]
    true_count = {c: 0 for c in range(num_classes)}
    pred_count = {c: 0 for c in range(num_classes)}
    tp_count = {c: 0 for c in range(num_classes)}
    for i in range(num_images):
        true_labels, true_boxes = _ground_truth(y_true, i)
        for cls, cnt in Counter(true_labels).items():
            true_count[cls] += cnt
        boxes, confs, labels = per_image_nms(i)
        if labels is None or len(labels) == 0:
            continue
        _count_image(boxes, confs, labels, true_boxes, true_labels, iou_thresh, pred_count, tp_count)
    return _finish(tp_count, true_count, pred_count, calc_now)


def evaluate_on_cpu(y_pred, y_true, num_classes, calc_now=True, score_thresh=0.5, iou_thresh=0.5, _nms=None):
    """eval_utils.py:23-88.  y_pred = (boxes [N,n,4], confs [N,n,1], probs [N,n,C]) as `predict` returns them
    (numpy); y_true = three arrays [N,g,g,3,5+C].  NMS = cpu_nms semantics (max_boxes 50 per class, the same
    iou_thresh for suppression and matching).  `_nms` (tests) replaces the NMS callable."""
    nms = _nms or nms_utils.cpu_nms
    pb, pc, pp = [np.asarray(a) for a in y_pred]

    def per_image(i):
        return nms(pb[i:i + 1], pc[i:i + 1] * pp[i:i + 1], num_classes, score_thresh=score_thresh, iou_thresh=iou_thresh)
    return _evaluate(per_image, y_pred, y_true, num_classes, calc_now, iou_thresh)


class gpu_nms_op(object):
    """Stands in for the TF op `gpu_nms(pred_boxes_flag, pred_scores_flag, class_num)` that eval.py:93-95 builds once and
    passes to evaluate_on_gpu: it only carries the NMS parameters (defaults of nms_utils.py:7)."""

    def __init__(self, num_classes, max_boxes=50, score_thresh=0.5, iou_thresh=0.5):
        self.num_classes, self.max_boxes = int(num_classes), int(max_boxes)
        self.score_thresh, self.iou_thresh = float(score_thresh), float(iou_thresh)


def evaluate_on_gpu(sess, gpu_nms_op_, pred_boxes_flag, pred_scores_flag, y_pred, y_true, num_classes, calc_now=True,
                    score_thresh=0.5, iou_thresh=0.5):
    """eval_utils.py:91-162 with the reference's positional signature.  `sess` and the two placeholder arguments have
    no meaning here and are ignored; `gpu_nms_op_` is a `gpu_nms_op` (or None for the defaults).  As in the
    reference the NMS thresholds are those of the op, `iou_thresh` is the matching threshold, and `score_thresh`
    is unused.  All images go through ONE batched NMS call (yolo_nms, mode TF)."""
    op = gpu_nms_op_ if gpu_nms_op_ is not None else gpu_nms_op(num_classes)
    if op.num_classes != num_classes:
        raise ValueError("gpu_nms_op was built for %d classes, evaluate_on_gpu called with %d" % (op.num_classes, num_classes))
    pb, pc, pp = y_pred
    if isinstance(pc, np.ndarray):
        scores = np.asarray(pc) * np.asarray(pp)                       # pred_confs * pred_probs (:127)
    else:
        scores = pc * pp
    ob, osc, ol, _, oc = nms_utils.gpu_nms_batched(pb, scores, num_classes, op.max_boxes, op.score_thresh, op.iou_thresh)
    ob, osc, ol, oc = ob.cpu().numpy(), osc.cpu().numpy(), ol.cpu().numpy(), oc.cpu().numpy()

    def per_image(i):
        m = int(oc[i])
        return ob[i, :m], osc[i, :m], ol[i, :m]
    return _evaluate(per_image, y_pred, y_true, num_classes, calc_now, iou_thresh)
