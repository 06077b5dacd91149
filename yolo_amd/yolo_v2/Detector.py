# coding: utf-8
"""`Detector` with the interface of reference yolo_v2/Detector.py:14-115, executed by libyolo_hip.so.

`calc_output` (sigmoid / exp / softmax decode, threshold, class-agnostic greedy NMS) runs as two HIP
kernels (yolo2_decode, yolo2_nms) instead of numpy + a Python double loop.  Image file I/O
and drawing (cv2.imread, draw, image_detect, video_detect) are outside the hot path; the cv2.resize / cvtColor /
normalisation of `detect` (Detector.py:37-39) run as one GPU kernel (yolo_preprocess) for uint8 images.
"""
import numpy as np
import torch

from .. import _lib, _tensor
from .yolo import config as cfg


class Detector(object):

    def __init__(self, yolo, weights_file=None):
        # Detector.py:15-24.  weights_file: optional .npy holding the variable stream (the reference
        # restores a TF checkpoint here, :28-29).
        self.yolo = yolo
        self.classes = cfg.CLASSES
        self.num_classes = len(self.classes)
        self.image_size = cfg.IMAGE_SIZE
        self.cell_size = cfg.CELL_SIZE
        self.batch_size = cfg.BATCH_SIZE
        self.box_per_cell = cfg.BOX_PRE_CELL
        self.threshold = cfg.THRESHOLD
        self.anchor = cfg.ANCHOR
        self.iou_threshold = 0.5          # literal at Detector.py:76
        if weights_file is not None:
            print('Restore weights from: ' + weights_file)
            self.yolo.load_variables(np.load(weights_file, allow_pickle=False))

    # ---- Detector.py:31-49 ------------------------------------------------------------------------
    def detect(self, image):
        """image: uint8 [h, w, 3] in BGR as cv2.imread returns it (any size: resized on the GPU, yolo_preprocess), or
        a float array [image_size, image_size, 3] in BGR with values 0..255 -> list of
        [class_name, x_center, y_center, w, h, prob] in the image's own pixels."""
        image = np.asarray(image)
        image_h, image_w, _ = image.shape
        if image.dtype == np.uint8:
            from ..preprocess import preprocess
            image = preprocess(image, (self.image_size, self.image_size), v2=True)         # :37-40
        else:
            if image_h != self.image_size or image_w != self.image_size:
                raise ValueError("detect: float images must already be %dx%d (pass uint8 to have them resized)"
                                 % (self.image_size, self.image_size))
            image = image[:, :, ::-1].astype(np.float32)                    # BGR -> RGB (:38)
            image = image / 255.0 * 2.0 - 1.0                               # :39
            image = np.reshape(image, [1, self.image_size, self.image_size, 3])
        output = self.yolo.build_networks(image)                        # :41
        results = self.calc_output(output)
        sx, sy = 1.0 * image_w / self.image_size, 1.0 * image_h / self.image_size
        for r in results:                                               # back to the image's pixels (:43-48)
            r[1] *= sx
            r[2] *= sy
            r[3] *= sx
            r[4] *= sy
        return results

    # ---- Detector.py:51-89 ------------------------------------------------------------------------
    def decode(self, output):
        """logits [B,S,S,NB*(5+C)] -> boxes float64 [B,S,S,NB,4] (cx,cy,w,h in input pixels, :54-55, :91-104),
        probs float32 [B,S,S,NB,C] (= softmax(classes) * sigmoid(conf), :56-62)."""
        S, NB, C = self.cell_size, self.box_per_cell, self.num_classes
        t = _tensor.as_dev_f32(output).reshape(-1, S, S, NB * (5 + C))
        B = int(t.shape[0])
        boxes = _tensor.empty((B, S, S, NB, 4), torch.float64)
        probs = _tensor.empty((B, S, S, NB, C))
        anchors = np.ascontiguousarray(self.anchor, np.float64)
        _lib.check(_lib.lib().yolo2_decode(t.data_ptr(), B, S, NB, C, self.image_size, anchors.ctypes.data,
                                           boxes.data_ptr(), probs.data_ptr(), _tensor.stream_ptr()))
        return boxes, probs

    def nms(self, boxes, probs, max_det=None):
        """Class-agnostic greedy NMS of Detector.py:63-83 per image on decoded tensors.
        -> (counts int32 [B], class ids int32 [B,max_det], boxes f64 [B,max_det,4], probs f32 [B,max_det],
        flat candidate index int32 [B,max_det]); counts[i] <= max_det rows of image i are valid.
        Normally at most one candidate per box survives (same box -> IoU 1 > 0.5), so the default of n rows suffices.
        Degenerate boxes (zero area, inf / NaN from an exp overflow) have IoU NaN with themselves, so several classes of one
        box can survive as they do in the reference (Detector.py:72-83, no cap): with the default max_det the call is then
        repeated with n * C rows, which always suffice; with an explicit max_det the lists are cut there and
        `self.last_truncated` (bool tensor [B]) says for which images."""
        S, NB, C = self.cell_size, self.box_per_cell, self.num_classes
        n = S * S * NB
        bx = boxes if isinstance(boxes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(boxes, np.float64))
        bx = bx.to(device=_tensor.device(), dtype=torch.float64).contiguous().reshape(-1, n, 4)
        pr = _tensor.as_dev_f32(probs).reshape(-1, n, C)
        B = int(pr.shape[0])
        L = _lib.lib()
        ws = _tensor.workspace(L.yolo2_nms_workspace_bytes(B, n, C))

        def run(rows):
            counts = _tensor.empty((B,), torch.int32)
            trunc = _tensor.empty((B,), torch.int32)
            cls = _tensor.empty((B, rows), torch.int32)
            ob = _tensor.empty((B, rows, 4), torch.float64)
            op = _tensor.empty((B, rows))
            flat = _tensor.empty((B, rows), torch.int32)
            _lib.check(L.yolo2_nms(bx.data_ptr(), pr.data_ptr(), B, n, C, self.threshold, self.iou_threshold, rows,
                                   counts.data_ptr(), cls.data_ptr(), ob.data_ptr(), op.data_ptr(), flat.data_ptr(),
                                   trunc.data_ptr(), ws.data_ptr(), ws.numel(), _tensor.stream_ptr()))
            return (counts, cls, ob, op, flat), trunc
        out, trunc = run(int(max_det or n))
        if max_det is None and bool(trunc.any().item()):
            out, trunc = run(n * C)
        self.last_truncated = trunc != 0
        return out

    def calc_output(self, output):
        """ONE image's logits (any shape with 13*13*125 values) -> [[class_name, cx, cy, w, h, prob], ...]."""
        S, NB, C = self.cell_size, self.box_per_cell, self.num_classes
        t = _tensor.as_dev_f32(output).reshape(1, S, S, NB * (5 + C))    # :53 (drops the batch dim)
        return self.calc_output_batched(t)[0]

    def calc_output_batched(self, output):
        """Per-image calc_output for a batch of logits -> list (per image) of result lists."""
        boxes, probs = self.decode(output)
        counts, cls, ob, op, _ = self.nms(boxes, probs)                           # repeats itself with n * C rows when n do not suffice
        counts, cls, ob, op = counts.cpu().numpy(), cls.cpu().numpy(), ob.cpu().numpy(), op.cpu().numpy()
        out = []
        for b in range(len(counts)):
            out.append([[self.classes[cls[b, i]], ob[b, i, 0], ob[b, i, 1], ob[b, i, 2], ob[b, i, 3], op[b, i]]
                        for i in range(int(counts[b]))])
        return out

    # ---- Detector.py:91-104 / :107-115: host-side helpers kept for API parity (tiny, numpy) ---------
    def get_boxes(self, boxes):
        """boxes [S,S,NB,4] logits -> normalised (cx,cy,w,h): one image through the decode kernel."""
        S, NB, C = self.cell_size, self.box_per_cell, self.num_classes
        t = torch.zeros((1, S, S, NB, 5 + C), dtype=torch.float32)
        t[0, ..., :4] = torch.as_tensor(np.asarray(boxes, np.float32))
        b, _ = self.decode(t)
        return b[0].cpu().numpy() / self.image_size

    def calc_iou(self, box1, box2):
        """IoU of two (cx, cy, w, h) boxes, float64 like the reference (:107-115)."""
        def overlap(c1, s1, c2, s2):
            return min(c1 + 0.5 * s1, c2 + 0.5 * s2) - max(c1 - 0.5 * s1, c2 - 0.5 * s2)
        width = overlap(box1[0], box1[2], box2[0], box2[2])
        height = overlap(box1[1], box1[3], box2[1], box2[3])
        intersection = width * height if (width > 0 and height > 0) else 0
        return intersection / (box1[2] * box1[3] + box2[2] * box2[3] - intersection)
