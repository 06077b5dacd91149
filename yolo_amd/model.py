# coding=utf-8
"""`yolov3` with the interface of reference yolo_v3/model.py:14-153, executed by libyolo_hip.so.

Differences forced by eager execution (there is no TF graph / Saver): weights come from
`load_darknet_weights(path)` / `load_weight_stream(array)` (the Darknet file layout that the
reference's convert_weight.py + misc_utils.load_weights consume), tensors are torch-ROCm tensors.
"""
from __future__ import division, print_function

import ctypes

import numpy as np
import torch

from . import _lib, _tensor
from .utils import misc_utils


class yolov3(object):

    def __init__(self, class_num, anchors, batch_norm_decay=0.9, dtype="fp32", small_batch_plan=False):
        # model.py:16-19 (batch_norm_decay only matters for training; kept for signature parity).  small_batch_plan: bf16 only,
        # split-K launches for layers whose grid would leave most CUs idle (batch 1 - 4 latency; include/yolo_hip.h)
        self.class_num = class_num
        self.anchors = np.asarray(anchors, np.float32).reshape(-1, 2)
        self.batch_norm_decay = batch_norm_decay
        self.dtype = dtype
        self._dtype_code = _lib.DTYPES[dtype]
        self.small_batch_plan = bool(small_batch_plan)
        self._net = None
        self._ws = None
        self._post_ws = None
        self._det_ws = None
        self._device = None
        self.img_size = None

    @classmethod
    def from_cfg(cls, cfgfile, weights_file=None, dtype="fp32"):
        """Darknet-style construction (yolo-v3/002/darknet.py:150-153 `Darknet(cfgfile)` + `load_weights`): class count and
        anchors come from the .cfg, which must describe the YOLOv3 topology (yolo_amd/cfg.py); `net_size` = (width, height)."""
        from . import cfg as _cfg
        plan = _cfg.load_cfg(cfgfile)
        m = cls(plan["class_num"], plan["anchors"], dtype=dtype)
        m.net_size = (plan["width"], plan["height"])
        if weights_file is not None:
            m.load_darknet_weights(weights_file)
        return m

    # ---- weights (replaces tf.train.Saver.restore at test_single_image.py:35-36) -----------------
    def load_weight_stream(self, stream):
        """stream: float32 body of a Darknet .weights file (conv order = call order of forward)."""
        stream = np.ascontiguousarray(stream, dtype=np.float32)
        _tensor.device()
        self._free()
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().yolo_net_create(self.class_num, self._dtype_code, stream.ctypes.data,
                                              stream.size, ctypes.byref(h)))
        self._net = h
        self._device = torch.cuda.current_device()               # the packed weights live on this GPU
        if self.small_batch_plan:
            _lib.check(_lib.lib().yolo_net_set_small_batch_plan(self._net, 1))
        return self

    def _check_ready(self, x):
        """Shared argument validation of forward / forward_timed / detect: weights loaded, NHWC 3-channel input, and the
        current GPU is the one that holds the weights (a launch on another device would read them through peer access or fault)."""
        if self._net is None:
            raise RuntimeError("weights not loaded: call load_darknet_weights()/load_weight_stream() first")
        if torch.cuda.current_device() != self._device:
            raise RuntimeError("yolov3 weights were loaded on cuda:%d but the current device is cuda:%d"
                               % (self._device, torch.cuda.current_device()))
        if x is not None and (x.dim() != 4 or x.shape[3] != 3):
            raise ValueError("inputs must be [N, H, W, 3], got %s" % (tuple(x.shape),))

    def load_darknet_weights(self, weights_file):
        return self.load_weight_stream(misc_utils.read_darknet_weights(weights_file))

    def load_state(self, state):
        """state: path of an .npz written by `python -m yolo_amd.convert_weight` / `save_state`, or a dict of arrays
        named like the reference's TF variables (`yolov3/darknet53_body/Conv_3/BatchNorm/gamma`, ... - what a
        maintainer dumps from their converted checkpoint).  Replaces Saver.restore (test_single_image.py:35-36)."""
        from . import weights
        if not isinstance(state, dict):
            with np.load(state, allow_pickle=False) as z:
                state = {k: z[k] for k in z.files}
        return self.load_weight_stream(weights.state_to_stream(state, self.class_num))

    def _free(self):
        if getattr(self, "_net", None) is not None:
            _lib.lib().yolo_net_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    # ---- model.py:21-67 ------------------------------------------------------------------------------
    def forward(self, inputs, is_training=False, reuse=False):
        if is_training:
            raise ValueError("yolo_amd implements the inference path only (is_training=False)")
        x = _tensor.as_dev_f32(inputs) if self._net is not None else None
        self._check_ready(x)
        B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        self.img_size = (H, W)                                   # model.py:22
        L = _lib.lib()
        need = L.yolo_net_workspace_bytes(self._net, B, H, W)
        if need == 0:
            raise ValueError("unsupported input shape %s (H, W must be multiples of 32)" % (tuple(x.shape),))
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = _tensor.workspace(need)
        nout = 3 * (5 + self.class_num)
        fms = [_tensor.empty((B, H // r, W // r, nout)) for r in (32, 16, 8)]
        _lib.check(L.yolo_net_forward(self._net, x.data_ptr(), B, H, W, self._ws.data_ptr(), self._ws.numel(),
                                      fms[0].data_ptr(), fms[1].data_ptr(), fms[2].data_ptr(), _tensor.stream_ptr()))
        return fms[0], fms[1], fms[2]

    def forward_timed(self, inputs):
        """Measurement aid: forward with per-layer hipEvent timing -> (fms, ms[75], [(cin,cout,k,stride)]);
        `self.layer_kernels` then holds the kernel instantiation launched for every layer."""
        x = _tensor.as_dev_f32(inputs) if self._net is not None else None
        self._check_ready(x)
        B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        self.img_size = (H, W)
        L = _lib.lib()
        need = L.yolo_net_workspace_bytes(self._net, B, H, W)
        if need == 0:
            raise ValueError("unsupported input shape %s (H, W must be multiples of 32)" % (tuple(x.shape),))
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = _tensor.workspace(need)
        nout = 3 * (5 + self.class_num)
        fms = [_tensor.empty((B, H // r, W // r, nout)) for r in (32, 16, 8)]
        nl = L.yolo_net_num_layers(self._net)
        ms = np.zeros(nl, np.float32)
        _lib.check(L.yolo_net_forward_timed(self._net, x.data_ptr(), B, H, W, self._ws.data_ptr(), self._ws.numel(),
                                            fms[0].data_ptr(), fms[1].data_ptr(), fms[2].data_ptr(),
                                            _tensor.stream_ptr(), ms.ctypes.data, nl))
        info = []
        for i in range(nl):
            v = [ctypes.c_int() for _ in range(4)]
            _lib.check(L.yolo_net_layer_info(self._net, i, *[ctypes.byref(t) for t in v]))
            info.append(tuple(t.value for t in v))
        self.layer_kernels = [L.yolo_net_layer_kernel(self._net, i).decode() for i in range(nl)]
        return fms, ms, info

    def detect(self, inputs, max_boxes=30, score_thresh=0.4, iou_thresh=0.5):
        """The hot chain of test_single_image.py:31-40 for a batch, per image: forward -> predict ->
        conf*prob -> per-class NMS.  Defaults are that script's thresholds (:39).
        -> (boxes [N,C*max_boxes,4], scores, labels int32, keep-index int32, counts int32 [N]).
        ONE library call (yolo_net_detect): identical results to postprocess(forward(inputs)) and to the
        reference's separate calls (detect_unfused), without the [N,n,C] probs / scores tensors; the feature maps
        stay in the workspace and every lane post-processes its own images."""
        x = _tensor.as_dev_f32(inputs) if self._net is not None else None
        self._check_ready(x)
        B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        self.img_size = (H, W)
        C = self.class_num
        anc = np.ascontiguousarray(self.anchors, np.float32)
        if anc.shape != (9, 2):
            raise ValueError("detect needs 9 anchors, got %s" % (anc.shape,))
        L = _lib.lib()
        if C > 250:
            # the fused post-processing keeps one box's class scores in an LDS tile (<= 250 classes); the reference has no such
            # limit (nms_utils.py:7), so wider heads take the reference's own sequence of calls on the GPU: forward -> predict
            # -> conf*prob -> per-image gpu_nms (same results, the [N, n, C] score tensor exists)
            return self.detect_unfused(x, max_boxes, score_thresh, iou_thresh)
        need = L.yolo_net_detect_workspace_bytes(self._net, B, H, W, int(max_boxes))
        if need == 0:
            raise ValueError("detect: bad sizes B=%d H=%d W=%d C=%d max_boxes=%d (H, W must be multiples of 32)"
                             % (B, H, W, C, max_boxes))
        if self._det_ws is None or self._det_ws.numel() < need or self._det_ws.device != x.device:
            self._det_ws = _tensor.workspace(need)
        cap = C * int(max_boxes)
        ob = _tensor.empty((B, cap, 4))
        osc = _tensor.empty((B, cap))
        ol = _tensor.empty((B, cap), torch.int32)
        oi = _tensor.empty((B, cap), torch.int32)
        oc = _tensor.empty((B,), torch.int32)
        _lib.check(L.yolo_net_detect(self._net, x.data_ptr(), B, H, W, anc.ctypes.data, int(max_boxes), float(score_thresh),
                                     float(iou_thresh), ob.data_ptr(), osc.data_ptr(), ol.data_ptr(), oi.data_ptr(),
                                     oc.data_ptr(), self._det_ws.data_ptr(), self._det_ws.numel(), _tensor.stream_ptr()))
        return ob, osc, ol, oi, oc

    def postprocess(self, feature_maps, max_boxes=30, score_thresh=0.4, iou_thresh=0.5):
        fm1, fm2, fm3 = [_tensor.as_dev_f32(f) for f in feature_maps]
        B = int(fm1.shape[0])
        if self.img_size is None:
            self.img_size = (int(fm1.shape[1]) * 32, int(fm1.shape[2]) * 32)
        H, W = self.img_size
        C = self.class_num
        anc = np.ascontiguousarray(self.anchors, np.float32)
        if anc.shape != (9, 2):
            raise ValueError("detect needs 9 anchors, got %s" % (anc.shape,))
        L = _lib.lib()
        cap = C * int(max_boxes)
        ob = _tensor.empty((B, cap, 4))
        osc = _tensor.empty((B, cap))
        ol = _tensor.empty((B, cap), torch.int32)
        oi = _tensor.empty((B, cap), torch.int32)
        oc = _tensor.empty((B,), torch.int32)
        need = L.yolo_detect_post_workspace_bytes(B, H, W, C, int(max_boxes))
        if need == 0:
            raise ValueError("detect: bad sizes B=%d H=%d W=%d C=%d max_boxes=%d" % (B, H, W, C, max_boxes))
        if self._post_ws is None or self._post_ws.numel() < need or self._post_ws.device != fm1.device:
            self._post_ws = _tensor.workspace(need)
        _lib.check(L.yolo_detect_post(fm1.data_ptr(), fm2.data_ptr(), fm3.data_ptr(), B, H, W, C, anc.ctypes.data,
                                      int(max_boxes), float(score_thresh), float(iou_thresh), ob.data_ptr(),
                                      osc.data_ptr(), ol.data_ptr(), oi.data_ptr(), oc.data_ptr(),
                                      self._post_ws.data_ptr(), self._post_ws.numel(), _tensor.stream_ptr()))
        return ob, osc, ol, oi, oc

    def detect_unfused(self, inputs, max_boxes=30, score_thresh=0.4, iou_thresh=0.5):
        """detect() written with the reference's separate calls (predict, conf*prob, per-image gpu_nms)."""
        from .utils.nms_utils import gpu_nms_batched
        fms = self.forward(inputs)
        boxes, _, _, scores = self.predict(fms, return_scores=True)
        return gpu_nms_batched(boxes, scores, self.class_num, max_boxes, score_thresh, iou_thresh)

    # ---- model.py:69-108 -----------------------------------------------------------------------------
    def reorg_layer(self, feature_map, anchors):
        if self.img_size is None:
            raise RuntimeError("reorg_layer needs img_size recorded by forward() (model.py:22,78)")
        fm = _tensor.as_dev_f32(feature_map)
        B, gh, gw = int(fm.shape[0]), int(fm.shape[1]), int(fm.shape[2])
        anc = np.ascontiguousarray(anchors, np.float32).reshape(3, 2)
        boxes = _tensor.empty((B, gh, gw, 3, 4))
        _lib.check(_lib.lib().yolo_reorg_boxes(fm.data_ptr(), B, gh, gw, self.img_size[0], self.img_size[1],
                                               self.class_num, anc.ctypes.data, boxes.data_ptr(),
                                               _tensor.stream_ptr()))
        v = fm.view(B, gh, gw, 3, 5 + self.class_num)                         # model.py:81
        conf_logits, prob_logits = v[..., 4:5], v[..., 5:]                    # model.py:87
        gx = torch.arange(gw, dtype=torch.float32, device=fm.device)
        gy = torch.arange(gh, dtype=torch.float32, device=fm.device)
        x_y_offset = torch.stack(torch.meshgrid(gx, gy, indexing="xy"), dim=-1).reshape(gh, gw, 1, 2)   # :90-97
        return x_y_offset, boxes, conf_logits, prob_logits

    # ---- model.py:110-153 ----------------------------------------------------------------------------
    def predict(self, feature_maps, return_scores=False):
        """-> (boxes [N,n,4] x0y0x1y1, confs [N,n,1], probs [N,n,C]); with return_scores also
        scores = confs * probs (the product the caller forms at test_single_image.py:38), fused."""
        fm1, fm2, fm3 = [_tensor.as_dev_f32(f) for f in feature_maps]
        B = int(fm1.shape[0])
        if self.img_size is None:
            self.img_size = (int(fm1.shape[1]) * 32, int(fm1.shape[2]) * 32)
        H, W = self.img_size
        C = self.class_num
        n = 3 * sum((H // r) * (W // r) for r in (32, 16, 8))
        boxes = _tensor.empty((B, n, 4))
        confs = _tensor.empty((B, n, 1))
        probs = _tensor.empty((B, n, C))
        scores = _tensor.empty((B, n, C)) if return_scores else None
        anc = np.ascontiguousarray(self.anchors, np.float32)
        if anc.shape != (9, 2):
            raise ValueError("predict needs 9 anchors, got %s" % (anc.shape,))
        _lib.check(_lib.lib().yolo_decode(fm1.data_ptr(), fm2.data_ptr(), fm3.data_ptr(), B, H, W, C,
                                          anc.ctypes.data, boxes.data_ptr(), confs.data_ptr(), probs.data_ptr(),
                                          _tensor.ptr(scores), _tensor.stream_ptr()))
        if return_scores:
            return boxes, confs, probs, scores
        return boxes, confs, probs
