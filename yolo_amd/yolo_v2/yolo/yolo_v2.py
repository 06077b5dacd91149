# coding: utf-8
"""`yolo_v2` with the interface of reference yolo_v2/yolo/yolo_v2.py:7-113 (inference part), executed
by libyolo_hip.so.

The reference builds a TF graph in __init__ (`self.images` placeholder -> `self.logits`) and restores the
variables from a checkpoint; here the variables come from `load_variables(stream)` (the TF variables
concatenated in creation order, yolo_v2.py:75-83) and `build_networks(inputs)` runs eagerly as ONE fused
plan (yolo2_net_forward: 23 convs, 5 max-pools, reorg + concat fused into conv epilogues).
`conv_layer` / `pooling_layer` / `reorg` are the layer-level drop-ins (one HIP launch each).
"""
import ctypes

import numpy as np

from ... import _lib, _tensor
from . import config as cfg


def conv_shapes(num_class=len(cfg.CLASSES), box_per_cell=cfg.BOX_PRE_CELL):
    """[(name, [k, k, cin, cout], batch_norm)] in call order of build_networks (yolo_v2.py:34-68)."""
    s = [("0_conv", 3, 3, 32), ("2_conv", 3, 32, 64), ("4_conv", 3, 64, 128), ("5_conv", 1, 128, 64),
         ("6_conv", 3, 64, 128), ("8_conv", 3, 128, 256), ("9_conv", 1, 256, 128), ("10_conv", 3, 128, 256),
         ("12_conv", 3, 256, 512), ("13_conv", 1, 512, 256), ("14_conv", 3, 256, 512), ("15_conv", 1, 512, 256),
         ("16_conv", 3, 256, 512), ("18_conv", 3, 512, 1024), ("19_conv", 1, 1024, 512), ("20_conv", 3, 512, 1024),
         ("21_conv", 1, 1024, 512), ("22_conv", 3, 512, 1024), ("23_conv", 3, 1024, 1024), ("24_conv", 3, 1024, 1024),
         ("26_conv", 1, 512, 64), ("29_conv", 3, 1280, 1024)]
    out = [(n, [k, k, ci, co], True) for n, k, ci, co in s]
    out.append(("30_conv", [1, 1, 1024, box_per_cell * (num_class + 5)], False))
    return out


def split_variables(stream, num_class=len(cfg.CLASSES), box_per_cell=cfg.BOX_PRE_CELL):
    """Variable stream -> {name: {weight, biases[, scale, shift, rolling_mean, rolling_variance]}}."""
    stream = np.asarray(stream, np.float32)
    out, ptr = {}, 0
    for name, shape, bn in conv_shapes(num_class, box_per_cell):
        n = int(np.prod(shape))
        d = {"weight": stream[ptr:ptr + n].reshape(shape)}
        ptr += n
        for v in (("biases", "scale", "shift", "rolling_mean", "rolling_variance") if bn else ("biases",)):
            d[v] = stream[ptr:ptr + shape[3]]
            ptr += shape[3]
        out[name] = d
    if ptr != stream.size:
        raise ValueError("variable stream has %d floats, the architecture needs %d" % (stream.size, ptr))
    return out


def darknet_stream_to_variables(stream, num_class=len(cfg.CLASSES), box_per_cell=cfg.BOX_PRE_CELL):
    """Float32 body of a Darknet yolov2 `.weights` file (after the header; e.g. yolov2-voc.weights) -> the TF variable stream
    `load_variables` / yolo2_net_create take.  Darknet stores, per conv in cfg order (= the call order of build_networks:
    the 20 trunk convs, the passthrough 1x1 conv, the two head convs: 23 in all), [biases, scales, rolling_mean, rolling_variance]
    (batch-norm convs) or [biases], then the kernel as [Cout, Cin, kh, kw] - the layout of yolo_v3/utils/misc_utils.py:45-96
    and yolo-v3/002/darknet.py:212-265.  The reference's conv_layer (yolo_v2.py:75-86) has BOTH a BN shift and an extra
    `biases` variable; Darknet has one: it becomes `shift`, `biases` is zero.  (The reference's reorg channel order,
    yolo_v2.py:108-112, is its own: a Darknet-trained passthrough layer keeps Darknet's order, which this build does not
    reproduce - load such weights for the trunk, not for bit-parity with Darknet's yolov2.)"""
    stream = np.asarray(stream, np.float32)
    parts, ptr = [], 0
    for name, (k, _, ci, co), bn in conv_shapes(num_class, box_per_cell):
        nb = 4 * co if bn else co
        if ptr + nb + k * k * ci * co > stream.size:
            raise ValueError("Darknet stream too short at %s" % name)
        head = stream[ptr:ptr + nb]
        ptr += nb
        w = stream[ptr:ptr + co * ci * k * k].reshape(co, ci, k, k)
        ptr += co * ci * k * k
        parts.append(np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0))).ravel())        # OIHW -> HWIO
        if bn:
            beta, gamma, mean, var = head[:co], head[co:2 * co], head[2 * co:3 * co], head[3 * co:]
            parts += [np.zeros(co, np.float32), gamma, beta, mean, var]                   # biases, scale, shift, mean, variance
        else:
            parts.append(head)
    if ptr != stream.size:
        raise ValueError("Darknet stream has %d floats, the architecture needs %d" % (stream.size, ptr))
    return np.concatenate(parts).astype(np.float32)


class yolo_v2(object):

    def __init__(self, isTraining=True, dtype="fp32", small_batch_plan=False):
        # yolo_v2.py:8-16.  The reference defaults to isTraining=True and its inference callers pass False
        # (test_val.py:167); the training graph (labels / loss_layer) is not part of this build.
        if isTraining:
            raise ValueError("yolo_amd implements the inference path only: construct with yolo_v2(False)")
        self.classes = cfg.CLASSES
        self.num_class = len(self.classes)
        self.box_per_cell = cfg.BOX_PRE_CELL
        self.cell_size = cfg.CELL_SIZE
        self.batch_size = cfg.BATCH_SIZE
        self.image_size = cfg.IMAGE_SIZE
        self.anchor = cfg.ANCHOR
        self.alpha = cfg.ALPHA
        self.dtype = dtype
        self._dtype_code = _lib.DTYPES[dtype]
        self._small_batch_plan = bool(small_batch_plan)    # bf16 only: split-K plan for batches of 1 - 8 (yolo_net_set_small_batch_plan)
        self._net = None
        self._ws = None
        self._vars = None
        self._convs = {}

    # ---- variables (replaces tf.train.Saver.restore, Detector.py:28-29) ---------------------------
    def load_variables(self, stream):
        stream = np.ascontiguousarray(stream, dtype=np.float32)
        _tensor.device()
        self._free()
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().yolo2_net_create(self.num_class, self.box_per_cell, self._dtype_code,
                                               stream.ctypes.data, stream.size, ctypes.byref(h)))
        self._net = h
        if self._small_batch_plan:
            _lib.check(_lib.lib().yolo_net_set_small_batch_plan(self._net, 1))
        self._vars = split_variables(stream, self.num_class, self.box_per_cell)
        return self

    def _free(self):
        if getattr(self, "_net", None) is not None:
            _lib.lib().yolo_net_destroy(self._net)
            self._net = None
        for h in getattr(self, "_convs", {}).values():
            _lib.lib().yolo_conv_destroy(h)
        self._convs = {}

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    # ---- yolo_v2.py:32-70 -------------------------------------------------------------------------
    def build_networks(self, inputs):
        """inputs [N, H, W, 3] float32 in [-1, 1] -> logits [N, H/32, W/32, box_per_cell*(num_class+5)]."""
        if self._net is None:
            raise RuntimeError("variables not loaded: call load_variables() first")
        x = _tensor.as_dev_f32(inputs)
        if x.dim() != 4 or x.shape[3] != 3:
            raise ValueError("inputs must be [N, H, W, 3], got %s" % (tuple(x.shape),))
        B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        L = _lib.lib()
        need = L.yolo_net_workspace_bytes(self._net, B, H, W)
        if need == 0:
            raise ValueError("unsupported input shape %s (H, W must be multiples of 32)" % (tuple(x.shape),))
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = _tensor.workspace(need)
        logits = _tensor.empty((B, H // 32, W // 32, self.box_per_cell * (self.num_class + 5)))
        _lib.check(L.yolo2_net_forward(self._net, x.data_ptr(), B, H, W, self._ws.data_ptr(), self._ws.numel(),
                                       logits.data_ptr(), _tensor.stream_ptr()))
        return logits

    # ---- yolo_v2.py:73-89 -------------------------------------------------------------------------
    def conv_layer(self, inputs, shape, batch_norm=True, name='0_conv'):
        """The reference creates the layer's variables here; this build looks them up by `name` in the
        loaded variable stream."""
        if self._vars is None or name not in self._vars:
            raise RuntimeError("conv_layer(%r): variables not loaded" % (name,))
        v = self._vars[name]
        if list(v["weight"].shape) != list(shape) or batch_norm != ("scale" in v):
            raise ValueError("conv_layer(%r): shape %s / batch_norm=%s do not fit the loaded variables %s"
                             % (name, list(shape), batch_norm, v["weight"].shape))
        x = _tensor.as_dev_f32(inputs)
        B, H, W, cin = [int(t) for t in x.shape]
        if cin != shape[2]:
            raise ValueError("conv_layer(%r): input has %d channels, kernel expects %d" % (name, cin, shape[2]))
        L = _lib.lib()
        if name not in self._convs:
            h = ctypes.c_void_p()
            arrs = [np.ascontiguousarray(v[k], np.float32) if k in v else None
                    for k in ("weight", "biases", "scale", "shift", "rolling_mean", "rolling_variance")]
            _lib.check(L.yolo2_conv_create(self._dtype_code, shape[2], shape[3], shape[0], int(batch_norm),
                                           *[None if a is None else a.ctypes.data for a in arrs], ctypes.byref(h)))
            self._convs[name] = h
        h = self._convs[name]
        out = _tensor.empty((B, H, W, shape[3]))
        ws = _tensor.workspace(L.yolo_conv_workspace_bytes(h, B, H, W))
        _lib.check(L.yolo_conv_forward(h, x.data_ptr(), B, H, W, None, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _tensor.stream_ptr()))
        return out

    # ---- yolo_v2.py:91-94 -------------------------------------------------------------------------
    def pooling_layer(self, inputs, name='1_pool'):
        x = _tensor.as_dev_f32(inputs)
        B, H, W, C = [int(t) for t in x.shape]
        if H % 2 or W % 2:
            raise ValueError("pooling_layer: odd sizes (SAME padding) are not implemented; the network never has them")
        out = _tensor.empty((B, H // 2, W // 2, C))
        _lib.check(_lib.lib().yolo_maxpool2x2(x.data_ptr(), B, H, W, C, out.data_ptr(), _tensor.stream_ptr()))
        return out

    # ---- yolo_v2.py:97-113 ------------------------------------------------------------------------
    def reorg(self, inputs):
        x = _tensor.as_dev_f32(inputs)
        B, H, W, C = [int(t) for t in x.shape]
        if H % 2 or W % 2:
            raise ValueError("reorg: needs even H and W")
        out = _tensor.empty((B, H // 2, W // 2, 4 * C))
        _lib.check(_lib.lib().yolo_reorg2x(x.data_ptr(), B, H, W, C, out.data_ptr(), _tensor.stream_ptr()))
        return out
