# coding: utf-8
"""conv2d / darknet53_body / yolo_block / upsample_layer with the signatures of reference
yolo_v3/utils/layer_utils.py, one HIP launch per call (yolo_conv_forward / yolo_upsample2x).

TF creates a layer's variables when the layer function is called; here the variables already exist
(a Darknet weight stream), so the calls consume them in creation order from the active
`variable_scope(params)` - the eager counterpart of `tf.variable_scope` + `Saver.restore`.
`yolov3.forward` does not go through these functions: it runs the whole fused plan
(yolo_net_forward).  These exist for layer-level drop-in use and tests.
"""
from __future__ import division, print_function

import contextlib
import ctypes

import numpy as np

from .. import _lib, _tensor

_SCOPE = None


class _Scope(object):
    def __init__(self, params, dtype):
        self.params, self.dtype, self.next = list(params), dtype, 0
        self.handles = {}

    def take(self, cin, filters, kernel_size, strides):
        if self.next >= len(self.params):
            raise RuntimeError("variable_scope exhausted: more conv2d calls than weight sets")
        i, p = self.next, self.params[self.next]
        self.next += 1
        w = p["w"]
        if w.shape != (kernel_size, kernel_size, cin, filters):
            raise ValueError("conv #%d: weights %s do not fit conv2d(%d->%d, k=%d)" % (i, w.shape, cin, filters, kernel_size))
        if i not in self.handles:
            L = _lib.lib()
            h = ctypes.c_void_p()
            bn = "gamma" in p
            ps = [p["beta"], p["gamma"], p["mean"], p["var"]] if bn else [p["bias"], None, None, None]
            ps = [None if a is None else np.ascontiguousarray(a, np.float32) for a in ps]
            w = np.ascontiguousarray(w, np.float32)
            _lib.check(L.yolo_conv_create(_lib.DTYPES[self.dtype], cin, filters, kernel_size, strides, int(bn), w.ctypes.data,
                                          *[None if a is None else a.ctypes.data for a in ps], ctypes.byref(h)))
            self.handles[i] = h
        return self.handles[i]

    def close(self):
        for h in self.handles.values():
            _lib.lib().yolo_conv_destroy(h)
        self.handles = {}


@contextlib.contextmanager
def variable_scope(params, dtype="fp32"):
    """params: list of per-conv dicts in creation order (yolo_amd.weights.split_stream)."""
    global _SCOPE
    prev, _SCOPE = _SCOPE, _Scope(params, dtype)
    try:
        yield _SCOPE
    finally:
        _SCOPE.close()
        _SCOPE = prev


def conv2d(inputs, filters, kernel_size, strides=1, _residual=None):
    """layer_utils.py:9-21 (+ the arg_scope of model.py:32-36).  `_residual` fuses `net + shortcut`."""
    if _SCOPE is None:
        raise RuntimeError("conv2d must be called inside `with variable_scope(params):`")
    x = _tensor.as_dev_f32(inputs)
    B, H, W, cin = [int(v) for v in x.shape]
    h = _SCOPE.take(cin, filters, kernel_size, strides)
    L = _lib.lib()
    ho, wo = (H // strides, W // strides)
    out = _tensor.empty((B, ho, wo, filters))
    ws = _tensor.workspace(L.yolo_conv_workspace_bytes(h, B, H, W))
    res = _tensor.as_dev_f32(_residual) if _residual is not None else None
    _lib.check(L.yolo_conv_forward(h, x.data_ptr(), B, H, W, _tensor.ptr(res), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                   _tensor.stream_ptr()))
    return out


def darknet53_body(inputs):
    """layer_utils.py:23-54 -> route_1, route_2, route_3."""
    def res_block(inputs, filters):
        shortcut = inputs
        net = conv2d(inputs, filters * 1, 1)
        return conv2d(net, filters * 2, 3, _residual=shortcut)        # net + shortcut fused (:28)

    net = conv2d(inputs, 32, 3, strides=1)
    net = conv2d(net, 64, 3, strides=2)
    net = res_block(net, 32)
    net = conv2d(net, 128, 3, strides=2)
    for i in range(2):
        net = res_block(net, 64)
    net = conv2d(net, 256, 3, strides=2)
    for i in range(8):
        net = res_block(net, 128)
    route_1 = net
    net = conv2d(net, 512, 3, strides=2)
    for i in range(8):
        net = res_block(net, 256)
    route_2 = net
    net = conv2d(net, 1024, 3, strides=2)
    for i in range(4):
        net = res_block(net, 512)
    route_3 = net
    return route_1, route_2, route_3


def yolo_block(inputs, filters):
    """layer_utils.py:56-65 -> (route, net)."""
    net = conv2d(inputs, filters * 1, 1)
    net = conv2d(net, filters * 2, 3)
    net = conv2d(net, filters * 1, 1)
    net = conv2d(net, filters * 2, 3)
    net = conv2d(net, filters * 1, 1)
    route = net
    net = conv2d(net, filters * 2, 3)
    return route, net


def upsample_layer(inputs, out_shape):
    """layer_utils.py:67-72: tf.image.resize_nearest_neighbor(inputs, (out_shape[1], out_shape[2]), align_corners=True).
    The exact-2x case of the network (where the TF rule equals src = dst // 2) runs the 16-byte-per-lane
    kernel; any other size the general one (TF 1.x rounding: src = min(round(dst * (in - 1) / (out - 1)), in - 1))."""
    x = _tensor.as_dev_f32(inputs)
    B, H, W, C = [int(v) for v in x.shape]
    new_height, new_width = int(out_shape[1]), int(out_shape[2])
    if new_height <= 0 or new_width <= 0:
        raise ValueError("upsample_layer: bad output size %dx%d" % (new_height, new_width))
    out = _tensor.empty((B, new_height, new_width, C))
    if new_height == 2 * H and new_width == 2 * W and C % 4 == 0:
        _lib.check(_lib.lib().yolo_upsample2x(x.data_ptr(), B, H, W, C, out.data_ptr(), _tensor.stream_ptr()))
    else:
        _lib.check(_lib.lib().yolo_upsample_nearest(x.data_ptr(), B, H, W, C, new_height, new_width, out.data_ptr(),
                                                    _tensor.stream_ptr()))
    return out
