"""ctypes binding of libyolo_hip.so (include/yolo_hip.h).  No fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO_HIP_LIB=<path>: measurement scripts load a diagnostic build (make STAMPS=1 / EXP=<n>) instead of the shipped library
LIB_PATH = os.environ.get("YOLO_HIP_LIB") or os.path.join(_HERE, "lib", "libyolo_hip.so")

YOLO_DTYPE_F32, YOLO_DTYPE_BF16 = 0, 1
YOLO_NMS_TF, YOLO_NMS_PY = 0, 1
DTYPES = {"fp32": YOLO_DTYPE_F32, "float32": YOLO_DTYPE_F32, "f32": YOLO_DTYPE_F32,
          "bf16": YOLO_DTYPE_BF16, "bfloat16": YOLO_DTYPE_BF16}

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); every symbol declared in include/yolo_hip.h
SIGNATURES = {
    "yolo_last_error": (_c.c_char_p, []),
    "yolo_abi_version": (_i, []),
    "yolo_net_create": (_i, [_i, _i, _vp, _sz, _c.POINTER(_vp)]),
    "yolo_net_destroy": (_i, [_vp]),
    "yolo_net_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "yolo_net_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "yolo_net_set_small_batch_plan": (_i, [_vp, _i]),
    "yolo_net_num_layers": (_i, [_vp]),
    "yolo_net_layer_info": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "yolo_net_layer_kernel": (_c.c_char_p, [_vp, _i]),
    "yolo_net_forward_timed": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i]),
    "yolo_conv_create": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _c.POINTER(_vp)]),
    "yolo_conv_destroy": (_i, [_vp]),
    "yolo_conv_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "yolo_conv_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "yolo_conv_benchmark": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "yolo_debug_set_conv_cfg": (_i, [_i]),
    "yolo_upsample2x": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "yolo_upsample_nearest": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "yolo_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "yolo_reorg_boxes": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "yolo2_net_create": (_i, [_i, _i, _i, _vp, _sz, _c.POINTER(_vp)]),
    "yolo2_net_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "yolo2_conv_create": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(_vp)]),
    "yolo_maxpool2x2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "yolo_reorg2x": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "yolo2_decode": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "yolo2_nms_workspace_bytes": (_sz, [_i, _i, _i]),
    "yolo2_nms": (_i, [_vp, _vp, _i, _i, _i, _f, _c.c_double, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "yolo_net_detect_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "yolo_net_detect": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "yolo_preprocess": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "yolo_detect_post_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "yolo_detect_post": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "yolo_nms_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "yolo_nms": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "yolo_pack_detections": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "yolo_nms_static": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: build it with `make -C yolo_amd/csrc` "
                               "(or `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        # torch-ROCm must load ITS HIP runtime first: libyolo_hip.so then binds to the same
        # libamdhip64 instance, so torch allocations, streams and our launches share one context.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def set_conv_cfg(cfg):
    """Test / measurement aid: force a conv tile configuration (None or < 0: the library's own choice)."""
    check(lib().yolo_debug_set_conv_cfg(-1 if cfg is None else int(cfg)))


class YoloHipError(RuntimeError):
    pass


def check(rc):
    """Map a C return code to the Python exception the reference's callers would see."""
    if rc == 0:
        return
    msg = lib().yolo_last_error().decode("utf-8", "replace")
    if rc in (-1, -2, -4):
        raise ValueError("libyolo_hip: %s (code %d)" % (msg, rc))
    raise YoloHipError("libyolo_hip: %s (code %d)" % (msg, rc))
