"""Per-layer microbenchmark of the conv kernels: python scripts/bench_layer.py [--dtype bf16] [--batch 64] [--only IDX,...]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from yolo_amd import _lib

# distinct shapes of YOLOv3-416: cin, cout, k, stride, input res, residual, count
SHAPES = [
    (3, 32, 3, 1, 416, 0, 1), (32, 64, 3, 2, 416, 0, 1), (64, 32, 1, 1, 208, 0, 1), (32, 64, 3, 1, 208, 1, 1),
    (64, 128, 3, 2, 208, 0, 1), (128, 64, 1, 1, 104, 0, 2), (64, 128, 3, 1, 104, 1, 2),
    (128, 256, 3, 2, 104, 0, 1), (256, 128, 1, 1, 52, 0, 10), (128, 256, 3, 1, 52, 1, 11),
    (256, 512, 3, 2, 52, 0, 1), (512, 256, 1, 1, 26, 0, 10), (256, 512, 3, 1, 26, 1, 11),
    (512, 1024, 3, 2, 26, 0, 1), (1024, 512, 1, 1, 13, 0, 7), (512, 1024, 3, 1, 13, 1, 7),
    (1024, 255, 1, 1, 13, 0, 1), (512, 256, 1, 1, 13, 0, 1), (768, 256, 1, 1, 26, 0, 1), (512, 255, 1, 1, 26, 0, 1),
    (256, 128, 1, 1, 26, 0, 1), (384, 128, 1, 1, 52, 0, 1), (256, 255, 1, 1, 52, 0, 1),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    L = _lib.lib()
    only = [int(x) for x in a.only.split(",")] if a.only else range(len(SHAPES))
    rng = np.random.default_rng(0)
    tot = 0.0
    for idx in only:
        cin, cout, k, s, res_in, resid, count = SHAPES[idx]
        w = rng.normal(0, 0.05, (k, k, cin, cout)).astype(np.float32)
        b = np.zeros(cout, np.float32); g = np.ones(cout, np.float32)
        bn = int(cout != 255)
        h = ctypes.c_void_p()
        _lib.check(L.yolo_conv_create(_lib.DTYPES[a.dtype], cin, cout, k, s, bn, w.ctypes.data, b.ctypes.data, g.ctypes.data, b.ctypes.data, g.ctypes.data, ctypes.byref(h)))
        ms = ctypes.c_float()
        _lib.check(L.yolo_conv_benchmark(h, a.batch, res_in, res_in, resid, a.iters, ctypes.byref(ms)))
        L.yolo_conv_destroy(h)
        ho = res_in // s
        gf = 2.0 * a.batch * ho * ho * cout * cin * k * k / 1e9
        tot += ms.value * count
        print("%2d  %4d->%4d k%d s%d @%3d res%d x%2d : %8.4f ms  %8.1f TFLOP/s" % (idx, cin, cout, k, s, res_in, resid, count, ms.value, gf / ms.value), flush=True)
    print("weighted total %.3f ms" % tot)

main()
