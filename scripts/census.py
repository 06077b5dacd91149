"""Diagnostic (stamps build): which workgroups of the 3x3 halo kernel were resident on one CU at the same time.
python scripts/census.py --only 12 [--batch 64]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--only", default="12"); ap.add_argument("--cfg", default=""); ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
if a.cfg: os.environ["YOLO_CONV_CFG"] = a.cfg       # read once, at the library's first conv launch
from yolo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "yolo_amd", "lib", "libyolo_hip_stamps.so")
L = _lib.lib()
ns = {}
src = open(os.path.join(ROOT, "scripts", "bench_layer.py")).read()
exec(src[src.index("SHAPES = ["):src.index("def main")], ns)
rng = np.random.default_rng(0)
for idx in [int(x) for x in a.only.split(",")]:
    cin, cout, k, s, res_in, resid, count = ns["SHAPES"][idx]
    w = rng.normal(0, 0.05, (k, k, cin, cout)).astype(np.float32)
    b = np.zeros(cout, np.float32); g = np.ones(cout, np.float32)
    h = ctypes.c_void_p()
    _lib.check(L.yolo_conv_create(1, cin, cout, k, s, 1, w.ctypes.data, b.ctypes.data, g.ctypes.data, b.ctypes.data, g.ctypes.data, ctypes.byref(h)))
    ms = ctypes.c_float()
    os.environ["YOLO_BENCH_RING_MB"] = "1"
    _lib.check(L.yolo_conv_benchmark(h, a.batch, res_in, res_in, resid, 1, ctypes.byref(ms)))
    buf = np.zeros(16384, np.uint32)
    L.yolo_debug_read_census.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.yolo_debug_read_census(buf.ctypes.data, 16384) == 0
    c = buf.reshape(4096, 4)
    n = int((c[:, 3] != 0).sum())
    c = c[:n].astype(np.int64)
    hw, xcc, t0, t1 = c[:, 0], c[:, 1] & 15, c[:, 2], c[:, 3]
    key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)
    tmin = t0.min()
    print("layer %d: %d workgroups on %d distinct CUs, kernel span %.1f us, tile time median %.1f us" % (
        idx, n, len(set(key.tolist())), (t1.max() - tmin) / 100.0, float(np.median(t1 - t0)) / 100.0))
    maxres = []
    for kk in sorted(set(key.tolist())):
        sel = np.where(key == kk)[0]
        ev = sorted([(int(t0[i]), 1) for i in sel] + [(int(t1[i]), -1) for i in sel])
        cur = best = 0
        for _, d in ev:
            cur += d; best = max(best, cur)
        maxres.append(best)
    print("   max co-resident workgroups per CU: histogram", np.bincount(maxres))
    for kk in sorted(set(key.tolist()))[:3]:
        sel = np.where(key == kk)[0]
        print("   cu %x:" % kk, ", ".join("wg %d %.1f..%.1f" % (i, (t0[i] - tmin) / 100.0, (t1[i] - tmin) / 100.0) for i in sel))
