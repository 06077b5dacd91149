"""Diagnostic: per-phase cycle breakdown of the conv kernel (library built with `make STAMPS=1`).
python scripts/stamps.py --only 12 [--cfg 0]"""
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
sys.path.insert(0, os.path.join(ROOT, "scripts"))
names = ["res+stores", "wait_vmcnt", "barrier", "stage_issue", "frag+mfma", "epi: sync1", "epi: acc->lds", "epi: sync2"]
import importlib
SH = None
src = open(os.path.join(ROOT, "scripts", "bench_layer.py")).read()
ns = {}
exec(src[src.index("SHAPES = ["):src.index("def main")], ns)
rng = np.random.default_rng(0)
for idx in [int(x) for x in a.only.split(",")]:
    cin, cout, k, s, res_in, resid, count = ns["SHAPES"][idx]
    w = rng.normal(0, 0.05, (k, k, cin, cout)).astype(np.float32)
    b = np.zeros(cout, np.float32); g = np.ones(cout, np.float32)
    h = ctypes.c_void_p()
    _lib.check(L.yolo_conv_create(1, cin, cout, k, s, 1, w.ctypes.data, b.ctypes.data, g.ctypes.data, b.ctypes.data, g.ctypes.data, ctypes.byref(h)))
    ms = ctypes.c_float()
    _lib.check(L.yolo_conv_benchmark(h, a.batch, res_in, res_in, resid, 3, ctypes.byref(ms)))
    buf = np.zeros(512, np.uint64)
    L.yolo_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.yolo_debug_read_stamps(buf.ctypes.data, 512) == 0
    st = buf.reshape(64, 8).astype(np.float64)
    st = st[st[:, 4] > 0]
    nsteps = (k * k * cin * 2 + 127) // 128
    print("layer %d cfg %s: %.4f ms, %d sampled blocks, %d K-steps" % (idx, a.cfg or "default", ms.value, len(st), nsteps))
    if a.cfg in ("22", "23"):
        nm = ["prologue", "barrier#1", "load issue", "frag+waits", "barrier#2", "mfma", "epilogue", "total"]
        for grp, rows in (("A", st[0::2]), ("B", st[1::2])):
            med = np.median(rows, axis=0)
            print("  group", grp)
            for n_, v in zip(nm, med):
                print("   %-12s %10.0f cycles  (%5.1f%%)   %.0f / step" % (n_, v, 100 * v / med[7], v / nsteps))
        continue
    raw0 = buf.reshape(64, 8)[:, 0]
    raw0 = raw0[buf.reshape(64, 8)[:, 4] > 0]
    ticks, cyc = (raw0 >> np.uint64(32)).astype(np.float64), (raw0 & np.uint64(0xffffffff)).astype(np.float64)
    if k == 3 and s == 1 and np.median(ticks) > 0:
        print("   whole tile: %.0f cycles in %.2f us -> in-kernel clock %.0f MHz (median over sampled workgroups)" % (
            np.median(cyc), np.median(ticks) / 100.0, np.median(cyc / ticks) * 100.0))
        st[:, 0] = 0
    med = np.median(st, axis=0)
    tot = med.sum()
    for n_, v in zip(names, med):
        per = v / nsteps if n_ in ("wait_vmcnt", "barrier", "stage_issue", "frag+mfma") else v
        print("   %-14s %10.0f cycles  (%5.1f%%)%s" % (n_, v, 100 * v / tot, "   %.0f / step" % per if per != v else ""))
    print("   %-14s %10.0f cycles" % ("sum (no prologue)", tot))
