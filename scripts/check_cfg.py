"""Single-layer A/B of tile configurations: python scripts/check_cfg.py --ref 88 --cfgs 90,91,...  (bf16 1x1 layers, several shapes;
prints max |diff| against the reference configuration's output and against an f32 torch conv of the bf16-rounded operands)."""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib, _tensor
ap = argparse.ArgumentParser(); ap.add_argument("--ref", default="88"); ap.add_argument("--cfgs", default="90,91,92,93,94,95")
a = ap.parse_args()
L = _lib.lib()
rng = np.random.default_rng(0)
SHAPES = [(512, 256, 3, 26, 26, False), (256, 128, 2, 52, 52, False), (1024, 512, 3, 13, 13, True), (128, 64, 1, 104, 104, False),
          (768, 256, 1, 26, 26, False), (384, 128, 1, 52, 20, True), (64, 128, 5, 7, 9, True)]
bad = 0
for cin, cout, B, H, W, resid in SHAPES:
    w = rng.normal(0, (2.0 / cin) ** 0.5, (1, 1, cin, cout)).astype(np.float32)
    beta = rng.normal(0, 0.1, cout).astype(np.float32); gamma = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    mean = rng.normal(0, 0.1, cout).astype(np.float32); var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    h = ctypes.c_void_p()
    _lib.check(L.yolo_conv_create(1, cin, cout, 1, 1, 1, w.ctypes.data, beta.ctypes.data, gamma.ctypes.data, mean.ctypes.data, var.ctypes.data, ctypes.byref(h)))
    x = torch.from_numpy(rng.normal(0, 1, (B, H, W, cin)).astype(np.float32)).cuda()
    r = torch.from_numpy(rng.normal(0, 1, (B, H, W, cout)).astype(np.float32)).cuda() if resid else None
    ws = _tensor.workspace(L.yolo_conv_workspace_bytes(h, B, H, W))
    outs = {}
    for cfg in [a.ref] + a.cfgs.split(","):
        _lib.set_conv_cfg(cfg)
        out = torch.full((B, H, W, cout), float("nan"), device="cuda")
        _lib.check(L.yolo_conv_forward(h, x.data_ptr(), B, H, W, _tensor.ptr(r), out.data_ptr(), ws.data_ptr(), ws.numel(), _tensor.stream_ptr()))
        torch.cuda.synchronize()
        outs[cfg] = out.cpu().numpy()
    ref = outs[a.ref]
    line = "%4d->%4d B%d %dx%d res%d:" % (cin, cout, B, H, W, resid)
    for cfg in a.cfgs.split(","):
        d = np.abs(outs[cfg] - ref)
        nan = int(np.isnan(outs[cfg]).sum())
        m = float(np.nanmax(d)) if d.size else 0.0
        ok = nan == 0 and m <= 0.07 * max(1.0, float(np.abs(ref).max())) * 0.5
        bad += not ok
        line += "  %s: max %.4f nan %d%s" % (cfg, m, nan, "" if ok else " <-- BAD")
    print(line, " (ref absmax %.2f)" % float(np.abs(ref).max()), flush=True)
    L.yolo_conv_destroy(h)
sys.exit(1 if bad else 0)
