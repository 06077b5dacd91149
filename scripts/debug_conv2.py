import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from yolo_amd import _lib, _tensor
L = _lib.lib()
def run(dt, cin, cout, x, w):
    h = ctypes.c_void_p()
    b = np.zeros(cout, np.float32)
    _lib.check(L.yolo_conv_create(_lib.DTYPES[dt], cin, cout, 1, 1, 0, w.ctypes.data, b.ctypes.data, None, None, None, ctypes.byref(h)))
    B, H, W = x.shape[:3]
    xd = _tensor.as_dev_f32(x); out = _tensor.empty((B, H, W, cout))
    ws = _tensor.workspace(L.yolo_conv_workspace_bytes(h, B, H, W))
    _lib.check(L.yolo_conv_forward(h, xd.data_ptr(), B, H, W, None, out.data_ptr(), ws.data_ptr(), ws.numel(), _tensor.stream_ptr()))
    return out.cpu().numpy()
cin, cout = 64, 64
x = np.zeros((1, 4, 4, cin), np.float32); x[...] = np.arange(cin, dtype=np.float32)
w = np.zeros((1, 1, cin, cout), np.float32)
for n in range(cout): w[0, 0, n, n] = 1
for dt in ("bf16", "fp32"):
    o = run(dt, cin, cout, x, w)
    print(dt, "pixel0:", o[0, 0, 0].tolist())
    print(dt, "pixel5:", o[0, 1, 1].tolist())
# w = ones -> out = sum_c x
w1 = np.ones((1, 1, cin, cout), np.float32)
print("sum", run("fp32", cin, cout, x, w1)[0, 0, 0, :8], "expect", x[0, 0, 0].sum())
