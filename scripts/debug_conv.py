import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import yolo_oracle as O
import test_gpu_parity as T
for dt in ("fp32", "bf16"):
    for case in T.CONV_CASES:
        got, ref = T._conv_case(torch, O, dt, *case, seed=100)
        d = np.abs(got - ref)
        bad = d > 1e-2 * (np.abs(ref) + 1)
        print(dt, case, "maxerr %.4g rms %.4g bad %.4f" % (d.max(), np.sqrt((ref**2).mean()), bad.mean()))
        if bad.any() and dt == "fp32":
            idx = np.argwhere(bad)
            print("  first bad", idx[:5].tolist(), "bad per channel(first 40):", bad.reshape(-1, bad.shape[-1]).mean(0)[:40].round(2).tolist())
            print("  bad per pixel row (first 40):", bad.reshape(-1, bad.shape[-1]).mean(1)[:40].round(2).tolist())
            i = tuple(idx[0]); print("  got", got[i], "ref", ref[i])
