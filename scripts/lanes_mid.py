"""One lane vs two at mid-size batches (where does the two-lane rule start to pay?): python scripts/lanes_mid.py"""
import sys, os, subprocess
for B in (20, 24, 28, 40, 48, 56):
    for lanes in ("1", "2", "3"):
        env = dict(os.environ); env["YOLO_LANES"] = lanes
        subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sb_608.py"), "--child", str(B), "416", "0"], env=env, check=True, timeout=120)
