"""G6: the block list the reference's own cfg parser (yolo-v3/002/darknet.py:12-36 parse_cfg) yields for its
cfg/yolov3.cfg, written as tests/golden/g6_yolov3_cfg_blocks.json.  Run in the build container (needs /root/reference,
read-only; the reference never travels, only this output does):  python -B oracle/gen_golden_cfg.py"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/yolo-v3/002"

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
os.chdir(REF)
sys.path.insert(0, REF)
import darknet   # noqa: E402  (reference module)

blocks = darknet.parse_cfg("cfg/yolov3.cfg")
out = os.path.join(ROOT, "tests", "golden", "g6_yolov3_cfg_blocks.json")
json.dump(blocks, open(out, "w"), indent=0)
print(len(blocks), "blocks ->", out)
