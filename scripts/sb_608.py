"""YOLOv3-608 batch 8 (config 5 end to end): default plan vs small-batch plan, lanes 1 / default (python scripts/sb_608.py [B])."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from yolo_amd.model import yolov3
    from yolo_amd.utils.misc_utils import DEFAULT_ANCHORS
    from yolo_amd.weights import synth_darknet_stream
    B, S, plan = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    net = yolov3(80, DEFAULT_ANCHORS, dtype="bf16", small_batch_plan=bool(plan)).load_weight_stream(synth_darknet_stream(80, seed=1234))
    x = torch.rand(B, S, S, 3, device="cuda")
    for _ in range(10):
        net.detect(x, 50, 0.5, 0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        net.detect(x, 50, 0.5, 0.5)
    e1.record(); torch.cuda.synchronize()
    print("batch %d %dx%d lanes %s plan %d: %.3f ms" % (B, S, S, os.environ.get("YOLO_LANES", "auto"), plan, e0.elapsed_time(e1) / 100), flush=True)
    sys.exit(0)
for B, S in ((8, 608), (8, 416), (16, 416), (32, 416)):
    for lanes in ("", "1", "2"):
        for plan in (0, 1):
            env = dict(os.environ)
            if lanes: env["YOLO_LANES"] = lanes
            subprocess.run([sys.executable, __file__, "--child", str(B), str(S), str(plan)], env=env, check=True, timeout=120)
