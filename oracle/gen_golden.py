"""Generate tests/golden/*.npz by RUNNING reference code in the build container.

Run from the repo root:  python -B oracle/gen_golden.py
Needs /root/reference (read-only).  The reference never travels: only the arrays written here do.

G1  conv stack      : reference yolo-v3/002 `Darknet` (cfg/yolov3.cfg) with a seeded synthetic Darknet
                      weight FILE loaded by the reference's own `Darknet.load_weights`
                      (002/darknet.py:212-265) -> raw outputs of the three pre-yolo convs
                      (module_list[81], [93], [105]) at 64x64 and 96x96 inputs.
G2  decode          : reference `predict_transform` (002/util.py) output [1, n, 85] of the same runs.
G3  cpu_nms/py_nms  : reference yolo_v3/utils/nms_utils.py numpy functions on seeded boxes.
                      (`tensorflow` is stubbed with an empty module: only `gpu_nms` touches it.)
G3b calc_iou        : reference yolo_v3/utils/eval_utils.py:7-21.
G4  v2 decode + NMS : reference yolo_v2/Detector.py `calc_output` (:51-89, with get_boxes / calc_iou) called on a
                      bare `Detector.__new__` object whose attributes come from the reference's yolo/config.py
                      (`tensorflow` and `cv2` stubbed with empty modules: only __init__/detect/draw touch them).
"""
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import yolo_oracle as O   # noqa: E402


def gen_conv_and_decode():
    import torch
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "yolo-v3", "002"))
    sys.path.insert(0, os.getcwd())
    try:
        import darknet                                     # reference module
        specs = O.yolov3_conv_specs(80)
        stream = O.synth_darknet_stream(specs, seed=1234)
        assert len(stream) == 62001757, len(stream)        # size of the public yolov3.weights body
        with tempfile.NamedTemporaryFile(suffix=".weights", dir="/tmp", delete=False) as tf_:
            path = tf_.name
        O.write_darknet_weights(path, stream)
        model = darknet.Darknet("cfg/yolov3.cfg")
        model.load_weights(path)
        os.unlink(path)
        model.eval()
        torch.set_grad_enabled(False)
        for size, seed in ((64, 11), (96, 12)):
            grabbed = {}
            hooks = [model.module_list[i].register_forward_hook(
                lambda m, a, out, i=i: grabbed.__setitem__(i, out.detach().clone()))
                for i in (81, 93, 105)]
            extra = {}
            for name, i in (("route_1", 36), ("route_2", 61), ("route_3", 74)):
                pass   # shortcut outputs are EmptyLayer (no tensor flows through the module): taken from `outputs` below
            model.net_info["height"] = size
            rng = np.random.default_rng(seed)
            img = rng.random((1, size, size, 3), dtype=np.float32)
            x = torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()
            det = model(x, False)
            for h in hooks:
                h.remove()
            fms = [grabbed[i].permute(0, 2, 3, 1).contiguous().numpy() for i in (81, 93, 105)]
            np.savez_compressed(os.path.join(OUT, "g1_convstack_%d.npz" % size),
                                img=img, fm1=fms[0], fm2=fms[1], fm3=fms[2],
                                det=det.numpy(), weight_seed=np.int64(1234),
                                stream_sum=np.float64(stream.astype(np.float64).sum()),
                                stream_probe=stream[::1000003].copy())
            print("G1/G2", size, [f.shape for f in fms], det.shape, float(np.abs(fms[0]).mean()))
    finally:
        os.chdir(cwd)


def gen_nms():
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    sys.path.insert(0, os.path.join(REF, "yolo_v3"))
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    from utils import nms_utils, eval_utils              # reference modules

    cases = {}

    def boxes_rand(rng, n, span=400.0, size=120.0):
        xy = rng.random((n, 2)) * span
        wh = rng.random((n, 2)) * size + 1.0
        return np.concatenate([xy, xy + wh], 1).astype(np.float32)

    def clustered(rng, n, seeds=12, jitter=6.0):
        base = boxes_rand(rng, seeds)
        pick = rng.integers(0, seeds, n)
        return (base[pick] + rng.normal(0, jitter, (n, 4))).astype(np.float32)

    def run(name, boxes, scores, C, **kw):
        b, s, l = nms_utils.cpu_nms(boxes.copy(), scores.copy(), C, **kw)
        d = dict(boxes=boxes, scores=scores, C=np.int32(C),
                 max_boxes=np.int32(kw.get("max_boxes", 50)),
                 score_thresh=np.float32(kw.get("score_thresh", 0.5)),
                 iou_thresh=np.float32(kw.get("iou_thresh", 0.5)),
                 none=np.bool_(b is None))
        if b is not None:
            d.update(out_boxes=b, out_scores=s, out_labels=l)
        cases[name] = d
        print("G3", name, None if b is None else (b.shape, l.dtype, s.dtype))

    rng = np.random.default_rng(31)
    b = boxes_rand(rng, 200)
    s = rng.random((200, 3)).astype(np.float32)
    run("rand200x3", b[None], s[None], 3)
    b = clustered(rng, 600)
    s = rng.random((600, 5)).astype(np.float32)
    run("clustered600x5", b[None], s[None], 5, max_boxes=20, score_thresh=0.3, iou_thresh=0.45)
    run("clustered600x5_maxcut", b[None], s[None], 5, max_boxes=3, score_thresh=0.1, iou_thresh=0.6)
    run("all_filtered", b[None], (s * 0.2)[None], 5, score_thresh=0.9)          # -> (None,)*3
    s2 = s.copy()
    s2[:, 1] = 0.0                                                               # empty class in the middle
    run("empty_class", b[None], s2[None], 5, score_thresh=0.4)
    bz = b.copy()
    bz[::7, 2:] = bz[::7, :2]                                                    # zero-area boxes
    run("zero_area", bz[None], s[None], 5, score_thresh=0.2)
    b3 = clustered(rng, 3000, seeds=40)
    s3 = (rng.random((3000, 4)) ** 0.5).astype(np.float32)
    run("big3000x4", b3[None], s3[None], 4, max_boxes=50, score_thresh=0.05, iou_thresh=0.5)
    # batch > 1: the reference flattens the batch dim (nms_utils.py:69-70)
    run("batch2_flatten", b[:400].reshape(2, 200, 4), s[:400].reshape(2, 200, 5), 5, score_thresh=0.5)
    # documented tie case: n <= 16 equal scores (numpy insertion sort is stable there)
    bt = boxes_rand(rng, 10, span=1000.0, size=20.0)
    st = np.full((10, 1), 0.75, np.float32)
    run("ties10", bt[None], st[None], 1)

    # py_nms directly (indices)
    bp = clustered(rng, 500)
    sp = rng.random(500).astype(np.float32)
    keep = nms_utils.py_nms(bp.copy(), sp.copy(), max_boxes=40, iou_thresh=0.5)
    cases["py_nms_500"] = dict(boxes=bp, scores=sp, keep=np.asarray(keep, np.int64),
                               max_boxes=np.int32(40), iou_thresh=np.float32(0.5))
    flat = {}
    for cname, d in cases.items():
        for k, v in d.items():
            flat["%s/%s" % (cname, k)] = v
    np.savez_compressed(os.path.join(OUT, "g3_cpu_nms.npz"), **flat)

    pb = boxes_rand(rng, 30)
    tb = boxes_rand(rng, 12)
    iou = eval_utils.calc_iou(pb.copy(), tb.copy())
    np.savez_compressed(os.path.join(OUT, "g3b_calc_iou.npz"), pred=pb, true=tb, iou=iou)
    print("G3b", iou.shape, iou.dtype)


def gen_eval():
    """G5: the reference's recall / precision bookkeeping (eval_utils.py:23-88) on seeded ground truth + predictions."""
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    sys.path.insert(0, os.path.join(REF, "yolo_v3"))
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    from utils import eval_utils                          # reference module
    rng = np.random.default_rng(51)
    flat = {}
    for name, N, C, n_noise, jitter, st, it in (("a", 3, 4, 150, 3.0, 0.5, 0.5), ("b", 2, 6, 400, 8.0, 0.3, 0.45),
                                                 ("dup", 2, 3, 60, 1.0, 0.2, 0.5)):
        grids = (4, 8, 16)
        y_true = [np.zeros((N, g, g, 3, 5 + C), np.float32) for g in grids]
        pb, pc, pp = [], [], []
        for i in range(N):
            boxes_i, conf_i, prob_i = [], [], []
            for j, g in enumerate(grids):
                k = int(rng.integers(2, 6))
                for _ in range(k):
                    y, x, a = int(rng.integers(0, g)), int(rng.integers(0, g)), int(rng.integers(0, 3))
                    cx, cy = rng.random(2) * 120 + 4
                    w, h = rng.random(2) * 40 + 6
                    c = int(rng.integers(0, C))
                    y_true[j][i, y, x, a, 0:4] = (cx, cy, w, h)
                    y_true[j][i, y, x, a, 4] = 1.0
                    y_true[j][i, y, x, a, 5 + c] = 1.0
                    reps = 3 if name == "dup" else int(rng.integers(0, 3))       # several predictions on one object
                    for _ in range(reps):
                        d = rng.normal(0, jitter, 4)
                        boxes_i.append([cx - w / 2 + d[0], cy - h / 2 + d[1], cx + w / 2 + d[2], cy + h / 2 + d[3]])
                        conf_i.append(0.6 + 0.4 * rng.random())
                        p = rng.random(C) * 0.2
                        p[c if rng.random() < 0.8 else int(rng.integers(0, C))] = 0.7 + 0.3 * rng.random()
                        prob_i.append(p)
            for _ in range(n_noise):
                xy = rng.random(2) * 120
                wh = rng.random(2) * 30 + 2
                boxes_i.append([xy[0], xy[1], xy[0] + wh[0], xy[1] + wh[1]])
                conf_i.append(rng.random() * 0.7)
                prob_i.append(rng.random(C))
            pb.append(boxes_i); pc.append(conf_i); pp.append(prob_i)
        n = max(len(b) for b in pb)
        for i in range(N):                                  # pad to a common n with zero-score boxes
            while len(pb[i]) < n:
                pb[i].append([0, 0, 1, 1]); pc[i].append(0.0); pp[i].append(np.zeros(C))
        y_pred = [np.asarray(pb, np.float32), np.asarray(pc, np.float32)[..., None], np.asarray(pp, np.float32)]
        tp, tl, pl = eval_utils.evaluate_on_cpu([a.copy() for a in y_pred], [a.copy() for a in y_true], C, calc_now=False,
                                                score_thresh=st, iou_thresh=it)
        rec, prec = eval_utils.evaluate_on_cpu([a.copy() for a in y_pred], [a.copy() for a in y_true], C, calc_now=True,
                                               score_thresh=st, iou_thresh=it)
        for j in range(3):
            flat["%s/y_true%d" % (name, j)] = y_true[j]
        flat[name + "/pred_boxes"], flat[name + "/pred_confs"], flat[name + "/pred_probs"] = y_pred
        flat[name + "/C"] = np.int32(C)
        flat[name + "/score_thresh"], flat[name + "/iou_thresh"] = np.float32(st), np.float32(it)
        flat[name + "/true_positive"] = np.asarray([tp[c] for c in range(C)], np.int64)
        flat[name + "/true_labels"] = np.asarray([tl[c] for c in range(C)], np.int64)
        flat[name + "/pred_labels"] = np.asarray([pl[c] for c in range(C)], np.int64)
        flat[name + "/recall_precision"] = np.asarray([rec, prec], np.float64)
        print("G5", name, dict(tp), dict(tl), dict(pl), rec, prec)
    np.savez_compressed(os.path.join(OUT, "g5_eval.npz"), **flat)


def gen_v2_decode():
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    for m in [k for k in sys.modules if k == "yolo" or k.startswith("yolo.") or k == "Detector"]:
        del sys.modules[m]
    sys.path.insert(0, os.path.join(REF, "yolo_v2"))
    import Detector as D                                   # reference module
    import yolo.config as cfg                              # reference constants
    det = D.Detector.__new__(D.Detector)                   # no tf.Session: calc_output is pure numpy
    det.classes = cfg.CLASSES
    det.num_classes = len(cfg.CLASSES)
    det.image_size = cfg.IMAGE_SIZE
    det.cell_size = cfg.CELL_SIZE
    det.box_per_cell = cfg.BOX_PRE_CELL
    det.threshold = cfg.THRESHOLD
    det.anchor = cfg.ANCHOR
    rng = np.random.default_rng(41)
    flat = {}

    def logits(conf_bias, cls_scale, wh_scale=0.4):
        t = rng.normal(0, 1, (1, 13, 13, 5, 25)).astype(np.float32)
        t[..., 2:4] *= wh_scale
        t[..., 4] += conf_bias
        t[..., 5:] *= cls_scale
        return t.reshape(1, 13, 13, 125)

    for name, t in (("a", logits(-2.0, 3.0)), ("b", logits(1.5, 4.0)), ("c", logits(0.0, 2.5, 0.8)),
                    ("empty", logits(-8.0, 1.0))):
        res = det.calc_output(t.copy())
        flat[name + "/logits"] = t
        flat[name + "/cls"] = np.asarray([cfg.CLASSES.index(r[0]) for r in res], np.int32)
        flat[name + "/boxes"] = np.asarray([r[1:5] for r in res], np.float64).reshape(-1, 4)
        flat[name + "/probs"] = np.asarray([r[5] for r in res], np.float32)
        print("G4", name, len(res))
    np.savez_compressed(os.path.join(OUT, "g4_v2_decode.npz"), **flat)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["conv", "nms", "v2"]
    if "v2" in which:
        gen_v2_decode()
    if "nms" in which:
        gen_nms()
    if "eval" in which:
        gen_eval()
    if "conv" in which:
        gen_conv_and_decode()
