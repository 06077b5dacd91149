"""Darknet `.cfg` ingestion for the YOLOv3 path: the text format the reference's PyTorch lesson parses
(yolo-v3/002/darknet.py:12-36 `parse_cfg`, cfg/yolov3.cfg) -> block list -> the conv specs / anchors / class count that
`yolov3` is built from.  The execution plan of libyolo_hip.so is the static YOLOv3 topology (yolo_v3/model.py:21-67 =
cfg/yolov3.cfg:607-786); `plan_from_blocks` therefore VALIDATES that a cfg describes that topology (any class count, any
anchors, any multiple-of-32 size) and raises otherwise - it does not build arbitrary networks."""
import numpy as np

from . import weights


def parse_cfg(cfgfile_or_text):
    """cfg file path, or the text itself (anything containing a newline) -> list of dicts, one per [section], 'type' = the
    section name; blank lines and lines starting with '#' are dropped, keys / values stripped (002/darknet.py:12-36)."""
    text = cfgfile_or_text
    if "\n" not in text:
        with open(text, "r") as f:
            text = f.read()
    lines = [x.strip() for x in text.split("\n")]
    lines = [x for x in lines if len(x) > 0 and x[0] != "#"]
    block, blocks = {}, []
    for line in lines:
        if line[0] == "[":
            if len(block) != 0:
                blocks.append(block)
                block = {}
            block["type"] = line[1:-1].rstrip()
        else:
            if "=" not in line:
                raise ValueError("cfg: expected key=value, got %r" % line)
            key, value = line.split("=", 1)
            block[key.rstrip()] = value.lstrip()
    if block:
        blocks.append(block)
    return blocks


def blocks_to_text(blocks):
    """Inverse of parse_cfg (used by tests and to write a cfg for a re-classed model)."""
    out = []
    for b in blocks:
        out.append("[%s]" % b["type"])
        out.extend("%s=%s" % (k, v) for k, v in b.items() if k != "type")
        out.append("")
    return "\n".join(out)


def plan_from_blocks(blocks):
    """-> dict(specs=[dict(cin, cout, k, stride, bn)] in file order, class_num, anchors float32 [9, 2], width, height).
    Walks the blocks like 002/darknet.py:55-148 `create_modules` (filter bookkeeping through route / shortcut / upsample)
    and checks the result against the topology the library executes."""
    if not blocks or blocks[0].get("type") != "net":
        raise ValueError("cfg: first section must be [net]")
    net = blocks[0]
    width, height, prev = int(net.get("width", 416)), int(net.get("height", 416)), int(net.get("channels", 3))
    out_filters, specs, yolo = [], [], []
    for idx, b in enumerate(blocks[1:]):
        t = b["type"]
        filters = prev
        if t == "convolutional":
            bn = int(b.get("batch_normalize", 0))
            filters, k, stride = int(b["filters"]), int(b["size"]), int(b["stride"])
            if int(b.get("pad", 0)) != 1 or k not in (1, 3) or stride not in (1, 2):
                raise ValueError("cfg: conv block %d: only pad=1, size 1/3, stride 1/2 are supported" % idx)
            act = b.get("activation", "linear")
            if (bn and act != "leaky") or (not bn and act != "linear"):
                raise ValueError("cfg: conv block %d: expected BN+leaky or bias+linear, got bn=%d %s" % (idx, bn, act))
            specs.append(dict(cin=prev, cout=filters, k=k, stride=stride, bn=bool(bn), block=idx))
        elif t == "shortcut":
            if int(b["from"]) != -3 or b.get("activation", "linear") != "linear":
                raise ValueError("cfg: shortcut block %d: only from=-3, linear (res_block, layer_utils.py:24-29)" % idx)
        elif t == "upsample":
            if int(b["stride"]) != 2:
                raise ValueError("cfg: upsample block %d: only stride 2" % idx)
        elif t == "route":
            layers = [int(x) for x in b["layers"].split(",")]
            layers = [l if l > 0 else idx + l for l in layers]
            filters = sum(out_filters[l] for l in layers)
        elif t == "yolo":
            mask = [int(x) for x in b["mask"].split(",")]
            anc = np.asarray([float(x) for x in b["anchors"].split(",")], np.float32).reshape(-1, 2)
            yolo.append(dict(mask=mask, anchors=anc, classes=int(b["classes"]), block=idx))
        else:
            raise ValueError("cfg: unsupported section [%s]" % t)
        out_filters.append(filters)
        prev = filters
    if len(yolo) != 3 or len({y["classes"] for y in yolo}) != 1:
        raise ValueError("cfg: expected three [yolo] sections with one class count")
    class_num = yolo[0]["classes"]
    anchors = yolo[0]["anchors"]
    if anchors.shape != (9, 2) or any(not np.array_equal(y["anchors"], anchors) for y in yolo) or \
            [y["mask"] for y in yolo] != [[6, 7, 8], [3, 4, 5], [0, 1, 2]]:
        raise ValueError("cfg: expected 9 shared anchors with masks 6,7,8 / 3,4,5 / 0,1,2 (model.py:116-118)")
    want = weights.yolov3_conv_specs(class_num)
    got = [(s["cin"], s["cout"], s["k"], s["stride"], s["bn"]) for s in specs]
    ref = [(s["cin"], s["cout"], s["k"], s["stride"], s["bn"]) for s in want]
    if got != ref:
        bad = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
        raise ValueError("cfg: not the YOLOv3 topology the library executes: %d convs (want %d), first difference at conv %d"
                         % (len(got), len(ref), bad))
    if width % 32 or height % 32:
        raise ValueError("cfg: width / height must be multiples of 32")
    return dict(specs=specs, class_num=class_num, anchors=anchors, width=width, height=height)


def load_cfg(cfgfile_or_text):
    return plan_from_blocks(parse_cfg(cfgfile_or_text))
