"""Darknet `.cfg` ingestion for the YOLOv3 path: the text format the reference's PyTorch lesson reads
(yolo-v3/002/darknet.py:12-36 `parse_cfg`, cfg/yolov3.cfg) -> block list -> the conv specs / anchors / class count that
`yolov3` is built from.  The execution plan of libyolo_hip.so is the static YOLOv3 topology (yolo_v3/model.py:21-67 =
cfg/yolov3.cfg:607-786); `plan_from_blocks` therefore VALIDATES that a cfg describes exactly that topology - every block's
type, every conv's geometry, every shortcut / route source (resolved to absolute block indices) and the position of the
two upsample blocks - for any class count, anchors and multiple-of-32 size, and raises otherwise.  It does not build
arbitrary networks."""
import re

import numpy as np

from . import weights

_SECTION = re.compile(r"^\[([^\]]*)\]$")


def parse_cfg(cfgfile_or_text):
    """cfg file path, or the text itself (anything containing a newline) -> list of dicts, one per [section], with the
    section name under 'type' and every `key = value` line of the section as stripped strings.  Blank lines and lines whose
    first non-blank character is '#' are skipped (the same result as 002/darknet.py:12-36 on its cfg/yolov3.cfg: pinned by
    tests/golden/g6_yolov3_cfg_blocks.json)."""
    if "\n" in cfgfile_or_text:
        text = cfgfile_or_text
    else:
        with open(cfgfile_or_text, "r") as f:
            text = f.read()
    blocks = []
    for lineno, raw in enumerate(text.splitlines(), 1):
        entry = raw.strip()
        if not entry or entry.startswith("#"):
            continue
        header = _SECTION.match(entry)
        if header is not None:
            blocks.append({"type": header.group(1).strip()})
            continue
        name, sep, value = entry.partition("=")
        if not sep:
            raise ValueError("cfg line %d: expected key=value, got %r" % (lineno, entry))
        if not blocks:
            raise ValueError("cfg line %d: %r comes before the first [section]" % (lineno, entry))
        blocks[-1][name.strip()] = value.strip()
    return blocks


def blocks_to_text(blocks):
    """Inverse of parse_cfg (used by tests and to write a cfg for a re-classed model)."""
    out = []
    for b in blocks:
        out.append("[%s]" % b["type"])
        out.extend("%s=%s" % (k, v) for k, v in b.items() if k != "type")
        out.append("")
    return "\n".join(out)


def expected_layout(class_num):
    """The block sequence (after [net]) of the topology libyolo_hip.so executes, with absolute source indices:
    ('convolutional', cin, cout, k, stride, bn) | ('shortcut', src) | ('route', [src, ...]) | ('upsample',) | ('yolo', mask).
    Backbone = darknet53_body (layer_utils.py:23-54), head = model.py:41-64 in the block order of cfg/yolov3.cfg:607-786."""
    specs = iter(weights.yolov3_conv_specs(class_num))
    out = []

    def conv():
        s = next(specs)
        out.append(("convolutional", s["cin"], s["cout"], s["k"], s["stride"], bool(s["bn"])))

    def res_blocks(n):
        for _ in range(n):
            conv(); conv()
            out.append(("shortcut", len(out) - 3))             # block input (layer_utils.py:26-28)

    conv()
    routes = []
    for n in (1, 2, 8, 8, 4):
        conv()                                                  # stride-2 downsample
        res_blocks(n)
        routes.append(len(out) - 1)                             # route_1 / route_2 / route_3 candidates
    route_1, route_2 = routes[2], routes[3]                     # blocks 36 and 61 of cfg/yolov3.cfg
    for scale, (mask, lateral) in enumerate((([6, 7, 8], route_2), ([3, 4, 5], route_1), ([0, 1, 2], None))):
        for _ in range(6):
            conv()                                              # yolo_block (layer_utils.py:56-65)
        conv()                                                  # detection conv: bias, linear
        out.append(("yolo", mask))
        if lateral is None:
            break
        out.append(("route", [len(out) - 4]))                   # the yolo_block's fifth conv (`route`, model.py:41,51)
        conv()                                                  # inter conv (model.py:47,57)
        out.append(("upsample",))
        out.append(("route", [len(out) - 1, lateral]))          # concat [upsampled, route_k] (model.py:49,59)
    return out


def _describe(e):
    return e[0] if len(e) == 1 else "%s %s" % (e[0], ", ".join(str(v) for v in e[1:]))


def plan_from_blocks(blocks):
    """-> dict(specs=[dict(cin, cout, k, stride, bn, block)] in file order, class_num, anchors float32 [9, 2], width, height).
    Walks the blocks with the filter bookkeeping of 002/darknet.py:55-148 `create_modules`, resolves every relative
    index, and compares the whole sequence with `expected_layout`."""
    if not blocks or blocks[0].get("type") != "net":
        raise ValueError("cfg: first section must be [net]")
    net = blocks[0]
    width, height, prev = int(net.get("width", 416)), int(net.get("height", 416)), int(net.get("channels", 3))
    out_filters, specs, yolo, seen = [], [], [], []
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
            seen.append(("convolutional", prev, filters, k, stride, bool(bn)))
        elif t == "shortcut":
            if b.get("activation", "linear") != "linear":
                raise ValueError("cfg: shortcut block %d: only a linear shortcut (res_block, layer_utils.py:24-29)" % idx)
            src = int(b["from"])
            seen.append(("shortcut", src if src >= 0 else idx + src))
        elif t == "upsample":
            if int(b["stride"]) != 2:
                raise ValueError("cfg: upsample block %d: only stride 2" % idx)
            seen.append(("upsample",))
        elif t == "route":
            layers = [int(x) for x in b["layers"].split(",")]
            layers = [l if l > 0 else idx + l for l in layers]      # 002/darknet.py:108-125: positive = absolute
            if any(l < 0 or l >= idx for l in layers):
                raise ValueError("cfg: route block %d points outside the blocks before it" % idx)
            filters = sum(out_filters[l] for l in layers)
            seen.append(("route", layers))
        elif t == "yolo":
            mask = [int(x) for x in b["mask"].split(",")]
            anc = np.asarray([float(x) for x in b["anchors"].split(",")], np.float32).reshape(-1, 2)
            yolo.append(dict(mask=mask, anchors=anc, classes=int(b["classes"]), block=idx))
            seen.append(("yolo", mask))
        else:
            raise ValueError("cfg: unsupported section [%s]" % t)
        out_filters.append(filters)
        prev = filters
    if len(yolo) != 3 or len({y["classes"] for y in yolo}) != 1:
        raise ValueError("cfg: expected three [yolo] sections with one class count")
    class_num = yolo[0]["classes"]
    anchors = yolo[0]["anchors"]
    if anchors.shape != (9, 2) or any(not np.array_equal(y["anchors"], anchors) for y in yolo):
        raise ValueError("cfg: expected 9 anchors shared by the three [yolo] sections (model.py:116-118)")
    want = expected_layout(class_num)
    if seen != want:
        bad = next((i for i, (a, w) in enumerate(zip(seen, want)) if a != w), min(len(seen), len(want)))
        got_s = _describe(seen[bad]) if bad < len(seen) else "end of file"
        want_s = _describe(want[bad]) if bad < len(want) else "end of file"
        raise ValueError("cfg: not the YOLOv3 topology the library executes: %d blocks (want %d); block %d is [%s], expected [%s]"
                         % (len(seen), len(want), bad, got_s, want_s))
    if width % 32 or height % 32:
        raise ValueError("cfg: width / height must be multiples of 32")
    return dict(specs=specs, class_num=class_num, anchors=anchors, width=width, height=height)


def load_cfg(cfgfile_or_text):
    return plan_from_blocks(parse_cfg(cfgfile_or_text))
