"""Host helpers mirroring reference yolo_v3/utils/misc_utils.py (inference-relevant part only)."""
import numpy as np

# The 9 COCO anchors (w, h) at network scale, as in the reference's data/yolo_anchors.txt:1.
DEFAULT_ANCHORS = np.asarray([[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119],
                              [116, 90], [156, 198], [373, 326]], np.float32)

COCO_NAMES = (
    "person bicycle car motorbike aeroplane bus train truck boat|traffic light|fire hydrant|stop sign|"
    "parking meter|bench bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella "
    "handbag tie suitcase frisbee skis snowboard|sports ball|kite|baseball bat|baseball glove|"
    "skateboard surfboard|tennis racket|bottle|wine glass|cup fork knife spoon bowl banana apple "
    "sandwich orange broccoli carrot|hot dog|pizza donut cake chair sofa pottedplant bed diningtable "
    "toilet tvmonitor laptop mouse remote keyboard|cell phone|microwave oven toaster sink refrigerator "
    "book clock vase scissors|teddy bear|hair drier|toothbrush")


def coco_class_names():
    """{id: name} for the 80 COCO classes (same ids as the reference's data/coco.names)."""
    names = []
    for part in COCO_NAMES.split("|"):
        part = part.strip()
        if " " in part and part in ("traffic light", "fire hydrant", "stop sign", "parking meter", "sports ball",
                                    "baseball bat", "baseball glove", "tennis racket", "wine glass", "hot dog",
                                    "cell phone", "teddy bear", "hair drier"):
            names.append(part)
        else:
            names.extend(part.split())
    assert len(names) == 80, len(names)
    return dict(enumerate(names))


def parse_anchors(anchor_path):
    """misc_utils.py:10-13: comma separated numbers -> float32 [N, 2] (w, h)."""
    with open(anchor_path, "r") as f:
        return np.reshape(np.asarray(f.read().split(","), np.float32), [-1, 2])


def read_class_names(class_name_path):
    """misc_utils.py:15-21: one name per line -> {id: name}."""
    with open(class_name_path, "r") as f:
        return {idx: line.strip("\n") for idx, line in enumerate(f)}


def read_darknet_weights(weights_file):
    """Float32 body of a Darknet .weights file: 5 x int32 header skipped (misc_utils.py:52-55)."""
    with open(weights_file, "rb") as fp:
        np.fromfile(fp, dtype=np.int32, count=5)
        return np.fromfile(fp, dtype=np.float32)
