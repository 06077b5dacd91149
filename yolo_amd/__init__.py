"""yolo_amd - MI355X-native YOLOv3 inference hot path behind the FenHua/yolo `yolo_v3` Python API.

    from yolo_amd.model import yolov3
    from yolo_amd.utils.nms_utils import gpu_nms, cpu_nms, py_nms
    from yolo_amd.utils.misc_utils import parse_anchors, read_class_names
"""
__all__ = ["model", "utils"]
