"""Counterpart of the reference's single-image driver (yolo_v3/test_single_image.py:23-45), SURVEY.md 8a A11:
preprocess -> forward -> predict -> conf*prob -> gpu_nms(30, 0.4, 0.5) -> boxes back to the original image's
pixels.  The reference script is never run; this module reproduces its call order, thresholds and rescale on the
GPU path.  Image decoding (cv2.imread) and drawing (plot_one_box / imshow) are outside the hot path: `main` reads
the file with PIL when it is present, in BGR order as cv2.imread would give it.

    python -m yolo_amd.test_single_image image.jpg --weights yolov3.weights --anchors yolo_anchors.txt --names coco.names
"""
from __future__ import division, print_function

import argparse

import numpy as np

from .model import yolov3
from .preprocess import preprocess
from .shard import pack_detections_device
from .utils.misc_utils import DEFAULT_ANCHORS, parse_anchors, read_class_names


def detect_image(yolo_model, img_ori, new_size=(416, 416), max_boxes=30, score_thresh=0.4, iou_thresh=0.5):
    """img_ori: uint8 [h, w, 3] BGR (cv2.imread) -> (boxes_ [m, 4] in img_ori pixels, scores_ [m], labels_ [m] int32),
    numpy, as the reference's sess.run returns them (test_single_image.py:40-45)."""
    boxes, scores, labels = detect_frames(yolo_model, np.asarray(img_ori)[np.newaxis], new_size, max_boxes,
                                          score_thresh, iou_thresh)[0]
    return boxes, scores, labels


def detect_frames(yolo_model, frames, new_size=(416, 416), max_boxes=30, score_thresh=0.4, iou_thresh=0.5):
    """The same chain for a batch of equally sized frames (the per-frame loop of video_test.py:58-71 as one
    step): frames uint8 [B, h, w, 3] BGR -> list of B (boxes_, scores_, labels_) numpy triples."""
    frames = frames if hasattr(frames, "shape") else np.asarray(frames)
    height_ori, width_ori = int(frames.shape[1]), int(frames.shape[2])
    img = preprocess(frames, new_size)                                                    # :25-28
    ob, osc, ol, _, oc = yolo_model.detect(img, max_boxes, score_thresh, iou_thresh)      # :31-40
    # :42-45 on the GPU: boxes_[:, 0|2] *= width_ori / new_w, boxes_[:, 1|3] *= height_ori / new_h (float32 multiplies)
    B = int(ob.shape[0])
    scale = np.tile(np.asarray([width_ori / float(new_size[0]), height_ori / float(new_size[1])], np.float32), (B, 1))
    rows, offsets = pack_detections_device(ob, osc, ol, oc, scale)
    rows, offsets = rows.cpu().numpy(), offsets.cpu().numpy()
    out = []
    for b in range(B):
        r = rows[offsets[b]:offsets[b + 1]]
        out.append((r[:, 1:5].copy(), r[:, 5].copy(), r[:, 6].astype(np.int32)))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="YOLOv3 single image detection on MI355X")
    ap.add_argument("input_image")
    ap.add_argument("--weights", required=True, help="Darknet yolov3.weights")
    ap.add_argument("--anchors", default="", help="anchor file (data/yolo_anchors.txt layout); default: the COCO anchors")
    ap.add_argument("--names", default="", help="class name file, one per line; default: 80 numbered classes")
    ap.add_argument("--new-size", type=int, nargs=2, default=[416, 416], metavar=("W", "H"))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args(argv)
    anchors = parse_anchors(args.anchors) if args.anchors else DEFAULT_ANCHORS
    classes = read_class_names(args.names) if args.names else {i: str(i) for i in range(80)}
    try:
        from PIL import Image
    except ImportError:
        raise SystemExit("reading image files needs PIL; pass arrays to detect_image() instead")
    img_ori = np.asarray(Image.open(args.input_image).convert("RGB"))[:, :, ::-1].copy()      # BGR like cv2.imread
    yolo_model = yolov3(len(classes), anchors, dtype=args.dtype).load_darknet_weights(args.weights)
    boxes_, scores_, labels_ = detect_image(yolo_model, img_ori, tuple(args.new_size))
    print("box coords:")
    print(boxes_)
    print('*' * 30)
    print("scores:")
    print(scores_)
    print('*' * 30)
    print("labels:")
    print(labels_)
    for i in range(len(boxes_)):
        print(classes[int(labels_[i])], boxes_[i].tolist())


if __name__ == "__main__":
    main()
