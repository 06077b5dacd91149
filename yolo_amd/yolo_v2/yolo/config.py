# coding: utf-8
"""Inference constants of reference yolo_v2/yolo/config.py:8-25 (training-only entries are omitted)."""
CLASSES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
           'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']   # VOC, :8-11
# five anchor widths followed by the five anchor heights, in cells (:13)
ANCHOR = [0.57273, 1.87446, 3.33843, 7.88282, 9.77052, 0.677385, 2.06253, 5.47434, 3.52778, 9.16828]
IMAGE_SIZE = 416       # :16
BOX_PRE_CELL = 5       # :21 (spelling as in the reference)
CELL_SIZE = 13         # :22
BATCH_SIZE = 10        # :23
ALPHA = 0.1            # :24
THRESHOLD = 0.3        # :25
