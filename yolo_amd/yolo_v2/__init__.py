"""YOLOv2 inference path (config 4) with the module layout of reference yolo_v2/: `yolo.config`,
`yolo.yolo_v2.yolo_v2`, `Detector.Detector`."""
