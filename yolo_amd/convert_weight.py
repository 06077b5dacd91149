# coding: utf-8
"""Counterpart of the reference's yolo_v3/convert_weight.py:15-31: Darknet `yolov3.weights` -> a checkpoint of this
build.  The reference restores a TF checkpoint written by that script; here the checkpoint is an .npz whose arrays
carry the same variable names and layouts (HWIO kernels, BatchNorm gamma / beta / moving_mean / moving_variance,
biases), loaded with `yolov3.load_state`.  (`yolov3.load_darknet_weights` reads the .weights file directly.)

    python -m yolo_amd.convert_weight yolov3.weights yolov3.npz [--num-class 80]
"""
from __future__ import division, print_function

import argparse

import numpy as np

from . import weights
from .utils.misc_utils import read_darknet_weights


def convert(weight_path, save_path, num_class=80):
    stream = read_darknet_weights(weight_path)
    need = weights.stream_length(weights.yolov3_conv_specs(num_class))
    if stream.size != need:
        raise ValueError("%s holds %d floats, YOLOv3 with %d classes needs %d" % (weight_path, stream.size, num_class, need))
    state = weights.stream_to_state(stream, num_class)
    np.savez(save_path, **state)
    return state


def main(argv=None):
    ap = argparse.ArgumentParser(description="Darknet yolov3.weights -> .npz checkpoint in the reference's variable naming")
    ap.add_argument("weight_path")
    ap.add_argument("save_path")
    ap.add_argument("--num-class", type=int, default=80)
    args = ap.parse_args(argv)
    state = convert(args.weight_path, args.save_path, args.num_class)
    print("checkpoint with %d variables has been saved to %s" % (len(state), args.save_path))


if __name__ == "__main__":
    main()
