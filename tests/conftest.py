import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch


@pytest.fixture(scope="session")
def oracle():
    from oracle import yolo_oracle
    return yolo_oracle


@pytest.fixture(scope="session")
def weight_stream(oracle):
    """The seeded synthetic Darknet-order stream every conv fixture was generated with."""
    specs = oracle.yolov3_conv_specs(80)
    return oracle.synth_darknet_stream(specs, seed=1234)


@pytest.fixture(scope="session")
def oracle_params(oracle, weight_stream):
    return oracle.split_darknet_stream(oracle.yolov3_conv_specs(80), weight_stream)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_rms_ok(a, b, tol=1e-3):
    """|a-b| <= tol*|b| + tol*rms(b) elementwise (the fp32 parity metric)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    rms = np.sqrt(np.mean(b * b))
    err = np.abs(a - b) - tol * np.abs(b) - tol * rms
    return float(err.max()) <= 0.0, float(np.abs(a - b).max()), float(rms)
