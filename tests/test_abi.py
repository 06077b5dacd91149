"""CPU: the C-ABI library loads and exports every symbol include/yolo_hip.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yolo2?_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from yolo_amd import _lib
    names = _declared()
    assert len(names) >= 15
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C yolo_amd/csrc"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libyolo_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "%s has no ctypes signature in yolo_amd/_lib.py" % n
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().yolo_abi_version() == 1


def test_argument_errors_without_gpu():
    """Error paths that return before any HIP call."""
    from yolo_amd import _lib
    L = _lib.lib()
    assert L.yolo_nms_workspace_bytes(0, 10, 80, 50) == 0
    assert L.yolo_nms_workspace_bytes(2, 10647, 80, 50) >= 2 * 80 * 10647 * 8
    rc = L.yolo_nms(None, None, 1, 10, 80, 50, 0.5, 0.5, 0, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"null" in L.yolo_last_error()
    rc = L.yolo_decode(None, None, None, 1, 416, 416, 80, None, None, None, None, None, None)
    assert rc == -1
    assert L.yolo2_nms_workspace_bytes(2, 845, 20) >= 2 * 845 * 20 * 8
    assert L.yolo2_decode(None, 1, 13, 5, 20, 416, None, None, None, None) == -1
    assert L.yolo_maxpool2x2(None, 1, 4, 4, 8, None, None) == -1
    assert L.yolo_reorg2x(None, 1, 4, 4, 8, None, None) == -1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "yolo_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.replace("the CPU oracle", ""), os.path.join(dp, fn)
