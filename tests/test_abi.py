"""CPU: the C-ABI shared library loads and exports every symbol include/panoptic_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "panoptic_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from panopticsegforlargescalepointcloud_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    # the ctypes table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.load().pp_version().decode().startswith("panoptic_hip")
    assert _lib.load().pp_hash_capacity(1000) == 2048


def test_ops_fail_loudly_without_gpu():
    import torch
    from panopticsegforlargescalepointcloud_amd import ops, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PanopticHipError):
        ops.hash_build(torch.zeros((4, 4), dtype=torch.int32))
