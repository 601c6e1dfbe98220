"""CPU-only checks of the C-ABI shared library: it loads without a GPU, exports every symbol include/diffphore_amd.h
declares, and the ctypes struct mirrors have the sizes the C compiler gives them.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from diffphore_amd import _lib, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "diffphore_amd.h")).read()
    names = set(re.findall(r"^\s*int\s+(dp_\w+)\s*\(", hdr, flags=re.M))
    assert {"dp_tp_conv", "dp_forward", "dp_sample_step", "dp_conformer_update", "dp_prepare", "dp_randomize_position", "dp_fit_overlap",
            "dp_pack_conv_weights", "dp_abi_version", "dp_struct_sizes"} <= names
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in the header but not exported"


def test_struct_layouts_match_the_compiler():
    L = _lib.lib()
    out = (C.c_int32 * 6)()
    L.dp_struct_sizes.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    assert L.dp_struct_sizes(out, 6) == 6
    want = [C.sizeof(_lib.DpConvArgs), C.sizeof(engine.DpBatch), C.sizeof(engine.DpWeights), C.sizeof(engine.DpWork),
            C.sizeof(engine.DpStepParams), C.sizeof(engine.DpMlp)]
    assert list(out) == want


def test_bad_arguments_are_rejected_not_crashed():
    L = _lib.lib()
    assert L.dp_tp_conv(None, None) == -1
    a = _lib.DpConvArgs()
    assert L.dp_tp_conv(C.byref(a), None) == -1
    engine._declare()
    assert L.dp_forward(None, None, None, None, None, None, None) == -1
    assert L.dp_pack_conv_weights(99, 0, None, None, None, None, None, None) == -1


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    from diffphore_amd import config, weights
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DiffPhoreLibraryError):
        engine.DeviceModel(weights.init_state_dict(), config.ModelConfig(), "cpu")
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdiffphore_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.DiffPhoreLibraryError):
        _lib.lib()
