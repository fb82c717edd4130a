"""The shared library must load without a GPU and export every function that
include/pqdecode.h declares; without a device its entry points must fail loudly."""
import ctypes as C
import os
import re

import pytest

from parquet_go_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pqdecode.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pq_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.pq_version() >= 100


def test_struct_layouts_match_header():
    # sizes the kernels and the Go/cgo binding rely on
    assert C.sizeof(capi.PageDesc) == 96 and C.sizeof(capi.PageResult) == 32
    assert capi.DESC_DTYPE.fields["dict_off"][1] == 48 and capi.DESC_DTYPE.fields["n_values"][1] == 56
    assert capi.RESULT_DTYPE.fields["consumed"][1] == 24


def test_status_strings():
    assert capi.status_string(capi.OK) == "ok"
    assert capi.status_string(capi.EOF) == "EOF"
    assert "miniblock" in capi.status_string(capi.INVALID_MINIBLOCK_WIDTH)


def test_no_device_means_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.PqError) as e:
        capi.Context(0)
    assert e.value.code in (capi.ERR_NO_DEVICE, capi.ERR_HIP)
