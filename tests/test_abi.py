"""CPU-only: the C-ABI library loads and exports every symbol include/neunet_hip.h declares, and the ctypes
signature table of the host package covers exactly that set.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "neunet_hip.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nnhip[A-Za-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import neunet_hip
    from neunet_hip import _lib
    path = _lib.lib_path()
    if not os.path.exists(path):
        import importlib.util
        spec = importlib.util.spec_from_file_location("nnhip_build", os.path.join(ROOT, "numpy-nn-model_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return neunet_hip.load_library()


def test_header_declares_expected_surface():
    syms = header_symbols()
    for must in ["nnhipLinearModuleForward", "nnhipLinearModuleBackward", "nnhipLinearSwishForward",
                 "nnhipLinearSwishBackward", "nnhipSwishForward", "nnhipSwishBackward", "nnhipFusedSwishAndMul",
                 "nnhipFusedSwishAndMulBackward", "nnhipSoftmaxForward", "nnhipSoftmaxBackward",
                 "nnhipCrossEntropyForwardBackward", "nnhipRMSNormForward", "nnhipRMSNormBackward",
                 "nnhipFusedAdamWStep", "nnhipCreateFusedOptimizer", "nnhipDestroyFusedOptimizer",
                 "nnhipFusedAdamWMultiTensorStep", "nnhipConv2dForward", "nnhipConv2dBackward", "nnhipCleanup"]:
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/neunet_hip.h but not exported: {missing}"


def test_ctypes_table_matches_header(lib):
    from neunet_hip import _lib
    assert sorted(_lib.exported_symbols()) == header_symbols()
    for name in _lib.exported_symbols():
        _lib.load_hip_function(name)  # argtypes bind


def test_version_and_error_string(lib):
    from neunet_hip import _lib
    assert _lib.load_hip_function("nnhipVersion")() >= 100
    assert isinstance(_lib.last_error(), str)


def test_argument_errors_are_status_codes_not_exits(lib):
    """Bad arguments come back as negative status (the reference printf+exit()s)."""
    from neunet_hip import _lib
    with pytest.raises(_lib.NeunetHipError, match="negative size"):
        _lib.call_hip_function("nnhipSwishForward", 16, 16, 1.0, -1, None)
    with pytest.raises(_lib.NeunetHipError, match="null"):
        _lib.call_hip_function("nnhipLinearModuleForward", None, None, None, None, 4, 4, 4, None)
    with pytest.raises(_lib.NeunetHipError, match="reduction"):
        _lib.call_hip_function("nnhipCrossEntropyForwardBackward", 16, 16, 16, 16, 4, -100, 2, 4, b"x", 1, None, None, None)


def test_numpy_arrays_rejected_like_reference():
    """utils.py:75-76 of the reference: to_pointer raises TypeError on NumPy input."""
    import numpy as np
    from neunet_hip import _lib
    with pytest.raises(TypeError):
        _lib.to_pointer(np.zeros(4, np.float32))


def test_missing_library_fails_loudly(monkeypatch):
    from neunet_hip import _lib
    monkeypatch.setattr(_lib, "_dll", None)
    monkeypatch.setattr(_lib, "_funcs", {})
    monkeypatch.setenv(_lib.LIB_ENV, "/nonexistent/libneunet_hip.so")
    with pytest.raises(_lib.NeunetHipError, match="no CPU fallback"):
        _lib.load_hip_function("nnhipVersion")


def test_comm_unique_id_needs_no_gpu(lib):
    """The RCCL entry points bind librccl lazily (dlopen): the id for nnhipCommInitRank can be drawn on a GPU-less host, bad
    arguments are status codes, and the library that got bound is an RCCL."""
    import ctypes
    from neunet_hip import _lib
    buf = ctypes.create_string_buffer(128)
    try:
        _lib.call_hip_function("nnhipCommUniqueId", buf)
    except _lib.NeunetHipError as exc:
        pytest.skip(f"no RCCL on this host: {exc}")
    assert any(buf.raw), "ncclGetUniqueId left the buffer untouched"
    path, ver = ctypes.create_string_buffer(256), ctypes.c_int(0)
    _lib.call_hip_function("nnhipCommLibrary", path, 256, ctypes.byref(ver))
    assert b"rccl" in path.value and ver.value > 0
    with pytest.raises(_lib.NeunetHipError, match="null communicator"):
        _lib.call_hip_function("nnhipAllReduceSumF32", None, 16, 4, None)
    with pytest.raises(_lib.NeunetHipError, match="rank"):
        h = ctypes.c_void_p()
        _lib.call_hip_function("nnhipCommInitRank", ctypes.byref(h), buf.raw, 3, 2)
    assert _lib.call_hip_function("nnhipCommDestroy", None) == 0
