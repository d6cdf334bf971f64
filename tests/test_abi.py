"""CPU: the C-ABI library loads and exports every symbol include/lfi.h declares (no compute calls)."""
import ctypes
import os
import re

from lets_face_it_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lfi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_binding_covers_header():
    assert sorted(_lib.EXPORTS) == declared_symbols()
    L = _lib.lib()
    assert L.lfi_version() >= 100
    for name in _lib.EXPORTS:
        assert getattr(L, name).argtypes is not None, name


def test_argument_errors_are_reported_not_crashes():
    L = _lib.lib()
    assert L.lfi_gemm_f32(None, None) == -1
    assert b"null descriptor" in L.lfi_last_error()
    d = _lib.FlowDims(4, 0, 16, 32, 32, 2, 1, 1, 1e-4)  # N = 0
    assert L.lfi_flow_stash_floats(ctypes.byref(d)) == -1
    assert b"bad dims" in L.lfi_last_error()
    # the LSTM cell stashes its cell state on top of what the GRU cell keeps (host-side arithmetic only)
    gru, lstm = _lib.FlowDims(4, 2, 16, 32, 32, 2, 1, 0, 1e-4), _lib.FlowDims(4, 2, 16, 32, 32, 2, 1, 1, 1e-4)
    assert L.lfi_flow_stash_floats(ctypes.byref(lstm)) == L.lfi_flow_stash_floats(ctypes.byref(gru)) + 2 * 8 * 32


def _struct_body(text, name):
    head = text[:text.index("} %s;" % name)]
    body = head[head.rindex("typedef struct {") + len("typedef struct {"):]
    return re.sub(r"/\*.*?\*/", "", body, flags=re.S)


def test_struct_layouts_match_header_field_order():
    text = open(os.path.join(ROOT, "include", "lfi.h")).read()
    names = re.findall(r"[\s\*,]([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", _struct_body(text, "lfi_gemm_desc"))
    assert names == [f[0] for f in _lib.GemmDesc._fields_]
    names = re.findall(r"[\s\*,]([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", _struct_body(text, "lfi_pgemm_desc"))
    assert names == [f[0] for f in _lib.PGemmDesc._fields_]
    names = re.findall(r"[\s\*,]([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", _struct_body(text, "lfi_enc_desc"))
    assert names == [f[0] for f in _lib.EncDesc._fields_]
    names = re.findall(r"[\s\*,]([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", _struct_body(text, "lfi_flow_dims"))
    assert names == [f[0] for f in _lib.FlowDims._fields_]
    names = re.findall(r"\*\s*([a-z_]+)", _struct_body(text, "lfi_flow_params"))
    assert names == [f[0] for f in _lib.FlowParams._fields_]
    names = re.findall(r"\*\s*([a-z_]+)", _struct_body(text, "lfi_flow_grads"))
    assert names == [f[0] for f in _lib.FlowGrads._fields_]
