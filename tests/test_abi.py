"""The C-ABI library loads and exports every symbol include/ovm3d.h declares (no GPU compute calls)."""
import ctypes as C
import os
import re

from common import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "ovm3d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ovm_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol():
    from ovmono3d_amd import lib
    L = lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libovm3d.so does not export {n}"
    assert sorted(lib.EXPORTS) == names, "lib.EXPORTS out of sync with include/ovm3d.h"


def test_struct_layouts_match_header():
    from ovmono3d_amd import lib
    # OvmDet3D is 48 x 4 bytes; OvmImage / OvmTensor as declared
    assert lib.OVM_REC_FLOATS * 4 == 192
    assert C.sizeof(lib.OvmImage) == 8 + 4 + 4 + 8 * 3 + 4 + 4 + 36 + 4          # trailing pad to 8
    assert C.sizeof(lib.OvmTensor) == 8 + 8 + 8 + 32
    assert C.sizeof(lib.OvmConfig) == 4 * 7 + 12 + 12 + 4 * 5 + 4 + 16 + 12 + 4 * 2 + 4 + 4 + 4 + 4 + 4 * 3 + 4 * 3
    # and the library agrees with every mirror (lib.load() also refuses a mismatch)
    L = lib.load()
    for name, mirror in (("OvmConfig", lib.OvmConfig), ("OvmTensor", lib.OvmTensor), ("OvmImage", lib.OvmImage),
                         ("OvmGdinoConfig", lib.OvmGdinoConfig)):
        assert L.ovm_abi_sizeof(name.encode()) == C.sizeof(mirror), name
    assert L.ovm_abi_sizeof(b"OvmDet3D") == 192 and L.ovm_abi_sizeof(b"nope") == -1


def test_version_and_error_strings():
    from ovmono3d_amd import lib
    L = lib.load()
    assert b"libovm3d" in L.ovm_version()
    assert L.ovm_last_error(None) == b"null handle"


def test_create_rejects_bad_config_without_gpu_work():
    """Argument validation happens before any device call."""
    from ovmono3d_amd import lib
    L = lib.load()
    cfg = lib.OvmConfig()
    cfg.embed_dim, cfg.depth, cfg.heads, cfg.canvas = 100, 1, 1, 225     # canvas not a multiple of 14
    cfg.precision, cfg.max_batch, cfg.max_rois, cfg.fpn_channels = 1, 1, 1, 256
    h = C.c_void_p()
    rc = L.ovm_create(C.byref(cfg), None, 0, 0, C.byref(h))
    assert rc == -1
    assert b"invalid config" in L.ovm_last_error(h)
    L.ovm_destroy(h)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ovmono3d_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        lib.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the shared object is absent")


def test_every_tune_key_is_documented_and_unknown_keys_are_refused():
    """ovm_tune_set is part of the C surface: every key csrc/ops.hip accepts is described in include/ovm3d.h, and a key it does not
    know is an error, not a silent no-op (a typo in OVM_TUNE must not pass for a measurement of the default)."""
    from ovmono3d_amd import lib
    ops = open(os.path.join(ROOT, "ovmono3d_amd", "csrc", "ops.hip")).read()
    hdr = open(os.path.join(ROOT, "include", "ovm3d.h")).read()
    keys = sorted(set(re.findall(r'!strcmp\(key, "([a-z0-9_]+)"\)', ops)))
    assert len(keys) >= 25
    missing = [k for k in keys if not re.search(r"\b" + k + r"\b", hdr)]
    assert not missing, f"tune keys without a line in include/ovm3d.h: {missing}"
    L = lib.load()
    assert L.ovm_tune_set(b"no_such_knob", 1) != 0
    assert L.ovm_tune_set(None, 1) != 0
