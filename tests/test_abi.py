"""The C-ABI library loads and exports every symbol include/cerberus_hip.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "cerberus_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cerb_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from cerberus_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `python -m cerberus_amd.build` first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libcerberus_hip.so does not export %s" % n
    # the python binding lists exactly what the header declares
    assert sorted(_lib.EXPORTS) == names


def test_every_binding_declares_its_argument_types():
    """A ctypes entry point without argtypes truncates 64-bit pointers to int: every export taking arguments must set them,
    with the arity of the header's declaration."""
    from cerberus_amd import _lib

    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cerberus_hip.h")).read(), flags=re.S)
    L = _lib.lib()
    for name in _lib.EXPORTS:
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, txt, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        arity = 0 if params in ("", "void") else params.count(",") + 1
        at = getattr(L, name).argtypes
        if arity == 0:
            continue
        assert at is not None and len(at) == arity, "%s: header has %d parameters, binding declares %s" % (name, arity, at)


def test_host_codec_library_exports_what_its_header_declares():
    """libcerberus_host.so (plain C, the slide reader's TIFF LZW / PackBits / predictor codecs): include/cerberus_host.h <-> exports <-> ctypes
    argument types, as for the HIP library."""
    from cerberus_amd import _hostlib

    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cerberus_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(cerb_host_[a-z0-9_]+)\s*\(", txt)))
    assert os.path.exists(_hostlib.LIB_PATH), "run `python -m cerberus_amd.build` first"
    raw = ctypes.CDLL(_hostlib.LIB_PATH)
    assert len(names) == 5 and sorted(_hostlib.EXPORTS) == names
    L = _hostlib.lib()
    for n in names:
        assert hasattr(raw, n), "libcerberus_host.so does not export %s" % n
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % n, txt, flags=re.S)
        params = m.group(1).strip()
        arity = 0 if params in ("", "void") else params.count(",") + 1
        at = getattr(L, n).argtypes
        assert (at is None or len(at) == 0) if arity == 0 else (at is not None and len(at) == arity), (n, arity, at)
    assert L.cerb_host_version() >= 1


def test_version_and_error_string():
    from cerberus_amd import _lib

    L = _lib.lib()
    assert L.cerb_version() >= 1
    assert isinstance(L.cerb_last_error(), bytes)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under cerberus_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "cerberus_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f
