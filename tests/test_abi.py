"""The C-ABI libraries load on a CPU-only box and export every symbol the headers under include/ declare (no compute
calls here); the ctypes signature tables cover exactly those symbols."""
import ctypes
import os
import re

import pytest

from chameleon_recsys_amd import _lib, _tfrecord

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cham_\w+)\s*\(", src)))


@pytest.mark.parametrize("header,mod", [("chameleon_nar.h", _lib), ("chameleon_tfrecord.h", _tfrecord)])
def test_library_exports_every_declared_symbol(header, mod):
    names = _declared(header)
    assert len(names) >= 10
    assert os.path.exists(mod.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    lib = ctypes.CDLL(mod.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "%s declared in include/%s but not exported by %s" % (n, header, os.path.basename(mod.LIB_PATH))
    assert sorted(mod.EXPORTED_SYMBOLS) == names, set(mod.EXPORTED_SYMBOLS) ^ set(names)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libchameleon_nar.so"))
    with pytest.raises(_lib.ChameleonLibError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "chameleon_recsys_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)
