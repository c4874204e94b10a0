"""CPU: the C-ABI library loads and exports every symbol include/gsx.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsx_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    assert declared_symbols() == sorted(pkg.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_struct_layouts(pkg):
    assert ctypes.sizeof(pkg.Uniforms) == 224          # main.cpp:48-58
    assert pkg.Uniforms.splatSize.offset == 192 and pkg.Uniforms.bayerSize.offset == 204
    assert ctypes.sizeof(pkg.Timings) == 96          # 8 floats, 2 u32, 1 u64, 2 u32, 2 u64


def test_no_oracle_in_product():
    """The product path must not reach into oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gaussian-splatting_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liborc" not in src and "import orc" not in src and "oracle/" not in src, f
