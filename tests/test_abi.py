"""The C-ABI library loads and exports everything include/scarplet_hip.h
declares; struct layouts seen by ctypes equal the C compiler's.  No compute
calls (there is no GPU on the CPU test box)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from scarplet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scarplet_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table out of sync with the header"


def test_host_library_exports_every_declared_symbol():
    """libscarplet_host.so (include/scarplet_host.h): plain C, linked against nothing of ROCm."""
    from scarplet_amd import _hostlib
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "scarplet_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(sch_[a-z0-9_]+)\s*\(", src)))
    lib = ctypes.CDLL(_hostlib.LIB_PATH)
    assert names and sorted(_hostlib.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    needed = subprocess.check_output(["readelf", "-d", _hostlib.LIB_PATH]).decode()
    assert "amdhip" not in needed and "rccl" not in needed


def test_load_binds_and_reports_version():
    lib = _lib.load()
    assert lib.sc_abi_version() == _lib.ABI_VERSION
    assert lib.sc_kernel_name(_lib.K_INV_ROWS) == b"k_inv_rows"


def test_struct_layouts_match_c(tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "scarplet_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(sc_template), offsetof(sc_template, cc),
         offsetof(sc_template, ilo), offsetof(sc_template, id), sizeof(sc_plan), sizeof(sc_xfer));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    T = _lib.sc_template
    assert vals == [ctypes.sizeof(T), T.cc.offset, T.ilo.offset, T.id.offset,
                    ctypes.sizeof(_lib.sc_plan), ctypes.sizeof(_lib.sc_xfer)]


def test_product_fails_loudly_without_gpu():
    if os.path.exists("/dev/kfd"):          # no HIP call here: later tests start subprocesses
        pytest.skip("a GPU box")
    lib = _lib.load()
    if lib.sc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.ScarpletHipError):
        _lib.Context(0)
    import numpy as np
    import scarplet_amd as sl
    g = sl.DEMGrid.from_array(np.zeros((32, 32)), 1.0)
    with pytest.raises(_lib.ScarpletHipError):
        sl.match(g, sl.Scarp, scale=5, age=10.)


def test_product_does_not_import_the_oracle():
    code = ("import sys; import scarplet_amd, scarplet_amd.dist, scarplet_amd.synthetic; "
            "assert not [m for m in sys.modules if 'oracle' in m], 'oracle imported'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scarplet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "scarplet_oracle" not in open(os.path.join(dirpath, f)).read(), f
