"""The C-ABI library loads and exports every symbol include/btr_pointnet2.h declares, and the
Python shim binds exactly those (no compute: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "btr_pointnet2.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(btr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from backtoreality_amd import build
    lib = ctypes.CDLL(build.build())
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libbtr_pointnet2.so lacks %s" % n
    lib.btr_abi_version.restype = ctypes.c_int
    assert lib.btr_abi_version() == 1
    lib.btr_opt_n_threads.restype = ctypes.c_int
    assert [lib.btr_opt_n_threads(n) for n in (1, 3, 511, 512, 40000)] == [1, 2, 256, 512, 512]


def test_shim_binds_every_declared_symbol():
    from backtoreality_amd.pointnet2 import _ext
    assert sorted(_ext._SIGNATURES) == _declared()
    for name in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
                 "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
                 "group_points_grad"):  # bindings.cpp:11-24
        assert callable(getattr(_ext, name))


def test_invalid_arguments_are_reported_not_fatal():
    from backtoreality_amd.pointnet2 import _ext
    rc = _ext._lib.btr_furthest_point_sampling_bs(1, 10, 2, None, None, None, 3, None)
    assert rc == -1 and b"null pointer" in _ext._lib.btr_last_error()


def test_no_oracle_in_product():
    """The product tree never references the oracle (CPU fallback voids parity claims)."""
    pkg = os.path.join(ROOT, "backtoreality_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libbtr_oracle" not in src, f
