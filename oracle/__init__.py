"""CPU oracle for the pointnet2 hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``backtoreality_amd``) never does.  See ``pointnet2_oracle.c`` for the
parity status of each op and the reference file:line every function follows.

Two layers:
  * ``lib()``       -- the ctypes handle on ``libbtr_oracle.so`` (numpy-level calls below);
  * ``ext_cpu``     -- an object with the nine callables of the reference's ``pointnet2._ext``
                       (bindings.cpp:11-24) over CPU torch tensors, so the reference-shaped
                       Python layers can run on the host as the checker / CPU baseline.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)
_c = ctypes.c_int


_NAMES = {1: "libbtr_oracle.so", 0: "libbtr_oracle_fmad0.so", 2: "libbtr_oracle_fmad2.so"}
_LIBS = {}
_MODE = int(os.environ.get("BTR_FMAD", "1"))


def build(force=False):
    """Compile the three libbtr_oracle*.so (one per BTR_FMAD rounding mode of the squared
    distance) with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    mk = os.path.join(_HERE, "Makefile")
    newest = max(os.path.getmtime(src), os.path.getmtime(mk))
    sos = [os.path.join(_HERE, n) for n in _NAMES.values()]
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < newest for so in sos):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])
    return os.path.join(_HERE, _NAMES[1])


def fmad():
    """The rounding mode the oracle currently restates (see pointnet2_oracle.c header)."""
    return _MODE


def set_fmad(mode):
    """Select the rounding mode (0, 1 or 2); tests switch it together with the HIP library's
    (backtoreality_amd.pointnet2._ext.set_fmad)."""
    global _MODE
    assert mode in _NAMES, mode
    _MODE = int(mode)


def lib():
    if _MODE not in _LIBS:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, _NAMES[_MODE]))
        L.btr_oracle_fmad_mode.restype = _c
        assert L.btr_oracle_fmad_mode() == _MODE
        L.btr_oracle_opt_n_threads.argtypes = [_c]
        L.btr_oracle_opt_n_threads.restype = _c
        L.btr_oracle_furthest_point_sampling_bs.argtypes = [_c, _c, _c, _f, _f, _i, _c]
        L.btr_oracle_furthest_point_sampling_closed_form.argtypes = [_c, _c, _c, _f, _f, _i, _c]
        L.btr_oracle_gather_points.argtypes = [_c, _c, _c, _c, _f, _i, _f]
        L.btr_oracle_gather_points_grad.argtypes = [_c, _c, _c, _c, _f, _i, _f]
        L.btr_oracle_ball_query.argtypes = [_c, _c, _c, ctypes.c_float, _c, _f, _f, _i]
        L.btr_oracle_group_points.argtypes = [_c, _c, _c, _c, _c, _f, _i, _f]
        L.btr_oracle_group_points_grad.argtypes = [_c, _c, _c, _c, _c, _f, _i, _f]
        L.btr_oracle_three_nn.argtypes = [_c, _c, _c, _f, _f, _f, _i]
        L.btr_oracle_three_interpolate.argtypes = [_c, _c, _c, _c, _f, _i, _f, _f]
        L.btr_oracle_three_interpolate_grad.argtypes = [_c, _c, _c, _c, _f, _i, _f, _f]
        _LIBS[_MODE] = L
    return _LIBS[_MODE]


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------------------- numpy-level API
def opt_n_threads(work_size):
    return int(lib().btr_oracle_opt_n_threads(int(work_size)))


def furthest_point_sampling(xyz, npoint, block_size=0, closed_form=False):
    """xyz (B,N,3) f32 -> idx (B,npoint) i32.  sampling.cpp:70-91 + sampling_gpu.cu:74-178."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, max(npoint, 0)), np.int32)
    temp = np.full((B, N), 1e10, np.float32)
    fn = (lib().btr_oracle_furthest_point_sampling_closed_form if closed_form
          else lib().btr_oracle_furthest_point_sampling_bs)
    if npoint > 0 and B > 0:
        fn(B, N, npoint, _fp(xyz), _fp(temp), _ip(out), int(block_size))
    return out


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.zeros((B, C, M), np.float32)
    lib().btr_oracle_gather_points(B, C, N, M, _fp(points), _ip(idx), _fp(out))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M = grad_out.shape
    out = np.zeros((B, C, n), np.float32)
    lib().btr_oracle_gather_points_grad(B, C, n, M, _fp(grad_out), _ip(idx), _fp(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """C++ argument order (new_xyz, xyz, radius, nsample) -- ball_query.h:9-10."""
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    out = np.zeros((B, M, nsample), np.int32)
    lib().btr_oracle_ball_query(B, N, M, ctypes.c_float(radius), nsample, _fp(new_xyz), _fp(xyz),
                                _ip(out))
    return out


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    B, C, N = points.shape
    _, M, S = idx.shape
    out = np.zeros((B, C, M, S), np.float32)
    lib().btr_oracle_group_points(B, C, N, M, S, _fp(points), _ip(idx), _fp(out))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, M, S = grad_out.shape
    out = np.zeros((B, C, n), np.float32)
    lib().btr_oracle_group_points_grad(B, C, n, M, S, _fp(grad_out), _ip(idx), _fp(out))
    return out


def three_nn(unknown, known):
    """-> (dist2 (B,n,3) f32 SQUARED, idx (B,n,3) i32).  interpolate.cpp:19-45."""
    unknown, known = _f32(unknown), _f32(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((B, n, 3), np.float32)
    idx = np.zeros((B, n, 3), np.int32)
    lib().btr_oracle_three_nn(B, n, m, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    B, C, m = points.shape
    n = idx.shape[1]
    out = np.zeros((B, C, n), np.float32)
    lib().btr_oracle_three_interpolate(B, C, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, m), np.float32)
    lib().btr_oracle_three_interpolate_grad(B, C, n, m, _fp(grad_out), _ip(idx), _fp(weight),
                                            _fp(out))
    return out


# ------------------------------------------------------------------ torch-level `_ext` adapter
class _ExtCPU(object):
    """The nine names of the reference's ``pointnet2._ext`` (bindings.cpp:11-24) over CPU
    torch tensors, with the wrappers' checks (utils.h:10-30) turned into RuntimeError."""

    @staticmethod
    def _chk(t, name, dtype):
        import torch
        if not t.is_contiguous():
            raise RuntimeError("%s must be a contiguous tensor" % name)
        if dtype == "float" and t.dtype != torch.float32:
            raise RuntimeError("%s must be a float tensor" % name)
        if dtype == "int" and t.dtype != torch.int32:
            raise RuntimeError("%s must be an int tensor" % name)
        return t.detach().numpy()

    def furthest_point_sampling(self, points, nsamples):
        import torch
        return torch.from_numpy(furthest_point_sampling(self._chk(points, "points", "float"),
                                                        int(nsamples)))

    def gather_points(self, points, idx):
        import torch
        return torch.from_numpy(gather_points(self._chk(points, "points", "float"),
                                              self._chk(idx, "idx", "int")))

    def gather_points_grad(self, grad_out, idx, n):
        import torch
        return torch.from_numpy(gather_points_grad(self._chk(grad_out, "grad_out", "float"),
                                                   self._chk(idx, "idx", "int"), int(n)))

    def ball_query(self, new_xyz, xyz, radius, nsample):
        import torch
        return torch.from_numpy(ball_query(self._chk(new_xyz, "new_xyz", "float"),
                                           self._chk(xyz, "xyz", "float"), float(radius),
                                           int(nsample)))

    def group_points(self, points, idx):
        import torch
        return torch.from_numpy(group_points(self._chk(points, "points", "float"),
                                             self._chk(idx, "idx", "int")))

    def group_points_grad(self, grad_out, idx, n):
        import torch
        return torch.from_numpy(group_points_grad(self._chk(grad_out, "grad_out", "float"),
                                                  self._chk(idx, "idx", "int"), int(n)))

    def three_nn(self, unknowns, knows):
        import torch
        d, i = three_nn(self._chk(unknowns, "unknowns", "float"),
                        self._chk(knows, "knows", "float"))
        return [torch.from_numpy(d), torch.from_numpy(i)]

    def three_interpolate(self, points, idx, weight):
        import torch
        return torch.from_numpy(three_interpolate(self._chk(points, "points", "float"),
                                                  self._chk(idx, "idx", "int"),
                                                  self._chk(weight, "weight", "float")))

    def three_interpolate_grad(self, grad_out, idx, weight, m):
        import torch
        return torch.from_numpy(three_interpolate_grad(self._chk(grad_out, "grad_out", "float"),
                                                       self._chk(idx, "idx", "int"),
                                                       self._chk(weight, "weight", "float"),
                                                       int(m)))


ext_cpu = _ExtCPU()
