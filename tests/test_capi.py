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


def test_plan_functions_lay_out_their_arenas_without_a_gpu():
    """btr_decoder_layer_plan / btr_backbone_plan / btr_pm_chain_plan are host-only: offsets are
    aligned and disjoint, the flat-gradient size is the parameter count, bad descriptions are
    reported through the status code."""
    from backtoreality_amd.pointnet2 import _ext
    lib = _ext._lib
    d, plan = _ext.DecoderLayer(), _ext.DecoderPlan()
    d.b, d.pq, d.pk, d.e, d.heads, d.ff, d.dropout = 4, 256, 1024, 288, 8, 2048, 0.1
    assert lib.btr_decoder_layer_plan(ctypes.addressof(d), ctypes.addressof(plan)) == 0
    assert (plan.rq, plan.rk) == (1024, 4096)
    E, F = 288, 2048
    assert plan.grads_floats == 2 * (3 * E * E + 3 * E + E * E + E) + 2 * E * F + F + E + 6 * E
    offs = [getattr(plan, n) for n in ("qp0", "qkv", "a1", "lse1", "xh1", "rs1", "x1", "qp1", "q2",
                                       "kp", "kv", "a2", "lse2", "xh2", "rs2", "x2", "h", "xh3",
                                       "rs3")]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs) and offs[-1] < plan.saved_bytes
    assert plan.fwd_scratch_bytes > 0 and plan.bwd_scratch_bytes > plan.saved_bytes // 4
    for field, bad in (("heads", 7), ("e", 290), ("dropout", 1.0), ("ff", 0)):
        d2 = _ext.DecoderLayer.from_buffer_copy(d)
        setattr(d2, field, bad)
        assert lib.btr_decoder_layer_plan(ctypes.addressof(d2), ctypes.addressof(plan)) == -1
        assert b"decoder_layer_plan" in lib.btr_last_error()
    assert lib.btr_decoder_layer_plan(None, ctypes.addressof(plan)) == -1
    assert lib.btr_gf_loss_part_floats(4, 256, 7) == 7 * 4 * 4 * 7
    # the entry points refuse null operands before touching the device
    assert lib.btr_decoder_layer_forward(ctypes.addressof(d), ctypes.addressof(plan), None, None,
                                         None, None, None, None, None, None, None) == -1
    assert lib.btr_gf_loss_fwd(*([None] * 21)) == -1


def test_build_guard_finds_packed_f32_in_any_object(tmp_path):
    """build.check_objects (round-3 review, What's weak #3; since round 6 every object, not only
    the index-producing ones): the device code of every object is disassembled at build time and
    packed f32 arithmetic fails the build.  The built objects pass; a probe kernel that multiplies
    float2 vectors is caught."""
    import subprocess
    from backtoreality_amd import build
    build.build()
    build.check_objects(force=True)      # every object, all three modes of the index sources
    src = tmp_path / "pk_probe.hip"
    src.write_text(
        '#include <hip/hip_runtime.h>\n'
        'typedef float v2f __attribute__((ext_vector_type(2)));\n'
        '__global__ void pk_probe(const v2f *a, const v2f *b, v2f *o) {\n'
        '  const int i = blockIdx.x * 64 + threadIdx.x;\n'
        '  o[i] = a[i] * b[i] + a[i];\n'
        '}\n')
    obj = tmp_path / "pk_probe.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", str(src),
                           "-o", str(obj)])
    sites = build.packed_f32_sites(str(obj))
    assert sites and all("pk_probe" in fn for fn, _ in sites), sites


def _runtime_defaults(env):
    """(btr_grid_cus, btr_fps_lds_reserve_kb) of a fresh process under `env` (both are read once
    per process)."""
    import subprocess
    import sys
    code = ("import ctypes; from backtoreality_amd import build; l = ctypes.CDLL(build.build()); "
            "print(l.btr_grid_cus(), l.btr_fps_lds_reserve_kb())")
    e = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "BTR_DP", "BTR_GRID_CUS", "BTR_FPS_LDS_KB", "BTR_COMM_CUS")}
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, check=True,
                         capture_output=True, text=True).stdout.split()
    return int(out[-2]), int(out[-1])


def test_grid_and_lds_defaults_follow_the_world_size():
    """One-round grids and the FPS's LDS reservation leave room for RCCL's kernels in a
    data-parallel run (train_GF_FSB.py:450-474 launches one process per GPU): no GPU needed."""
    import torch
    cus = 256   # (no device here; on a GPU box the device's count)
    if torch.cuda.is_available():
        cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert _runtime_defaults({}) == (cus - 8, 128)
    assert _runtime_defaults({"WORLD_SIZE": "1"}) == (cus - 8, 128)
    # the flat all-reduce runs behind the backward: only the LDS reservation moves
    assert _runtime_defaults({"WORLD_SIZE": "8"}) == (cus - 8, 96)
    # DistributedDataParallel overlaps its buckets with the backward: CUs set aside for them
    assert _runtime_defaults({"WORLD_SIZE": "8", "BTR_DP": "ddp"}) == (cus - 24, 96)
    assert _runtime_defaults({"WORLD_SIZE": "8", "BTR_DP": "ddp", "BTR_COMM_CUS": "32"}) == (
        cus - 40, 96)
    # explicit settings win
    assert _runtime_defaults({"WORLD_SIZE": "8", "BTR_GRID_CUS": "200", "BTR_FPS_LDS_KB": "0"}) == (
        200, 0)
