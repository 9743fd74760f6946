"""gemm_nt_sm_kernel (csrc/sa_mlp.hip): the small-M NT GEMM of the point-wise chains -- 32 x 64
tiles, the weights as pre-split bf16 planes (btr_pm_weight_planes), k in at most two staged
chunks -- against a float64 evaluation of the 1x1 convolution (+ BatchNorm + ReLU prologue, bias)
it computes (reference: Conv1d / Conv2d with kernel size 1 in SharedMLP, pytorch_utils.py:11-36,
voting_module.py:38-65, proposal_module.py:84-120), its BatchNorm statistics partials, and against
gemm_nt_kernel on the same operands."""
import pytest
import torch

from backtoreality_amd.pointnet2 import _ext

pytestmark = pytest.mark.gpu
_lib, _p = _ext._lib, _ext._p


@pytest.fixture(autouse=True)
def _up_to_4096_rows(monkeypatch):
    # (the library's default takes the kernel up to 2 048 rows -- where it pays inside a training
    # step; the kernel itself is exercised up to 4 096 here)
    monkeypatch.setenv("BTR_PM_SM_ROWS", "4096")


def _run(dev, rows, n, k, pro, mode, seed=0, lda_pad=0):
    g = torch.Generator(device="cpu").manual_seed(seed + rows + n + k)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    lda = k + lda_pad
    A, W = rnd(rows, lda), (rnd(n, k) * 0.3).contiguous()
    pa, pb = (rnd(k), rnd(k) * 0.3) if pro else (None, None)
    bias = rnd(n) if mode == "bias" else None
    grid = _lib.btr_pm_gemm_grid(rows)
    part = torch.full((grid, 2, n), float("nan"), device=dev) if mode == "stats" else None
    C = torch.full((rows, n), float("nan"), device=dev)
    planes = torch.empty((int(_lib.btr_pm_weight_planes_bytes(n, k)),), dtype=torch.uint8, device=dev)
    assert _lib.btr_pm_gemm_nt_sm_supported(rows, n, k)
    with _ext._on(A) as d:
        st = _ext._stream(d)
        _ext._call(_lib.btr_pm_weight_planes, n, k, _p(W), k, _p(planes), st)
        _ext._call(_lib.btr_pm_gemm_nt_sm, rows, n, k, _p(A), lda, _p(planes), _p(C), n, _p(pa),
                   _p(pb), _p(part), _p(bias), st)
        C_old = torch.full((rows, n), float("nan"), device=dev)
        part_old = torch.full((grid, 2, n), float("nan"), device=dev) if mode == "stats" else None
        _ext._call(_lib.btr_pm_gemm_nt, rows, n, k, _p(A), lda, _p(W), k, _p(C_old), n, _p(pa),
                   _p(pb), _p(part_old), _p(bias), st)
    torch.cuda.synchronize()
    X = A[:, :k].double()
    if pro:
        X = torch.relu(pa.double() * X + pb.double())
    ref = X @ W.double().t()
    if bias is not None:
        ref = ref + bias.double()
    return C, part, C_old, part_old, ref, grid


@pytest.mark.parametrize("rows,n,k,pro,mode", [
    (4096, 256, 256, 1, "stats"),     # the first feature-propagation module's hidden layer
    (4096, 256, 512, 0, "stats"),     # ... its first layer: two k chunks
    (1024, 288, 288, 1, "stats"),     # GroupFree3D's decoder width
    (2048, 128, 128, 1, "stats"),     # the proposal head
    (2048, 80, 128, 1, "bias"),       # ... its bare last layer (79 -> 80 columns)
    (4096, 260, 256, 1, "bias"),      # the vote generator's last layer (at half the rows)
    (1000, 100, 36, 0, "plain"),      # ragged everything
    (4000, 64, 512, 1, "plain"),      # the largest k
    (4096, 256, 288, 0, "plain"),     # an input gradient
    (33, 32, 16, 0, "stats"),         # two row tiles, one of one row
])
def test_small_m_gemm_matches_float64(cuda, rows, n, k, pro, mode):
    C, part, C_old, part_old, ref, grid = _run(cuda, rows, n, k, pro, mode,
                                                 lda_pad=4 if k % 8 else 0)
    scale = float(ref.abs().max())
    assert torch.isfinite(C).all()
    assert float((C.double() - ref).abs().max()) <= 2e-6 * scale
    # ... and the 64-row-tile kernel on the same operands: the same products, another order
    assert float((C - C_old).abs().max()) <= 2e-6 * scale
    if mode == "stats":
        s1, s2 = part.double()[:, 0].sum(0), part.double()[:, 1].sum(0)
        r1, r2 = ref.sum(0), (ref * ref).sum(0)
        assert float((s1 - r1).abs().max()) <= 3e-5 * float(ref.abs().sum(0).max())
        assert float((s2 - r2).abs().max()) <= 1e-5 * float(r2.abs().max())
        o1, o2 = part_old.double()[:, 0].sum(0), part_old.double()[:, 1].sum(0)
        assert float((s1 - o1).abs().max()) <= 3e-5 * float(ref.abs().sum(0).max())
        assert float((s2 - o2).abs().max()) <= 1e-5 * float(r2.abs().max())


def test_small_m_gemm_is_bit_reproducible_and_refuses_large_shapes(cuda):
    a = _run(cuda, 4096, 256, 256, 1, "stats", seed=1)
    b = _run(cuda, 4096, 256, 256, 1, "stats", seed=1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert not _lib.btr_pm_gemm_nt_sm_supported(4097, 256, 256)     # rows
    assert not _lib.btr_pm_gemm_nt_sm_supported(4096, 256, 516)     # k > 512
    assert not _lib.btr_pm_gemm_nt_sm_supported(4096, 256, 130)     # k not a multiple of 4


@pytest.mark.gpu
def test_gemm_trace_brackets_the_family(cuda):
    """btr_gemm_trace_begin / _end: event pairs around the family's entry points, on whatever
    host path calls them (here: two direct calls; bench.py: the whole-layer calls)."""
    import ctypes
    lib = _ext._lib
    a = torch.randn(4096, 128, device=cuda)
    w = torch.randn(128, 128, device=cuda)
    c = torch.empty(4096, 128, device=cuda)
    st = _ext._stream(0)
    lib.btr_gemm_trace_begin()
    for _ in range(2):
        _ext._call(lib.btr_pm_gemm_nt, 4096, 128, 128, _ext._p(a), 128, _ext._p(w), 128, _ext._p(c),
                   128, None, None, None, None, st)
    ms, pairs = ctypes.c_double(0.0), ctypes.c_int(0)
    assert lib.btr_gemm_trace_end(ctypes.addressof(ms), ctypes.addressof(pairs)) == 0
    assert pairs.value == 2 and 0.0 < ms.value < 5.0
    # closed: nothing is recorded any more
    _ext._call(lib.btr_pm_gemm_nt, 4096, 128, 128, _ext._p(a), 128, _ext._p(w), 128, _ext._p(c),
               128, None, None, None, None, st)
    lib.btr_gemm_trace_begin()
    assert lib.btr_gemm_trace_end(ctypes.addressof(ms), ctypes.addressof(pairs)) == 0
    assert pairs.value == 0
    torch.testing.assert_close(c, a @ w.t(), rtol=1e-4, atol=1e-3)
