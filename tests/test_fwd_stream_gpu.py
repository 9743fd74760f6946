"""sa_fwd_stream_kernel (csrc/sa_mlp.hip): the streaming form of the narrow layers' NT GEMMs --
taken by btr_sa_gemm_nt / _rc / _poolfwd for n <= 256, k <= 128 and >= 16 384 rows -- against a
float64 evaluation: the product (with the BatchNorm + ReLU / first-layer-recompute prologues),
the BatchNorm statistics partials, and the pooling extrema of 8-row blocks / 16-row groups with the arg-max row
(first maximum; the minimum where gamma < 0)."""
import pytest
import torch

from backtoreality_amd.pointnet2 import _ext

pytestmark = pytest.mark.gpu
_lib, _p = _ext._lib, _ext._p


def _sums(part, nblk, n):
    p = part[:nblk].double()
    return p[:, 0, :n].sum(0), p[:, 1, :n].sum(0)


@pytest.mark.parametrize("rows,n,k,pro,stats", [
    (20000, 128, 64, 1, True),     # a hidden / last layer of SA1's width
    (16384, 64, 64, 1, True),
    (33000, 128, 128, 1, True),    # SA2-SA4 hidden layers, ragged rows
    (20000, 128, 128, 0, False),   # an input gradient
    (17000, 64, 64, 0, False),
    (16390, 100, 36, 1, True),     # n, k not multiples of 32
    (20000, 128, 64, 0, False),
    (18000, 256, 128, 1, True),    # two 128-column slabs per row chunk
    (16400, 200, 64, 0, False),    # ... the second one not full
])
def test_stream_nt_matches_float64(cuda, rows, n, k, pro, stats):
    g = torch.Generator(device="cpu").manual_seed(rows + n)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    A, W = rnd(rows, k), rnd(n, k) * 0.3
    pa, pb = (rnd(k), rnd(k) * 0.3) if pro else (None, None)
    C = torch.full((rows, n), float("nan"), device=cuda)
    nblk = _lib.btr_sa_gemm_grid(rows)
    part = torch.full((nblk, 2, n), float("nan"), device=cuda) if stats else None
    with _ext._on(A) as d:
        _ext._call(_lib.btr_sa_gemm_nt, rows, n, k, _p(A), k, _p(W), k, _p(C), n, _p(pa), _p(pb),
                   _p(part), _ext._stream(d))
    X = A.double()
    if pro:
        X = torch.relu(pa.double() * X + pb.double())
    ref = X @ W.double().t()
    scale = float(ref.abs().max())
    assert torch.isfinite(C).all()
    assert float((C.double() - ref).abs().max()) <= 2e-6 * scale
    if stats:
        s1, s2 = _sums(part, nblk, n)
        r1, r2 = ref.sum(0), (ref * ref).sum(0)
        assert float((s1 - r1).abs().max()) <= 3e-5 * float(r1.abs().max() + (ref.abs()).sum(0).max())
        assert float((s2 - r2).abs().max()) <= 1e-5 * float(r2.abs().max())


def test_stream_first_layer_recompute(cuda):
    rows, n, k = 24000, 64, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    x0, w0, W = rnd(rows, 4), rnd(k, 4), rnd(n, k) * 0.3
    pa, pb = rnd(k), rnd(k) * 0.3
    C = torch.full((rows, n), float("nan"), device=cuda)
    nblk = _lib.btr_sa_gemm_grid(rows)
    part = torch.full((nblk, 2, n), float("nan"), device=cuda)
    with _ext._on(x0) as d:
        _ext._call(_lib.btr_sa_gemm_nt_rc, rows, n, k, _p(x0), _p(w0), _p(W), k, _p(C), n, _p(pa),
                   _p(pb), _p(part), _ext._stream(d))
    y0 = x0.double() @ w0.double().t()
    ref = torch.relu(pa.double() * y0 + pb.double()) @ W.double().t()
    assert float((C.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    s1, s2 = _sums(part, nblk, n)
    assert float((s2 - (ref * ref).sum(0)).abs().max()) <= 1e-5 * float((ref * ref).sum(0).max())


@pytest.mark.parametrize("rows,n,k,ps", [
    (20000, 128, 64, 8), (16392, 128, 128, 8), (24000, 100, 64, 8),
    (18008, 256, 128, 8),     # SA2's pooled layer: 256 wide
    (32768, 128, 128, 16),    # the vote aggregation's pooled layer: groups of 16 dense rows
    (65536, 256, 128, 16),    # SA3's
    (16400, 200, 100, 16),
])
def test_stream_pooling_epilogue_of_8_row_blocks(cuda, rows, n, k, ps):
    g = torch.Generator(device="cpu").manual_seed(n + k)
    rnd = lambda *s: torch.randn(*s, generator=g).to(cuda)
    A, W = rnd(rows, k), rnd(n, k) * 0.3
    pa, pb, gamma = rnd(k), rnd(k) * 0.3, rnd(n)
    C = torch.full((rows, n), float("nan"), device=cuda)
    nblk = _lib.btr_sa_gemm_grid(rows)
    part = torch.full((nblk, 2, n), float("nan"), device=cuda)
    groups = rows // ps
    gext = torch.full((groups, n), float("nan"), device=cuda)
    aext = torch.full((groups, n), 255, dtype=torch.uint8, device=cuda)
    assert _lib.btr_sa_gemm_nt_poolfwd_supported(rows, n, ps)
    with _ext._on(A) as d:
        _ext._call(_lib.btr_sa_gemm_nt_poolfwd, rows, n, k, _p(A), k, _p(W), k, _p(C), n, _p(pa),
                   _p(pb), _p(part), ps, _p(gamma), _p(gext), _p(aext), _ext._stream(d))
    ref = torch.relu(pa.double() * A.double() + pb.double()) @ W.double().t()
    assert float((C.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # the extrema of the kernel's OWN C (the selection is exact on the values it stored)
    blocks = C.view(groups, ps, n)
    sign = torch.where(gamma < 0, -1.0, 1.0)
    best, arg = (blocks * sign).max(1)
    assert torch.equal(gext, best * sign)
    first = ((blocks * sign) == best.unsqueeze(1)).float().argmax(1)   # first maximum
    assert torch.equal(aext.long(), first)
    s1, s2 = _sums(part, nblk, n)
    r2 = (ref * ref).sum(0)
    assert float((s2 - r2).abs().max()) <= 1e-5 * float(r2.abs().max())
