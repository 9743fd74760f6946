"""btr_sa_gemm_tn (weight gradient dW[n][k] = sum_r g[r][n] f(x[r][k]), split over row chunks and
reduced in a fixed order) against a float64 evaluation, at ragged shapes: rows that do not fill
the last 32-row step or the last chunk, n / k that do not fill the 128 / 64-wide tiles, with
and without the BatchNorm + ReLU prologue on x.  The default kernel is the bf16x6 form whose
operand fragments come through gfx950's LDS transpose read (csrc/sa_mlp.hip gemm_tn_x6_kernel);
BTR_GEMM=f32 selects the f32-input MFMA kernels -- both must sit at f32 rounding distance."""
import pytest
import torch

from backtoreality_amd.pointnet2 import _ext

pytestmark = pytest.mark.gpu

SHAPES = [(37, 64, 8), (1000, 132, 68), (4099, 128, 132), (2048, 288, 288), (1024, 288, 2048),
          (5000, 576, 100), (65536, 128, 260), (300, 4, 4)]


@pytest.mark.parametrize("rows,n,k", SHAPES)
@pytest.mark.parametrize("pro", [False, True])
def test_weight_gradient_matches_float64(cuda, rows, n, k, pro):
    lib = _ext._lib
    g = torch.Generator(device="cpu").manual_seed(rows + n + k)
    gy = torch.randn(rows, n, generator=g).to(cuda)
    x = torch.randn(rows, k, generator=g).to(cuda)
    pa = (torch.rand(k, generator=g) + 0.5).to(cuda) if pro else None
    pb = (torch.rand(k, generator=g) - 0.5).to(cuda) if pro else None
    chunks = lib.btr_sa_gemm_tn_chunks(rows, n, k)
    pw = torch.full((chunks, n, k), float("nan"), device=cuda)
    dw = torch.full((n, k), float("nan"), device=cuda)
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.btr_sa_gemm_tn(rows, n, k, p(gy), n, p(x), k, p(pa), p(pb), p(pw), p(dw), st)
    assert rc == 0, lib.btr_last_error()
    xe = torch.relu(x.double() * pa.double() + pb.double()) if pro else x.double()
    ref = gy.double().t() @ xe
    assert torch.isfinite(dw).all()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    # deterministic: the same call again gives the same bits (fixed-order reduction)
    dw2 = torch.empty_like(dw)
    rc = lib.btr_sa_gemm_tn(rows, n, k, p(gy), n, p(x), k, p(pa), p(pb), p(pw), p(dw2), st)
    assert rc == 0 and torch.equal(dw, dw2)


@pytest.mark.parametrize("groups,s,n,k", [(700, 16, 128, 64), (333, 32, 132, 68), (129, 64, 256, 128),
                                          (50, 128, 64, 36), (2000, 16, 64, 64)])
@pytest.mark.parametrize("pro", [False, True])
def test_pooled_gradient_form_matches_float64(cuda, groups, s, n, k, pro):
    """btr_sa_gemm_tn_pool: the gradient operand is formed while staging,
    dY[r][c] = alpha[c] * y[r][c] + beta[c]  (+ dcl[g][c] on the group's arg-max row)."""
    lib = _ext._lib
    rows = groups * s
    g = torch.Generator(device="cpu").manual_seed(groups + n)
    y = torch.randn(rows, n, generator=g).to(cuda)
    x = torch.randn(rows, k, generator=g).to(cuda)
    arg = torch.randint(0, s, (groups, n), generator=g, dtype=torch.uint8).to(cuda)
    dcl = torch.randn(groups, n, generator=g).to(cuda) * 5
    alpha = (torch.randn(n, generator=g) * 0.1).to(cuda)
    beta = (torch.randn(n, generator=g) * 0.1).to(cuda)
    pa = (torch.rand(k, generator=g) + 0.5).to(cuda) if pro else None
    pb = (torch.rand(k, generator=g) - 0.5).to(cuda) if pro else None
    chunks = lib.btr_sa_gemm_tn_chunks(rows, n, k)
    pw = torch.full((chunks, n, k), float("nan"), device=cuda)
    dw = torch.full((n, k), float("nan"), device=cuda)
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.btr_sa_gemm_tn_pool(rows, n, k, p(y), n, s, p(arg), p(dcl), p(alpha), p(beta), p(x), k,
                                 p(pa), p(pb), p(pw), p(dw), st)
    assert rc == 0, lib.btr_last_error()
    dy = alpha.double() * y.double() + beta.double()
    dy = dy.view(groups, s, n).scatter_add(1, arg.long().unsqueeze(1), dcl.double().unsqueeze(1))
    xe = torch.relu(x.double() * pa.double() + pb.double()) if pro else x.double()
    ref = dy.view(rows, n).t() @ xe
    assert torch.isfinite(dw).all()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err


@pytest.mark.parametrize("rows,n,k", [(5000, 64, 64), (70001, 128, 64), (999, 64, 32)])
def test_first_layer_recompute_form_matches_float64(cuda, rows, n, k):
    """btr_sa_gemm_tn_rc: X = relu(pa * (x0 . w0^T) + pb) rebuilt from the 4-column input rows."""
    lib = _ext._lib
    g = torch.Generator(device="cpu").manual_seed(rows)
    gy = torch.randn(rows, n, generator=g).to(cuda)
    x0 = torch.randn(rows, 4, generator=g).to(cuda)
    w0 = torch.randn(k, 4, generator=g).to(cuda)
    pa = (torch.rand(k, generator=g) + 0.5).to(cuda)
    pb = (torch.rand(k, generator=g) - 0.5).to(cuda)
    chunks = lib.btr_sa_gemm_tn_chunks(rows, n, k)
    pw = torch.full((chunks, n, k), float("nan"), device=cuda)
    dw = torch.full((n, k), float("nan"), device=cuda)
    p = _ext._p
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.btr_sa_gemm_tn_rc(rows, n, k, p(gy), n, p(x0), p(w0), p(pa), p(pb), p(pw), p(dw), st)
    assert rc == 0, lib.btr_last_error()
    # (the kernel evaluates x0 . w0^T in f32 with its own fma order: compare against the same y0)
    y0 = (x0.double() @ w0.double().t())
    ref = gy.double().t() @ torch.relu(y0 * pa.double() + pb.double())
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
