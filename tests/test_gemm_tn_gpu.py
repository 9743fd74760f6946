"""btr_sa_gemm_tn (weight gradient dW[n][k] = sum_r g[r][n] f(x[r][k]), split over row chunks and
reduced in a fixed order) against a float64 evaluation, at ragged shapes: rows that do not fill
the last 32-row step or the last chunk, n / k that do not fill the 128 / 64-wide tiles, with
and without the BatchNorm + ReLU prologue on x.  The default kernel is the bf16x6 form whose
operand fragments come through gfx950's LDS transpose read (csrc/sa_mlp.hip gemm_tn_x6_kernel);
BTR_GEMM_TN=f32 selects the f32-input MFMA kernel -- both must sit at f32 rounding distance."""
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
