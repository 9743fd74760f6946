"""btr_sa_bwd_fused (csrc/sa_mlp.hip sa_bwd_fused_kernel): a hidden layer's whole backward as one
pass -- dY formed while staging (pooled-layer gradient or BatchNorm backward), weight gradient,
input gradient and the next BatchNorm's backward sums -- against a float64 evaluation of the
autograd backward it replaces (SharedMLP's Conv2d + BatchNorm2d + ReLU stack,
pointnet2/pytorch_utils.py:11-36, 157-188), at ragged shapes, with the first-layer recompute,
and bit-reproducible."""
import pytest
import torch

from backtoreality_amd.pointnet2 import _ext

pytestmark = pytest.mark.gpu
_lib, _p = _ext._lib, _ext._p


def _run(dev, rows, n, k, pooled, rc, s=16, seed=0, ldx_pad=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    G = rnd(rows, n)
    Yl = rnd(rows, n)
    sc, sh, mu, isd = rnd(n), rnd(n) * 0.3, rnd(n) * 0.2, torch.rand(n, generator=g).to(dev) + 0.5
    m1l, m2l = rnd(n) * 0.05, rnd(n) * 0.05
    if rc:
        x = rnd(rows, 4)
        w0 = rnd(k, 4)
        ldx = 4
        yprev = x.double() @ w0.double().t()
    else:
        ldx = k + ldx_pad
        x = rnd(rows, ldx)
        w0 = None
        yprev = x[:, :k].double()
    pa, pb = rnd(k), rnd(k) * 0.3
    mu_p, is_p = rnd(k) * 0.2, torch.rand(k, generator=g).to(dev) + 0.5
    Wt = rnd(k, n) * 0.2
    groups = rows // s
    if pooled:
        arg = torch.randint(0, s, (groups, n), generator=g, dtype=torch.uint8).to(dev)
        dcl = rnd(groups, n)
        alpha, beta = rnd(n) * 0.1, rnd(n) * 0.1
        rr = torch.arange(rows, device=dev)
        hit = (rr % s).unsqueeze(1) == arg.long()[rr // s]
        dY = alpha.double() * G.double() + beta.double() + torch.where(
            hit, dcl.double()[rr // s], torch.zeros((), dtype=torch.float64, device=dev))
    else:
        arg = dcl = alpha = beta = None
        m = (sc.double() * Yl.double() + sh.double()) > 0
        xhat = (Yl.double() - mu.double()) * isd.double()
        dY = sc.double() * (torch.where(m, G.double(), torch.zeros_like(xhat)) -
                            (m1l.double() + xhat * m2l.double()))
    X = torch.relu(pa.double() * yprev + pb.double())
    dz_ref = dY @ Wt.double().t()
    dw_ref = dY.t() @ X
    mp = (pa.double() * yprev + pb.double()) > 0
    gm = torch.where(mp, dz_ref, torch.zeros_like(dz_ref))
    s1_ref = gm.sum(0)
    s2_ref = (gm * (yprev - mu_p.double()) * is_p.double()).sum(0)

    chunks = _lib.btr_sa_bwd_fused_chunks(rows, n, k)
    f32 = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
    dz, pw, dw = f32(rows, k), f32(chunks, n, k), f32(n, k)
    spart = f32(chunks, 2, k)
    m1, m2, dg, db = f32(k), f32(k), f32(k), f32(k)
    assert _lib.btr_sa_bwd_fused_supported(rows, n, k)
    with _ext._on(G) as d:
        _ext._call(_lib.btr_sa_bwd_fused, rows, n, k, _p(G), n, _p(None if pooled else Yl),
                   _p(sc), _p(sh), _p(mu), _p(isd), _p(m1l), _p(m2l), s, _p(arg), _p(dcl),
                   _p(alpha), _p(beta), _p(x), ldx, _p(w0), _p(pa), _p(pb), _p(mu_p), _p(is_p),
                   _p(Wt), n, _p(dz), k, _p(pw), _p(dw), _p(spart), _p(m1), _p(m2), _p(dg),
                   _p(db), _ext._stream(d))
    torch.cuda.synchronize()
    return (dz, dw, dg, db, m1, m2), (dz_ref, dw_ref, s2_ref, s1_ref, s1_ref / rows, s2_ref / rows)


def _check(got, ref):
    names = ("dz", "dw", "dgamma", "dbeta", "m1", "m2")
    for name, a, b in zip(names, got, ref):
        assert torch.isfinite(a).all(), name
        scale = float(b.abs().max())
        err = float((a.double() - b).abs().max())
        # sums over up to 20 000 rows of f32 products: a few f32 roundings of the largest entry
        assert err <= 3e-5 * scale, (name, err / scale)


@pytest.mark.parametrize("rows,n,k,pooled,rc,s", [
    (4096, 128, 64, True, False, 64),      # SA1's pooled layer
    (4096, 64, 64, False, True, 16),       # SA1's hidden layer over the recomputed first layer
    (2048, 128, 128, False, False, 16),    # SA2's hidden layer (two k blocks)
    (1000, 64, 64, False, False, 16),      # ragged row count (last step of 8 rows)
    (1040, 128, 64, True, False, 16),      # ragged, pooled, groups of 16
    (736, 100, 36, False, False, 16),      # n, k not multiples of 32
    (20000, 128, 64, True, True, 32),      # many chunks, pooled over a recomputed layer
    (3000, 48, 132, False, False, 16),     # three k blocks, the last one of 4 columns
    (4096, 256, 128, True, False, 32),     # SA2's pooled layer: the 256-wide variant
    (3008, 256, 128, False, False, 16),    # 256-wide, BatchNorm source, ragged rows
    (1024, 200, 260, False, False, 16),    # 256-wide variant on n = 200, five k blocks
    (8192, 256, 256, False, False, 16),    # a feature-propagation chain layer
])
def test_fused_backward_matches_float64(cuda, rows, n, k, pooled, rc, s):
    got, ref = _run(cuda, rows, n, k, pooled, rc, s=s, ldx_pad=4 if (k % 8 and not rc) else 0)
    _check(got, ref)


def test_fused_backward_is_bit_reproducible(cuda):
    a, _ = _run(cuda, 8960, 128, 64, True, False, s=64, seed=3)
    b, _ = _run(cuda, 8960, 128, 64, True, False, s=64, seed=3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_unsupported_shapes_are_refused(cuda):
    assert not _lib.btr_sa_bwd_fused_supported(1024, 260, 128)    # n > 256
    assert not _lib.btr_sa_bwd_fused_supported(1024, 128, 130)    # k not a multiple of 4
    assert not _lib.btr_sa_bwd_fused_supported(0, 128, 64)


# ---------------------------------------------------------------- the pooled layer in Gram form
def _run_gram(dev, rows, n, k, s, seed=0, ldx_pad=0):
    """btr_sa_bwd_gram against the float64 evaluation of what it replaces: the pooled layer's
    dY = alpha * Y_l + beta + sparse with Y_l = X W_l^T (the affine structure the Gram form rests
    on), then dW_l, dZ_{l-1} and BatchNorm_{l-1}'s backward sums."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    ldx = k + ldx_pad
    x = rnd(rows, ldx)
    yprev = x[:, :k].double()
    pa, pb = rnd(k), rnd(k) * 0.3
    mu_p, is_p = rnd(k) * 0.2, torch.rand(k, generator=g).to(dev) + 0.5
    W = rnd(n, k) * 0.2
    Wt = W.t().contiguous()
    groups = rows // s
    arg = torch.randint(0, s, (groups, n), generator=g, dtype=torch.uint8).to(dev)
    dcl = rnd(groups, n)
    # alpha * y + beta with a mean that does not vanish: the dense part cancels in dW
    alpha, beta = rnd(n) * 0.1, rnd(n) * 0.1
    X = torch.relu(pa.double() * yprev + pb.double())
    Y = X @ W.double().t()
    rr = torch.arange(rows, device=dev)
    hit = (rr % s).unsqueeze(1) == arg.long()[rr // s]
    dY = alpha.double() * Y + beta.double() + torch.where(
        hit, dcl.double()[rr // s], torch.zeros((), dtype=torch.float64, device=dev))
    dz_ref = dY @ W.double()
    dw_ref = dY.t() @ X
    mp = (pa.double() * yprev + pb.double()) > 0
    gm = torch.where(mp, dz_ref, torch.zeros_like(dz_ref))
    s1_ref = gm.sum(0)
    s2_ref = (gm * (yprev - mu_p.double()) * is_p.double()).sum(0)

    assert _lib.btr_sa_bwd_gram_supported(rows, n, k)
    chunks = _lib.btr_sa_bwd_gram_chunks(rows, n, k)
    f32 = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
    dz, pw, dw = f32(rows, k), f32(chunks, n, k), f32(n, k)
    spart = f32(chunks, 2, k)
    gs = f32(int(_lib.btr_sa_bwd_gram_scratch_floats(rows, n, k)))
    m1, m2, dg, db = f32(k), f32(k), f32(k), f32(k)
    with _ext._on(x) as d:
        _ext._call(_lib.btr_sa_bwd_gram, rows, n, k, _p(x), ldx, _p(pa), _p(pb), _p(mu_p),
                   _p(is_p), _p(W), _p(Wt), n, s, _p(arg), _p(dcl), _p(alpha), _p(beta), _p(dz),
                   k, _p(pw), _p(dw), _p(gs), _p(spart), _p(m1), _p(m2), _p(dg), _p(db),
                   _ext._stream(d))
    torch.cuda.synchronize()
    return (dz, dw, dg, db, m1, m2), (dz_ref, dw_ref, s2_ref, s1_ref, s1_ref / rows, s2_ref / rows)


@pytest.mark.parametrize("rows,n,k,s", [
    (4096, 128, 64, 64),       # SA1's pooled layer
    (1040, 128, 64, 16),       # ragged rows (last step of 16)
    (1024, 100, 36, 16),       # n, k not multiples of 32
    (20000, 128, 64, 32),      # many chunks
    (2048, 64, 64, 16),        # the 64-column variant
    (4096, 128, 32, 16),       # half a k block
])
def test_gram_backward_matches_float64(cuda, rows, n, k, s):
    got, ref = _run_gram(cuda, rows, n, k, s, ldx_pad=4 if k % 8 else 0)
    _check(got, ref)


def test_gram_backward_is_bit_reproducible(cuda):
    a, _ = _run_gram(cuda, 8960, 128, 64, 64, seed=3)
    b, _ = _run_gram(cuda, 8960, 128, 64, 64, seed=3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_gram_backward_covers_the_producer_consumer_shapes_only(cuda):
    """n <= 128, k <= 64 (SA1's pooled layer); wider layers keep the Y_l-reading form, and so does
    a layer with more rows than the kernel's block -> group table can index per chunk (the plan
    asks btr_sa_bwd_gram_supported, so the forward still stores Y_l there)."""
    assert _lib.btr_sa_bwd_gram_supported(4096, 128, 64)
    assert not _lib.btr_sa_bwd_gram_supported(4096, 256, 128)
    assert not _lib.btr_sa_bwd_gram_supported(4096, 128, 128)
    assert _lib.btr_sa_bwd_gram_supported(8 * 2048 * 64, 128, 64)          # the benchmark's SA1
    assert not _lib.btr_sa_bwd_gram_supported(2000 * 2048 * 64, 128, 64)   # beyond the table
