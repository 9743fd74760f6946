"""Fused point-wise MLP chains (backtoreality_amd/pointnet2/fused_mlp.py) against the stock
torch ops they replace, on the GPU: same module, weights, inputs; outputs, every gradient and
the BatchNorm running statistics."""
import copy

import pytest
import torch

from backtoreality_amd.pointnet2 import _ext
from backtoreality_amd.pointnet2 import pointnet2_modules as M
from backtoreality_amd.votenet import config, proposal_module, voting_module

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _compare(run, mod, monkeypatch, zero_ok=()):
    monkeypatch.setenv("BTR_CHAIN_MIN_ROWS", "0")   # (small test shapes: below the default gate)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("BTR_FUSED_MLP", flag)
        m = copy.deepcopy(mod)
        outs, ins = run(m)
        loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=o.device).view_as(o)).sum()
                   for o in outs)
        loss.backward()
        res[flag] = {"out%d" % i: o.detach() for i, o in enumerate(outs)}
        res[flag].update({"din%d" % i: t.grad for i, t in enumerate(ins)})
        res[flag].update({"d" + n: p.grad for n, p in m.named_parameters()})
        res[flag].update({n: b.detach().clone().float() for n, b in m.named_buffers()})
    for k, want in res["0"].items():
        got = res["1"][k]
        if any(z in k for z in zero_ok):   # bias in front of a BatchNorm: true gradient 0
            assert float(got.abs().max()) <= 1e-5 * (1 + float(want.abs().max())), k
            continue
        tol = 1e-4 if k.startswith("out") or "running" in k or "tracked" in k else 5e-4
        assert got.shape == want.shape, k
        assert _rel(got, want) < tol, (k, _rel(got, want))


def test_fp_module_mlp(cuda, monkeypatch):
    torch.manual_seed(0)
    fp = M.PointnetFPModule(mlp=[256 + 256, 256, 256]).to(cuda)
    with torch.no_grad():
        for layer in fp.mlp:
            layer.bn.bn.weight.uniform_(0.5, 1.5)
            layer.bn.bn.bias.uniform_(-0.3, 0.3)
    unknown = torch.rand(2, 1024, 3, device=cuda)
    known = unknown[:, :512].contiguous()
    uf = torch.randn(2, 256, 1024, device=cuda)
    kf = torch.randn(2, 256, 512, device=cuda)

    def run(m):
        a, b = uf.clone().requires_grad_(True), kf.clone().requires_grad_(True)
        return [m(unknown, known, a, b)], [a, b]
    _compare(run, fp, monkeypatch)


def test_voting_module(cuda, monkeypatch):
    torch.manual_seed(1)
    vg = voting_module.VotingModule(1, 256).to(cuda)
    xyz = torch.rand(2, 1024, 3, device=cuda)
    feats = torch.randn(2, 256, 1024, device=cuda)

    def run(m):
        f = feats.clone().requires_grad_(True)
        vx, vf = m(xyz, f)
        return [vx, vf], [f]
    _compare(run, vg, monkeypatch, zero_ok=("dconv1.bias", "dconv2.bias"))


def test_proposal_head(cuda, monkeypatch):
    cfg = config.scannet_md40()
    torch.manual_seed(2)
    pm = proposal_module.ProposalModule(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                        cfg.mean_size_arr, 64, 'vote_fps').to(cuda)
    xyz = torch.rand(2, 512, 3, device=cuda) * 2
    feats = torch.randn(2, 256, 512, device=cuda) * 0.1
    monkeypatch.setenv("BTR_FUSED_SA", "0")   # the head is what is compared here

    def run(m):
        x = xyz.clone().requires_grad_(True)
        f = feats.clone().requires_grad_(True)
        end = m(x, f, {'seed_xyz': xyz})
        return [end['_head_output'], end['center']], [x, f]
    _compare(run, pm, monkeypatch, zero_ok=("dconv1.bias", "dconv2.bias"))


@pytest.mark.parametrize("native", ["1", "0"])
def test_chain_backward_through_the_fused_layer_kernel(cuda, monkeypatch, native):
    """Round 4: a hidden chain layer behind a BatchNorm runs its backward as one
    btr_sa_bwd_fused call (BTR_CHAIN_FUSED=0: weight-gradient + input-gradient GEMMs and the
    BatchNorm-backward passes).  Both sequences against float64 torch ops on the same chain --
    three BatchNorm layers + a bare last layer, padded last width -- and against each other."""
    from backtoreality_amd.pointnet2 import fused_mlp
    monkeypatch.setenv("BTR_CHAIN_MIN_ROWS", "0")
    monkeypatch.setenv("BTR_NATIVE_LAYERS", native)
    torch.manual_seed(5)
    widths = [128, 256, 256, 128, 79]
    convs = [torch.nn.Conv1d(a, b, 1).to(cuda) for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(w).to(cuda) for w in widths[1:-1]]
    with torch.no_grad():
        for bn in bns:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    chain = [(c, b, True) for c, b in zip(convs[:-1], bns)] + [(convs[-1], None, False)]
    x = torch.randn(4, 128, 1024, device=cuda)
    wgt = torch.randn(4, 79, 1024, device=cuda)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("BTR_CHAIN_FUSED", flag)
        for m in convs + bns:
            m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        out = fused_mlp.run_chain(xi, chain)
        assert out is not None
        (out * wgt).sum().backward()
        res[flag] = [xi.grad] + [p.grad.clone() for m in convs + bns for p in m.parameters()]
    # float64 reference
    c64 = [copy.deepcopy(c).double() for c in convs]
    b64 = [copy.deepcopy(b).double() for b in bns]
    for m in c64 + b64:
        m.zero_grad(set_to_none=True)
    xi = x.double().requires_grad_(True)
    h = xi
    for i, c in enumerate(c64):
        h = c(h)
        if i < len(b64):
            h = torch.relu(b64[i](h))
    (h * wgt.double()).sum().backward()
    ref = [xi.grad] + [p.grad for m in c64 + b64 for p in m.parameters()]
    names = ["dx"] + ["%s.%s" % (type(m).__name__, n) for m in convs + bns
                     for n, _ in m.named_parameters()]
    for name, a, b, r in zip(names, res["0"], res["1"], ref):
        scale = float(r.abs().max()) + 1e-12
        if "Conv1d.bias" in name and scale < 1e-6:
            continue
        ea, eb = float((a - r).abs().max()) / scale, float((b - r).abs().max()) / scale
        # (a convolution bias in front of a BatchNorm: true gradient 0, rounding noise in float64)
        if "Conv1d.bias" in name and float(b.abs().max()) == 0.0:
            continue
        assert eb <= 2 * ea + 1e-5, (name, ea, eb)
        assert eb < 2e-4, (name, eb)


def test_small_problems_stay_on_the_stock_ops(cuda, monkeypatch):
    """The row gate of run_chain (fused_mlp._min_rows): one library call per chain takes every
    size; the Python-sequenced form and the path a HIP-graph capture takes leave chains below
    2048 rows to torch's ops; BTR_CHAIN_MIN_ROWS overrides."""
    from backtoreality_amd.pointnet2 import fused_backbone, fused_mlp
    vg = voting_module.VotingModule(1, 256).to(cuda)
    chain = [(vg.conv1, vg.bn1, True), (vg.conv2, vg.bn2, True), (vg.conv3, None, False)]
    small, large = torch.randn(2, 256, 512, device=cuda), torch.randn(2, 256, 1024, device=cuda)
    assert fused_mlp.run_chain(small, chain) is not None
    with fused_backbone.layerwise():
        assert fused_mlp.run_chain(small, chain) is None
        assert fused_mlp.run_chain(large, chain) is not None
    monkeypatch.setenv("BTR_NATIVE_LAYERS", "0")
    assert fused_mlp.run_chain(small, chain) is None
    monkeypatch.setenv("BTR_CHAIN_MIN_ROWS", "0")
    assert fused_mlp.run_chain(small, chain) is not None


def test_vote_assembly_and_fp_weights_match_the_torch_composition(cuda, monkeypatch):
    """The one-launch vote assembly (btr_vote_assemble) and the FP module's in-kernel blend
    weights (btr_three_nn_weights) against the reference's torch op strings: outputs and
    gradients."""
    from backtoreality_amd.pointnet2 import pointnet2_utils
    monkeypatch.setenv("BTR_CHAIN_MIN_ROWS", "0")
    torch.manual_seed(5)
    vg = voting_module.VotingModule(1, 256).to(cuda)
    xyz = torch.rand(2, 700, 3, device=cuda)
    feats = torch.randn(2, 256, 700, device=cuda)
    for normalize in (False, True):   # True: + the L2 normalisation VoteNet.forward applies next
        res = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("BTR_FUSED_VOTES", flag)
            m = copy.deepcopy(vg)
            f = feats.clone().requires_grad_(True)
            vx, vf = m(xyz, f, normalize=normalize)
            if flag == "1":
                assert torch.equal(_ext.twin_of(vf), vf.transpose(1, 2))
            ((vx * torch.linspace(-1, 1, vx.numel(), device=cuda).view_as(vx)).sum() +
             (vf * torch.linspace(0.5, 1.5, vf.numel(), device=cuda).view_as(vf)).sum()).backward()
            res[flag] = {"vx": vx.detach(), "vf": vf.detach(), "df": f.grad,
                         "dw3": m.conv3.weight.grad, "db3": m.conv3.bias.grad,
                         "dw1": m.conv1.weight.grad}
        for k, want in res["0"].items():
            assert _rel(res["1"][k], want) < 2e-5, (normalize, k, _rel(res["1"][k], want))

    unknown = torch.rand(2, 1024, 3, device=cuda)
    known = unknown[:, ::2].contiguous() + 0.01
    idx, weight = pointnet2_utils.three_nn_weights(unknown, known)
    dist, idx0 = pointnet2_utils.three_nn(unknown, known)
    r = 1.0 / (dist + 1e-8)
    assert torch.equal(idx, idx0)
    assert _rel(weight, r / r.sum(2, keepdim=True)) < 1e-6
    few = known[:, :2].contiguous()                       # fewer than three known points
    _, w2 = pointnet2_utils.three_nn_weights(unknown, few)
    d2, _ = pointnet2_utils.three_nn(unknown, few)
    r2 = 1.0 / (d2 + 1e-8)
    assert torch.allclose(w2, r2 / r2.sum(2, keepdim=True), rtol=1e-6, atol=1e-12)
