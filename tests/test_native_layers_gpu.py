"""Whole-layer entry points (csrc/sa_layer.hip: btr_sa_layer_* / btr_pm_chain_*, one call per
layer) against the same launches sequenced from Python (fused_sa.FusedSAFunction /
fused_mlp.PointwiseMLP, BTR_NATIVE_LAYERS=0): same kernels in the same order, so every output,
gradient and BatchNorm buffer must be BIT-identical -- except the bias gradient of a bare last
layer, which the library sums itself (torch.sum in the Python sequence), and the scattered
input gradients, whose summation order is the order atomics filled the per-point lists in
(not reproducible between two runs of the SAME path either): 1e-5 relative."""
import copy

import numpy as np
import pytest
import torch

from backtoreality_amd.pointnet2 import fused_mlp, fused_sa
from backtoreality_amd.pointnet2 import _ext
from backtoreality_amd.pointnet2 import pointnet2_modules as M
from backtoreality_amd.pointnet2 import pointnet2_utils
from backtoreality_amd.votenet import config, proposal_module, synthetic, voting_module

pytestmark = pytest.mark.gpu


def _sa_run(sa, xyz, feats, inds, xyz_grad, feat_grad=True):
    xyz = xyz.clone().requires_grad_(xyz_grad)
    feats = feats.clone().requires_grad_(feat_grad) if feats is not None else None
    new_xyz, out, _ = sa(xyz, feats, inds)
    torch.manual_seed(3)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g = {"out": out.detach(), "out_cl": _ext.twin_of(out).detach()}
    if feats is not None and feat_grad:
        g["dfeat"] = feats.grad
    if xyz_grad:
        g["dxyz"] = xyz.grad
    for n, p in sa.named_parameters():
        g["d" + n] = p.grad
    for n, b in sa.named_buffers():
        g[n] = b.detach().clone()
    return g


@pytest.mark.parametrize("N,npoint,radius,S,mlp,C,xyz_grad,feat_grad", [
    (4096, 512, 0.2, 64, [1, 64, 64, 128], 1, False, False),      # SA1: compact + recompute
    (4096, 512, 0.2, 64, [1, 64, 64, 128], 1, False, True),       # SA1-shaped, feature gradient
    (2048, 256, 0.4, 32, [128, 128, 128, 256], 128, False, True),  # SA2: compact, scatter
    (1024, 256, 0.3, 16, [256, 128, 128, 128], 256, True, True),   # vote aggregation
    (1024, 128, 0.8, 16, [0, 32, 48], 0, True, True),              # no features, 2 layers
    (700, 100, 0.5, 7, [5, 20], 5, True, True),                    # ragged, single layer
    (3000, 64, 1.0, 128, [3, 32, 64], 3, True, True),              # nsample 128
    (9000, 256, 0.3, 16, [8, 32, 64], 8, True, True),              # N > 8192
    (2048, 256, 0.4, 32, [128, 128, 128, 256], 128, True, True),   # xyz gradient: dense rows
])
@pytest.mark.parametrize("options", ["default", "plain"])
def test_sa_layer_call_equals_python_sequence(cuda, monkeypatch, N, npoint, radius, S, mlp, C,
                                              xyz_grad, feat_grad, options):
    if options == "plain":   # no compact rows / recompute / pooling epilogue / prologue gradient
        for k in ("BTR_SA_COMPACT", "BTR_SA_RECOMPUTE", "BTR_POOL_EPILOGUE", "BTR_POOLGRAD"):
            monkeypatch.setenv(k, "0")
    # (the pooled layer's Gram-form backward and the per-point first layer exist in the whole-layer
    # call only: the Python sequence reads Y_l and multiplies row by row;
    # test_gram_form_equals_the_y_reading_form / test_per_point_first_layer below compare them)
    monkeypatch.setenv("BTR_POOL_GRAM", "0")
    monkeypatch.setenv("BTR_SA_PPFL", "0")
    B = 2
    xyz = torch.from_numpy(np.stack([synthetic.make_scene(60 + i, N, use_height=False)[
        'point_clouds'] for i in range(B)], 0)).to(cuda)
    torch.manual_seed(0)
    feats = torch.randn(B, C, N, device=cuda) if C else None
    sa = M.PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=S, mlp=list(mlp),
                                 use_xyz=True, normalize_xyz=True).to(cuda)
    with torch.no_grad():
        for layer in sa.mlp_module:
            layer.bn.bn.weight.uniform_(0.5, 1.5)
            layer.bn.bn.bias.uniform_(-0.3, 0.3)
    inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("BTR_NATIVE_LAYERS", flag)
        assert fused_sa.native_enabled() == (flag == "1")
        res[flag] = _sa_run(copy.deepcopy(sa), xyz, feats, inds, xyz_grad, feat_grad)
    assert set(res["0"]) == set(res["1"])
    for k, want in res["0"].items():
        got = res["1"][k]
        assert (want is None) == (got is None), k
        if want is not None:
            assert got.shape == want.shape and got.dtype == want.dtype, k
            if k in ("dfeat", "dxyz"):   # summed in the order atomics built the point lists
                rel = float((got - want).abs().max() / want.abs().max())
                assert rel < 1e-5, (k, rel)
            else:
                assert torch.equal(got, want), (k, float((got.float() - want.float()).abs().max()))


@pytest.mark.parametrize("N,npoint,radius,S,mlp,C,feat_grad", [
    (8192, 1024, 0.2, 64, [1, 64, 64, 128], 1, False),        # SA1: compact rows + recompute
    (4096, 512, 0.4, 32, [32, 64, 64, 128], 32, True),        # features, groups of 32
    (4096, 512, 0.3, 64, [8, 32, 64, 128], 8, True),          # three layers behind 8 features
    (4096, 512, 0.4, 32, [16, 64, 100], 16, True),            # two layers, n = 100
])
def test_gram_form_equals_the_y_reading_form(cuda, monkeypatch, N, npoint, radius, S, mlp, C,
                                             feat_grad):
    """The pooled layer without its stored pre-BN output (BTR_SA_OPT_POOL_GRAM: forward keeps
    the group extrema only, btr_sa_bwd_gram forms dW_l / dZ_{l-1} from X_{l-1}) against the
    Y_l-reading whole-layer call: the forward is the same arithmetic (outputs and BatchNorm
    buffers bit-identical), the gradients agree to float32 rounding of their largest entry."""
    B = 2
    xyz = torch.from_numpy(np.stack([synthetic.make_scene(60 + i, N, use_height=False)[
        'point_clouds'] for i in range(B)], 0)).to(cuda)
    torch.manual_seed(0)
    feats = torch.randn(B, C, N, device=cuda) if C else None
    sa = M.PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=S, mlp=list(mlp),
                                 use_xyz=True, normalize_xyz=True).to(cuda)
    with torch.no_grad():
        for layer in sa.mlp_module:
            layer.bn.bn.weight.uniform_(-1.5, 1.5)     # (negative scales: minima are pooled)
            layer.bn.bn.bias.uniform_(-0.3, 0.3)
    inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    res, plans = {}, {}
    for flag in ("0", "2"):   # ("2": any value but "0" -- the Gram form wherever it applies)
        monkeypatch.setenv("BTR_POOL_GRAM", flag)
        mod = copy.deepcopy(sa)
        res[flag] = _sa_run(mod, xyz, feats, inds, False, feat_grad)
        plans[flag] = [ent[1].pool_grad for ent in fused_sa._LAYER_CACHE.get(mod, {}).values()]
    assert plans["0"] == [1] and plans["2"] == [2], plans   # (the variant that actually ran)
    for k, want in res["0"].items():
        got = res["2"][k]
        assert (want is None) == (got is None), k
        if want is None:
            continue
        if k in ("out", "out_cl") or "running" in k or "tracked" in k:
            assert torch.equal(got, want), k
        else:
            rel = float((got - want).abs().max() / want.abs().max())
            assert rel < 2e-5, (k, rel)


@pytest.mark.parametrize("N,npoint,radius,S,mlp,C,xyz_grad", [
    (4096, 1024, 0.4, 32, [128, 128, 128, 256], 128, False),    # SA2: compact rows
    (2048, 512, 0.8, 16, [256, 128, 128, 256], 256, False),     # SA3 / SA4: dense rows of 16
    (3000, 300, 0.5, 32, [64, 100, 128], 64, False),            # ragged, first layer of 100 columns
    (1024, 256, 0.3, 16, [256, 128, 128, 128], 256, True),      # vote aggregation: xyz gradient
    (1500, 200, 0.6, 8, [32, 64, 64], 32, True),                # xyz gradient, ragged
])
def test_per_point_first_layer(cuda, monkeypatch, N, npoint, radius, S, mlp, C, xyz_grad):
    """BTR_SA_OPT_PPFL (W_f f_j once per point, the rows gather it; dW_f / dF as products over the
    points behind a per-point sum of dY_0) against the row-wise first layer: the same sums in
    another order -- outputs, BatchNorm buffers and every gradient to float32 rounding."""
    B = 2
    xyz = torch.from_numpy(np.stack([synthetic.make_scene(70 + i, N, use_height=False)[
        'point_clouds'] for i in range(B)], 0)).to(cuda)
    torch.manual_seed(0)
    feats = torch.randn(B, C, N, device=cuda)
    sa = M.PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=S, mlp=list(mlp),
                                 use_xyz=True, normalize_xyz=True).to(cuda)
    with torch.no_grad():
        for layer in sa.mlp_module:
            layer.bn.bn.weight.uniform_(0.5, 1.5)
            layer.bn.bn.bias.uniform_(-0.3, 0.3)
    inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    res = {}
    monkeypatch.setenv("BTR_SA_PPFL_XYZ", "1")   # (the form with coordinate gradients: off by default)
    for flag in ("0", "1"):
        monkeypatch.setenv("BTR_SA_PPFL", flag)
        mod = copy.deepcopy(sa)
        res[flag] = _sa_run(mod, xyz, feats, inds, xyz_grad, True)
        import ctypes
        took = [_ext._lib.btr_sa_layer_ppfl(ctypes.addressof(ent[0]), ctypes.addressof(ent[1]))
                for ent in fused_sa._LAYER_CACHE.get(mod, {}).values()]
        assert took == [int(flag)], (flag, took)   # (the variant that actually ran)
    for k, want in res["0"].items():
        got = res["1"][k]
        assert (want is None) == (got is None), k
        if want is None:
            continue
        # (Y_0 differs in its last bits, so a max-pool element within rounding of a tie may pick its
        # other candidate: outputs stay at rounding, a gradient may move by one element's share --
        # relative L2 as in tests/test_configs_gpu.py; without a flip everything sits at ~4e-7)
        if k.startswith("d"):
            rel = float((got.float() - want.float()).norm() / (want.float().norm() + 1e-30))
            assert rel < 1e-2, (k, rel)
        else:
            rel = float((got.float() - want.float()).abs().max() /
                        (want.float().abs().max() + 1e-30))
            assert rel < 2e-6, (k, rel)


def _chain_compare(run, mod, monkeypatch, summed=()):
    monkeypatch.setenv("BTR_CHAIN_MIN_ROWS", "0")   # (small test shapes: below the default gate)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("BTR_NATIVE_LAYERS", flag)
        assert fused_mlp.native_enabled() == (flag == "1")
        m = copy.deepcopy(mod)
        outs, ins = run(m)
        loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=o.device).view_as(o)).sum()
                   for o in outs)
        loss.backward()
        res[flag] = {"out%d" % i: o.detach() for i, o in enumerate(outs)}
        res[flag].update({"din%d" % i: t.grad for i, t in enumerate(ins)})
        res[flag].update({"d" + n: p.grad for n, p in m.named_parameters()})
        res[flag].update({n: b.detach().clone() for n, b in m.named_buffers()})
    for k, want in res["0"].items():
        got = res["1"][k]
        assert got.shape == want.shape, k
        if k in summed or k.startswith("din") or k.endswith("running_mean"):
            # din*: behind atomically built lists (three_interpolate / scatter backward);
            # running_mean: momentum * bias added by the library (no fma) vs torch.add_(alpha=)
            rel = float((got - want).abs().max() / want.abs().max())
            assert rel < (1e-6 if k.endswith("running_mean") else 1e-5), (k, rel)
        else:
            assert torch.equal(got, want), (k, float((got.float() - want.float()).abs().max()))


def test_fp_chain_call_equals_python_sequence(cuda, monkeypatch):
    torch.manual_seed(0)
    fp = M.PointnetFPModule(mlp=[256 + 256, 256, 256]).to(cuda)
    unknown = torch.rand(2, 1024, 3, device=cuda)
    known = unknown[:, :512].contiguous()
    uf = torch.randn(2, 256, 1024, device=cuda)
    kf = torch.randn(2, 256, 512, device=cuda)

    def run(m):
        a, b = uf.clone().requires_grad_(True), kf.clone().requires_grad_(True)
        return [m(unknown, known, a, b)], [a, b]
    _chain_compare(run, fp, monkeypatch)


def test_voting_chain_call_equals_python_sequence(cuda, monkeypatch):
    torch.manual_seed(1)
    vg = voting_module.VotingModule(1, 256).to(cuda)
    xyz = torch.rand(2, 1024, 3, device=cuda)
    feats = torch.randn(2, 256, 1024, device=cuda)

    def run(m):   # 259 output channels: the padded (260) last layer
        f = feats.clone().requires_grad_(True)
        vx, vf = m(xyz, f)
        return [vx, vf], [f]
    _chain_compare(run, vg, monkeypatch, summed=("dconv3.bias",))


def test_proposal_chain_call_equals_python_sequence(cuda, monkeypatch):
    cfg = config.scannet_md40()
    torch.manual_seed(2)
    pm = proposal_module.ProposalModule(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                        cfg.mean_size_arr, 64, 'vote_fps').to(cuda)
    xyz = torch.rand(2, 512, 3, device=cuda) * 2
    feats = torch.randn(2, 256, 512, device=cuda) * 0.1

    # (the per-point first layer of the vote aggregation exists in the whole-layer call only:
    # test_per_point_first_layer compares it with the row-wise form)
    monkeypatch.setenv("BTR_SA_PPFL", "0")

    def run(m):   # vote aggregation (SA layer with coordinate gradients) + the head chain
        x = xyz.clone().requires_grad_(True)
        f = feats.clone().requires_grad_(True)
        end = m(x, f, {'seed_xyz': xyz})
        return [end['_head_output'], end['center']], [x, f]
    _chain_compare(run, pm, monkeypatch, summed=("dconv3.bias",))


def test_descriptions_are_cached_per_shape_and_modules_stay_copyable(cuda):
    sa = M.PointnetSAModuleVotes(npoint=64, radius=0.5, nsample=16, mlp=[0, 16, 32],
                                 use_xyz=True, normalize_xyz=True).to(cuda)
    for n in (512, 512, 700):
        xyz = torch.rand(2, n, 3, device=cuda)
        sa(xyz, None)
    assert len(fused_sa._LAYER_CACHE[sa]) == 2
    clone = copy.deepcopy(sa)            # ctypes descriptions are not part of the module
    assert clone not in fused_sa._LAYER_CACHE
    clone(torch.rand(2, 512, 3, device=cuda), None)


def test_in_kernel_batchnorm_finalisation_at_every_size():
    """The statistics GEMM's last workgroup finalises the BatchNorm for layers of <= 2 048 rows
    by default (csrc/internal.hpp BnFin; larger layers measured slower that way).  The comparisons
    of this file with the limit lifted -- 512-workgroup grids, two column blocks -- in a child
    process (the switch is read once per process): still bit-identical to the finalize kernel."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the streaming GEMM kernel steps aside for an armed ticket: keep both sequences on
    # gemm_nt_kernel, whose statistics this test compares bit for bit)
    env = dict(os.environ, BTR_BN_TICKET_MAX_ROWS="1000000000", BTR_FWD_STREAM="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x",
                        "-k", "equals_python_sequence", "-p", "no:cacheprovider"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900)
    tail = r.stdout.decode(errors="replace")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


@pytest.mark.parametrize("env", [
    {"BTR_GRID_CUS": "100"},                            # far fewer row chunks than the default
    {"BTR_GRID_CUS": "256", "BTR_FPS_LDS_KB": "0"},     # every CU counted, no LDS held by the FPS
    {"BTR_WGRAD_STREAM": "1"},                          # the round-3 stream arrangement
])
def test_chunk_count_knobs_leave_the_results_alone(env):
    """The row-chunk counts of the streaming / fused / weight-gradient kernels are sized from the
    CUs a launch can use (csrc/fps_bucket.hip grid_cus, BTR_GRID_CUS); the partial-sum buffers
    are sized by the same functions at plan time and at launch time.  This file's C-sequence-vs-Python-sequence comparisons (bit-identical outputs,
    gradients, running statistics) and the golden training step, in child processes with other
    chunk counts (the switches are read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__),
                        os.path.join(root, "tests", "test_golden_cpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "equals_python_sequence or votenet_step", "-p", "no:cacheprovider"],
                       cwd=root, env=child, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=1200)
    tail = r.stdout.decode(errors="replace")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
