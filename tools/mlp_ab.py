"""Point-wise MLP chains, fused (fused_mlp.py) vs stock torch ops: GPU time (events) and host
enqueue time per forward+backward at the benchmark shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import pointnet2_modules as M
from backtoreality_amd.votenet import config, proposal_module, voting_module

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
B = 8


def bench(name, mod, make_inputs, call):
    for flag in ("0", "1"):
        os.environ["BTR_FUSED_MLP"] = flag
        ins = make_inputs()
        def step():
            for t in ins:
                t.grad = None
            outs = call(mod, *ins)
            sum(o.sum() for o in outs).backward()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        host = 0.0
        e0.record()
        for _ in range(20):
            t0 = time.perf_counter()
            step()
            host += time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        print("%-10s fused=%s  GPU %.3f ms  host enqueue %.3f ms per fwd+bwd" % (
            name, flag, e0.elapsed_time(e1) / 20, host / 20 * 1e3))


torch.manual_seed(0)
fp2 = M.PointnetFPModule(mlp=[512, 256, 256]).to(dev)
unknown = torch.rand(B, 1024, 3, device=dev); known = unknown[:, :512].contiguous()
bench("fp2", fp2, lambda: [torch.randn(B, 256, 1024, device=dev, requires_grad=True),
                           torch.randn(B, 256, 512, device=dev, requires_grad=True)],
      lambda m, a, b: [m(unknown, known, a, b)])
vg = voting_module.VotingModule(1, 256).to(dev)
xyz = torch.rand(B, 1024, 3, device=dev)
bench("vgen", vg, lambda: [torch.randn(B, 256, 1024, device=dev, requires_grad=True)],
      lambda m, f: list(m(xyz, f)))
os.environ["BTR_FUSED_SA"] = "1"
pm = proposal_module.ProposalModule(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster,
                                    cfg.mean_size_arr, 256, 'vote_fps').to(dev)
vxyz = torch.rand(B, 1024, 3, device=dev) * 3
bench("proposal", pm, lambda: [torch.randn(B, 256, 1024, device=dev, requires_grad=True)],
      lambda m, f: [m(vxyz, f, {'seed_xyz': vxyz})['_head_output']])
