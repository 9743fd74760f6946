import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from backtoreality_amd.votenet import config, loss_helper, synthetic, train
dev = torch.device("cuda:0")
def rel(a,b): return float((a-b).abs().max()/(b.abs().max()+1e-12))
for (cfgf, N, es) in [(config.scannet_md40, 40000, 1.0), (config.matterport_md40, 80000, 1.7)]:
    cfg = cfgf()
    batch = synthetic.make_batch(0, 2, N, cfg, extent_scale=es, device=dev)
    outs = []
    for fused in ("1", "0"):
        os.environ["BTR_FUSED_SA"] = fused
        net = train.build_model(cfg, dev, seed=0)
        end = net({'point_clouds': batch['point_clouds']})
        outs.append({k: v.detach() for k, v in end.items() if torch.is_tensor(v)})
    f, u = outs
    print(N, {k: "%.1e" % rel(f[k].float(), u[k].float()) for k in ('sa1_features','sa2_features','sa3_features','sa4_features','fp2_features','vote_xyz','aggregated_vote_features')},
          "inds equal:", bool(torch.equal(f['aggregated_vote_inds'], u['aggregated_vote_inds'])))

print("---- gradient deviations (C5 shape)")
cfg = config.matterport_md40()
batch = synthetic.make_batch(0, 2, 80000, cfg, extent_scale=1.7, device=dev)
gs = []
for fused in ("1", "0"):
    os.environ["BTR_FUSED_SA"] = fused
    net = train.build_model(cfg, dev, seed=0)
    end = net({'point_clouds': batch['point_clouds']}); end.update(batch)
    loss, end = loss_helper.get_loss(end, cfg); loss.backward()
    gs.append({n: p.grad.detach().clone() for n, p in net.named_parameters()})
    print("loss", float(loss))
rows = sorted(((rel(gs[0][n], gs[1][n]), float(gs[1][n].abs().max()), n) for n in gs[1]), reverse=True)[:8]
for r in rows: print("%.2e  max|g|=%.2e  %s" % r)
