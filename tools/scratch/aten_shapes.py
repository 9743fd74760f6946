import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import ProfilerActivity, profile
from backtoreality_amd.votenet import config, synthetic, train
dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
opt = train.make_optimizer(net)
batch = synthetic.make_batch(0, 8, 40000, cfg, device=dev)
for _ in range(4):
    train.train_step(net, opt, batch, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train.train_step(net, opt, batch, cfg)
    torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::mul", "aten::sum", "aten::cat", "aten::index_select", "aten::gather"):
        dt = getattr(e, "device_time_total", 0)
        st = [s for s in (e.stack or []) if "backtoreality" in s or "train" in s][:3]
        print("%-16s %7.1f us  %s  %s" % (e.name, dt, e.input_shapes, " <- ".join(st)))
