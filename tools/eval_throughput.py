#!/usr/bin/env python3
"""Inference throughput at BASELINE config[1] (8 scenes x 40 000 points): the eval-mode forward
alone (net.eval(), no_grad: the fused inference path of the five SA layers) and the whole
evaluation pass of the training scripts (train.evaluate_one_epoch: forward, loss statistics,
parse_predictions / parse_groundtruths, AP; reference train_Votenet_FSB.py:246-293)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import config, synthetic, train  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
net = train.build_model(cfg, dev)
B, N = 8, 40000
batches = [synthetic.make_batch(1000 * i, B, N, cfg, device=dev) for i in range(4)]
opt = train.make_optimizer(net)
for b in batches[:2]:          # running statistics that are not the initial ones
    train.train_step(net, opt, b, cfg)
net.eval()


def forward_only(n):
    with torch.no_grad():
        for i in range(n):
            net({'point_clouds': batches[i % 4]['point_clouds']})


def forward_prefetched(n):
    """the next batch's sampling pyramid under this batch's forward (coordinates only)"""
    bb = net.backbone_net
    with torch.no_grad():
        s = bb.prefetch_sampling(batches[0]['point_clouds'])
        for i in range(n):
            nxt = bb.prefetch_sampling(batches[(i + 1) % 4]['point_clouds'])
            net({'point_clouds': batches[i % 4]['point_clouds'], 'sampling': s})
            s = nxt


def forward_depth(depth):
    """`depth` pyramids in flight on alternating prefetch streams (the inference forward is ~1 ms
    of main-stream work against 2.2 - 2.7 ms of dependent FPS steps per pyramid)"""
    def run(n):
        import collections
        bb = net.backbone_net
        q = collections.deque()
        with torch.no_grad():
            for j in range(depth):
                q.append(bb.prefetch_sampling(batches[j % 4]['point_clouds'], slot=j % depth))
            for i in range(n):
                s = q.popleft()
                q.append(bb.prefetch_sampling(batches[(i + depth) % 4]['point_clouds'],
                                              slot=(i + depth) % depth))
                net({'point_clouds': batches[i % 4]['point_clouds'], 'sampling': s})
    return run


CASES = (("forward only", forward_only), ("forward, pyramid prefetched", forward_prefetched),
         ("forward, 2 pyramids in flight", forward_depth(2)),
         ("forward, 3 pyramids in flight", forward_depth(3)),
         ("forward, 4 pyramids in flight", forward_depth(4)))
if os.environ.get("EVAL_ONLY") == "forward":   # (tools/profile_eval.sh: one case under the profiler)
    CASES = CASES[:1]
for name, fn in CASES:
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(20)
    th = (time.perf_counter() - t0) / 20     # host done enqueuing
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-30s %.2f ms per batch = %.0f scenes/s (host enqueue %.2f ms)" % (
        name, dt * 1e3, B / dt, th * 1e3))
if os.environ.get("EVAL_ONLY"):
    sys.exit(0)
train.evaluate_one_epoch(net, batches[:1], cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
stats, metrics = train.evaluate_one_epoch(net, batches, cfg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(batches)
print("%-30s %.2f ms per batch = %.0f scenes/s" % ("evaluate_one_epoch", dt * 1e3, B / dt))
