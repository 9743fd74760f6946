#!/usr/bin/env python3
"""fps_regs_kernel<4,8> (2048 -> 1024) on a FIXED FPS-ordered input, thousands of launches:
alone on the device, and with a GEMM-like kernel running on another stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from backtoreality_amd.pointnet2 import pointnet2_utils as pu
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)
pc = torch.rand((3, 20000, 3), generator=g) * torch.tensor([6.0, 5.0, 2.5])
pc = pc.to(dev)
i1 = pu.furthest_point_sample(pc, 2048)
xyz1 = pu.gather_operation(pc.transpose(1, 2).contiguous(), i1).transpose(1, 2).contiguous()
torch.cuda.synchronize()
want = torch.arange(1024, device=dev, dtype=torch.int32).expand(3, 1024)
iters = int(os.environ.get("ITERS", "3000"))
from backtoreality_amd.pointnet2 import _ext
lib = _ext._lib
R, NN, KK = 262144, 128, 128
ga = torch.randn(R, KK, device=dev); gw = torch.randn(NN, KK, device=dev); gc = torch.empty(R, NN, device=dev)
gpa = torch.rand(KK, device=dev); gpb = torch.rand(KK, device=dev)
gpart = torch.empty(lib.btr_sa_gemm_grid(R), 2, NN, device=dev)
for mode in ("with_btr_gemm", "with_btr_gemm_many_small"):
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    bad = 0
    for it in range(iters):
        if mode == "with_side_load":
            with torch.cuda.stream(side):
                b = a @ a
        if mode.startswith("with_btr_gemm"):
            reps = 1 if mode == "with_btr_gemm" else 6
            rows = R if mode == "with_btr_gemm" else 8192
            for _ in range(reps):
                lib.btr_sa_gemm_nt(rows, NN, KK, ga.data_ptr(), KK, gw.data_ptr(), KK, gc.data_ptr(), NN,
                                   gpa.data_ptr(), gpb.data_ptr(), gpart.data_ptr(), side.cuda_stream)
        inds = pu.furthest_point_sample(xyz1, 1024)
        if it % 50 == 49 or it == iters - 1:
            torch.cuda.synchronize()
        if not torch.equal(inds, want):
            bad += 1
    torch.cuda.synchronize()
    print(mode, "bad", bad, "of", iters)
