"""One split-f16 GEMM shape / tile configuration, a few launches: the target of rocprofv3 --pmc passes (developer tool).
    python3 tools/pmc_f16x3.py N K cfg [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import ops
N, K, cfg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
g = torch.Generator(device="cuda").manual_seed(0)
x = ops.split_pack(torch.randn(M, K, device="cuda", generator=g), 0)
w = ops.split_pack(torch.randn(N, K, device="cuda", generator=g) * 0.05, 12)
b = torch.randn(N, device="cuda", generator=g)
out = ops.split_empty(M, N, 0, "cuda")
for _ in range(10):
    ops.linear_f16x3(x, w, b, act="relu", tile_config=cfg, out=out)
torch.cuda.synchronize()
