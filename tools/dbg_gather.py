"""Gather vs dense first layer against the CPU oracle on a pure-categorical table with more than 64 features (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from weights import mlp_shapes, seeded_state
from clavaddpm_amd import split_nets
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
from oracle import chain, schedule
n_num, K, b, T, steps = 0, [5] * 70, 258, 1000, 2
sumK = sum(K)
state = seeded_state(mlp_shapes(n_num + sumK, [256, 512, 256]), 17, scale=0.05)
net = MLPDiffusion(n_num + sumK, 0, "none", {"d_layers": [256, 512, 256], "dropout": 0.0}); net.load_state_dict(state)
d = GaussianMultinomialDiffusion(np.array(K), n_num, net.cuda().eval(), num_timesteps=T, device=torch.device("cuda")).cuda().eval()
g = torch.Generator().manual_seed(5)
nz = {"z0": torch.randn(b, n_num, generator=g), "u_init": torch.rand(b, sumK, generator=g), "noise": torch.randn(steps, b, n_num, generator=g).double(),
      "u": torch.rand(steps, b, sumK, generator=g), "y": torch.zeros(b, dtype=torch.long)}
ref = chain.sample(schedule.tables("cosine", T), T, state, n_num, K, nz, steps=steps)
res = {}
for mode, flag in (("f32", "0"), ("f16x3", "0"), ("f16x3", "1")):
    os.environ["CLAVA_CAT_GATHER"] = flag
    split_nets.set_gemm_mode(mode)
    x, _ = d.sample(b, torch.ones(1), noise=nz, n_steps=steps)
    res[(mode, flag)] = x
    bad = (x != ref).sum(0)
    print(mode, "gather" if flag == "1" else "dense ", "mismatching draws per feature vs oracle:", bad.tolist(), "fallbacks", split_nets.fallback_count)
print("f16x3 dense vs f32:", (res[("f16x3", "0")] != res[("f32", "0")]).sum(0).tolist())
print("f16x3 gather vs f32:", (res[("f16x3", "1")] != res[("f32", "0")]).sum(0).tolist())
for st in (0, 1):
    r2 = chain.sample(schedule.tables("cosine", T), T, state, n_num, K, nz, steps=st)
    x, _ = d.sample(b, torch.ones(1), noise=nz, n_steps=st)
    print("steps", st, "f16x3 gather vs oracle:", (x != r2).sum(0).tolist())
