"""cat_wgrad kernel alone at the scale-sweep shape (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import training as tr
from clavaddpm_amd._lib import lib, check, ptr, c_int, c_float, stream_ptr
from clavaddpm_amd.training import c_ll
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion

def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

B, dev, n_num, S = 4096, "cuda", 100, 8
K = np.array([int(v) for v in np.exp(np.random.RandomState(0).uniform(np.log(2), np.log(1000), 50)).round().clip(2, 1000)])
sk = int(K.sum()); D = n_num + sk
net = MLPDiffusion(D, 0, "none", {"d_layers": [64, 64], "dropout": 0.0}).to(dev)
d = GaussianMultinomialDiffusion(K, n_num, net, num_timesteps=1000, device=torch.device(dev)).to(dev)
st = tr.SplitTrainState()
gp = st.gather_plan(d, dev)
gen = torch.Generator(device=dev).manual_seed(0)
idx = torch.stack([torch.randint(0, int(k), (B,), device=dev, generator=gen) for k in K], 1).to(torch.int32).contiguous()
g = torch.randn(B, 128, device=dev, generator=gen)
total = 128 * D + 128
slabs = torch.zeros((S, total), device=dev)
bsl = slabs[:, 128 * D:]
bsl.copy_(g.view(S, B // S, 128).sum(1))
def run():
    check(lib().clava_cat_wgrad_f32(ptr(g), c_ll(128), ptr(bsl), c_int(B), c_int(128), ptr(idx), c_int(gp["m"]), ptr(d._offsets_i32), ptr(gp["wf"]),
                                    ptr(gp["wk"]), c_int(gp["nwg"]), c_int(S), c_float(gp["log_tiny"]), ptr(slabs), c_ll(D), c_int(n_num), c_ll(total),
                                    stream_ptr()), "cat_wgrad")
print(f"cat_wgrad alone: {timeit(run):.1f} us  ({gp['nwg']} x {S} workgroups)")
# reference
x = torch.full((B, sk), gp["log_tiny"], device=dev, dtype=torch.float64)
off = d._offsets_i32.long()
x.scatter_(1, (idx.long() + off[:-1][None, :]), 0.0)
ref = g.double().T @ x
got = slabs[:, :128 * D].double().sum(0).view(128, D)[:, n_num:]
print("rel err", float((got - ref).abs().max() / ref.abs().max()))
