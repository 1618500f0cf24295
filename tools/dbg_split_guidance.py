"""Debug: the guidance gradient the split-f16 chain applies, recovered from one-step chains, vs the float32 path / oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from weights import classifier_shapes, mlp_shapes, seeded_state
from clavaddpm_amd import split_nets
from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
from oracle import guidance

DEV = "cuda"
DL = [512, 1024, 1024, 1024, 1024, 512]
CL = [128, 256, 512, 1024, 512, 256, 128]
n_num, C, T, b = 3, 20, 1000, 192
ps = seeded_state(mlp_shapes(n_num, DL), 91, scale=0.5)
cs = seeded_state(classifier_shapes(n_num, C, CL), 92)
net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(DL), "dropout": 0.0}); net.load_state_dict(ps)
clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=CL); clf.load_state_dict(cs)
d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.to(DEV).eval(), num_timesteps=T, device=torch.device(DEV)).to(DEV).eval()
clf = clf.to(DEV).eval()
g = torch.Generator().manual_seed(5)
ys = torch.randint(0, C, (b, 1), generator=g)
for mag in (1.0, 30.0, 100.0):
    z0 = torch.randn(b, n_num, generator=g) * mag
    nz = {"z0": z0, "noise": torch.zeros(1, b, n_num).double()}
    t = torch.full((b,), T - 1, dtype=torch.long)
    gref = guidance.cond_grad(cs, z0, t, ys, 1.0).double()
    var = float(d._model_variance[T - 1])
    res = {}
    for mode in ("f32", "f16x3"):
        split_nets.set_gemm_mode(mode)
        x0, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=make_cond_fn(clf, 0.0), noise=nz, n_steps=1)
        S = 1000.0
        x1, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=make_cond_fn(clf, S), noise=nz, n_steps=1)
        res[mode] = (x1 - x0) / (var * S)
    split_nets.set_gemm_mode("f16x3")
    r = lambda a, b_: float((a - b_).abs().max() / b_.abs().max())
    print("input scale %5.1f  max|grad| %.3e   f32 chain vs oracle %.2e   f16x3 chain vs oracle %.2e   fallbacks %d" % (
        mag, float(gref.abs().max()), r(res["f32"], gref), r(res["f16x3"], gref), split_nets.fallback_count))
