"""f16x3 vs f32 chain on the golden fixtures (developer tool)."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0,"/root/repo/tests/golden")
import numpy as np, torch
from clavaddpm_amd import split_nets as sn_
from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
from conftest import golden, sub, relerr
dev="cuda"
g = golden("g5_chain_cond.npz")
T, n_num, C = int(g["T"]), int(g["n_num"]), int(g["C"])
net = MLPDiffusion(n_num, 0, "none", {"d_layers": [64, 64], "dropout": 0.0}); net.load_state_dict(sub(g,"p:")); net=net.to(dev).eval()
clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=g["hidden"].tolist()); clf.load_state_dict(sub(g,"c:")); clf=clf.to(dev).eval()
d = GaussianMultinomialDiffusion(np.array([0]), n_num, net, num_timesteps=T, device=torch.device(dev)).to(dev).eval()
ys = g["ys"]
for steps in (1, 5, 20, 50):
    for cf in (None, make_cond_fn(clf, 1.0)):
        sn_.set_gemm_mode("f16x3")
        x1, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, noise=sub(g, "nz:"), n_steps=steps)
        sn_.set_gemm_mode("f32")
        x2, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, noise=sub(g, "nz:"), n_steps=steps)
        print("guided" if cf else "unguided", "steps", steps, "f16x3 vs f32", relerr(x1, x2), "max|x|", float(x2.abs().max()),
              ("vs golden %.2e" % relerr(x1, g["sample"])) if steps == 50 and cf else "")
