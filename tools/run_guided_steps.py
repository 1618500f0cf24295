"""A few guided reverse steps at the bench shapes (20000 rows, n_num 2, 50 clusters): the target of rocprofv3 passes.
    python3 tools/run_guided_steps.py [steps] [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from weights import classifier_shapes, mlp_shapes, seeded_state
from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
DL, CL, n_num, C, T = [512, 1024, 1024, 1024, 1024, 512], [128, 256, 512, 1024, 512, 256, 128], 2, 50, 1000
net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(DL), "dropout": 0.0}); net.load_state_dict(seeded_state(mlp_shapes(n_num, DL), 3, scale=0.3))
clf = Classifier(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=CL); clf.load_state_dict(seeded_state(classifier_shapes(n_num, C, CL), 4))
d = GaussianMultinomialDiffusion(np.array([0]), n_num, net.cuda().eval(), num_timesteps=T, device=torch.device("cuda")).cuda().eval()
ys = torch.randint(0, C, (B, 1), generator=torch.Generator().manual_seed(0))
cf = make_cond_fn(clf.cuda().eval(), 1.0)
for rep in range(2):
    x, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cf, seed=11, n_steps=steps)
print("ok", float(x.abs().max()))
