"""Classifier-training step timing (developer tool): graph replay vs eager wall per step, and the summed kernel time per
family from the library's per-launch events.  Shapes: configs/movie_lens.json classifier, batch 4096, 50 clusters."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import _lib
from clavaddpm_amd.classifier_training import ClassifierTrainer
from clavaddpm_amd.pipeline_utils import Classifier
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion
L = _lib.lib(); L.clava_prof_family_name.restype = ctypes.c_char_p; nf = L.clava_prof_family_count()
dev = "cuda"
d_in, C, B = 2, 50, 4096
hidden = [128, 256, 512, 1024, 512, 256, 128]
x = torch.randn(B, d_in); y = torch.randint(0, C, (B,))
for graph in ("1", "0"):
    os.environ["CLAVA_TRAIN_GRAPH"] = graph
    torch.manual_seed(0)
    clf = Classifier(d_in=d_in, d_out=C, dim_t=128, hidden_sizes=hidden).to(dev).train()
    diff = GaussianMultinomialDiffusion(np.array([0]), d_in, None, num_timesteps=1000, device=torch.device(dev)).to(dev)
    tr = ClassifierTrainer(clf, diff, lr=1e-4)
    for _ in range(5): tr.step(x, y)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n): tr.step(x, y)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"graph={graph}: wall {t_all/n*1e3:.3f} ms/step ({n/t_all:.0f} steps/s), enqueue {t_enq/n*1e3:.3f} ms/step")
    if graph == "0":
        L.clava_prof_enable(ctypes.c_int(8192))
        for _ in range(20): tr.step(x, y)
        torch.cuda.synchronize(); L.clava_prof_disable()
        la, ms, fl, by = (ctypes.c_int * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)()
        L.clava_prof_collect(la, ms, fl, by)
        print(f"    library kernels {sum(ms)/20:.3f} ms/step")
        for i in range(nf):
            if la[i]:
                print(f"    {L.clava_prof_family_name(i).decode():28s} {la[i]/20:5.1f}/step avg {1e3*ms[i]/la[i]:7.1f} us  total {ms[i]/20*1e3:7.1f} us/step"
                      f"  {fl[i]/ms[i]/1e9 if ms[i] else 0:7.1f} TF/s {by[i]/ms[i]/1e6 if ms[i] else 0:7.1f} GB/s")
