"""Training-step timing: wall per step vs summed kernel time from the library's per-launch events (developer tool)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import _lib, split_nets
from clavaddpm_amd.scripts.train import Trainer
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
L = _lib.lib(); L.clava_prof_family_name.restype = ctypes.c_char_p; nf = L.clava_prof_family_count()
dev = "cuda"
for mode in os.environ.get("MODES", "f32,f16x3").split(","):
    split_nets.set_gemm_mode(mode)
    torch.manual_seed(0)
    net = MLPDiffusion(2, 0, "none", {"d_layers": [512, 1024, 1024, 1024, 1024, 512], "dropout": 0.0}).to(dev).train()
    diff = GaussianMultinomialDiffusion(np.array([0]), 2, net, num_timesteps=1000, device=torch.device(dev)).to(dev).train()
    B = int(os.environ.get("B", "4096")); data = torch.randn(B, 2); y = torch.zeros(B, dtype=torch.long)
    if os.environ.get("X_CUDA") == "1": data = data.cuda()
    tr = Trainer(diff, None, lr=6e-4, weight_decay=1e-5, steps=10**9, device=torch.device(dev))
    for _ in range(5): tr._run_step(data, {"y": y})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = int(os.environ.get("STEPS", "40"))
    for _ in range(N): tr._run_step(data, {"y": y})
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    L.clava_prof_enable(ctypes.c_int(4096))
    for _ in range(20): tr._run_step(data, {"y": y})
    torch.cuda.synchronize(); L.clava_prof_disable()
    la, ms, fl, by = (ctypes.c_int * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)()
    L.clava_prof_collect(la, ms, fl, by)
    print(f"{mode}: wall {t_all/N*1e3:.3f} ms/step, enqueue {t_enq/N*1e3:.3f} ms/step, library kernels {sum(ms)/20:.3f} ms/step")
    for i in range(nf):
        if la[i]: print(f"    {L.clava_prof_family_name(i).decode():28s} {la[i]/20:5.1f}/step avg {1e3*ms[i]/la[i]:7.1f} us  total {ms[i]/20*1e3:7.1f} us/step")
