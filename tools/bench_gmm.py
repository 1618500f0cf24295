"""GMM E-step kernel timing through the library's per-launch events (developer tool)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clavaddpm_amd import _lib
from clavaddpm_amd.clustering import gmm_estep
L = _lib.lib()
L.clava_prof_family_name.restype = ctypes.c_char_p
nf = L.clava_prof_family_count()
for (N, d, k, resp) in [(1_000_000, 32, 50, True), (1_000_000, 32, 50, False), (1_000_000, 32, 16, True), (1_000_000, 64, 50, True), (1_000_000, 8, 50, True), (4_000_000, 32, 50, True)]:
    rng = np.random.RandomState(0)
    f = lambda a: torch.from_numpy(a).cuda()
    means, pc, w = f(rng.standard_normal((k, d))), f(rng.uniform(0.3, 3.0, (k, d))), f(rng.dirichlet(np.ones(k)))
    X = torch.randn(N, d, device="cuda", dtype=torch.float64) * 2
    for _ in range(3): gmm_estep(X, w, means, pc, want_resp=resp)
    torch.cuda.synchronize()
    L.clava_prof_enable(ctypes.c_int(64))
    for _ in range(10): gmm_estep(X, w, means, pc, want_resp=resp)
    torch.cuda.synchronize()
    L.clava_prof_disable()
    la, ms, fl, by = (ctypes.c_int * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)(), (ctypes.c_double * nf)()
    L.clava_prof_collect(la, ms, fl, by)
    for i in range(nf):
        if la[i]:
            print(f"N={N} d={d} k={k} resp={resp}: {L.clava_prof_family_name(i).decode()} {la[i]} launches avg {1e3*ms[i]/la[i]:.1f} us  "
                  f"{fl[i]/ms[i]/1e9:.2f} TF(f64, unpadded)  {by[i]/ms[i]/1e6:.0f} GB/s", flush=True)
