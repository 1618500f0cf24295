"""Init-mode categorical draw vs numpy for many narrow features (developer tool)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from clavaddpm_amd._lib import c_int, c_u64, check, lib, ptr, stream_ptr
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
for (m, Kf, b) in [(70, 5, 258), (16, 5, 258), (9, 5, 64), (70, 5, 8), (70, 12, 258), (70, 40, 258)]:
    K = [Kf] * m
    sumK = sum(K)
    net = MLPDiffusion(sumK, 0, "none", {"d_layers": [128, 128], "dropout": 0.0})
    d = GaussianMultinomialDiffusion(np.array(K), 0, net.cuda().eval(), num_timesteps=100, device=torch.device("cuda")).cuda().eval()
    g = torch.Generator().manual_seed(1)
    u = torch.rand(b, sumK, generator=g)
    gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    want = gum.view(b, m, Kf).argmax(-1)
    sched = d._c_schedule()
    outs = []
    for rep in range(2):
        idx = torch.full((b, m), -7, device="cuda", dtype=torch.int32)
        ud = u.cuda().contiguous()
        check(lib().clava_multinomial_step_f32(ctypes.byref(sched), c_int(1), c_int(b), c_int(m), c_int(sumK), ptr(d._offsets_i32), ptr(d._log_k),
                                               None, c_int(0), None, ptr(idx), ptr(ud), c_int(sumK), None, c_int(0), None, c_int(0), None, c_int(0),
                                               c_u64(3), c_u64(0), ctypes.c_uint(0), stream_ptr()), "init")
        torch.cuda.synchronize()
        outs.append(idx.cpu().long())
    bad = (outs[0] != want)
    print(f"m={m} K={Kf} b={b}: unwritten {int((outs[0] == -7).sum())}, wrong {int(bad.sum())} (per feature {bad.sum(0).tolist()[:24]}...), runs differ {int((outs[0] != outs[1]).sum())}")
    if bad.any():
        r, f = [int(v[0]) for v in torch.nonzero(bad, as_tuple=True)]
        print("   first bad: row", r, "feature", f, "got", int(outs[0][r, f]), "want", int(want[r, f]), "gumbel", gum.view(b, m, Kf)[r, f].tolist())
