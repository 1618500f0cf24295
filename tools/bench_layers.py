"""Per-layer timing of one guided reverse step at the bench shapes (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clavaddpm_amd import ops

def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
den = [2, 128, 512, 1024, 1024, 1024, 1024, 512, 2]
cls = [2, 128, 128, 256, 512, 1024, 512, 256, 128, 50]
tot = 0; totf = 0
def run(tag, K, N, fn):
    global tot, totf
    ms = timeit(fn); fl = 2.0 * M * N * K
    tot += ms; totf += fl
    print(f"{tag:10s} K={K:5d} N={N:5d}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF")
for i in range(len(den) - 1):
    K, N = den[i], den[i + 1]
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    run("den fwd", K, N, lambda: ops.linear(x, w, b, act="relu", out=out))
for i in range(len(cls) - 1):
    K, N = cls[i], cls[i + 1]
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    sc = torch.rand(K, device="cuda"); sh = torch.rand(K, device="cuda")
    run("cls fwd", K, N, lambda: ops.linear(x, w, b, act="leaky_relu", out=out, in_scale=sc if i > 1 else None, in_shift=sh if i > 1 else None))
for i in range(len(cls) - 2, -1, -1):
    Kc, Nn = cls[i + 1], cls[i]     # contraction, output width
    dy = torch.randn(M, Kc, device="cuda"); w = torch.randn(Kc, Nn, device="cuda"); a = torch.randn(M, Nn, device="cuda"); out = torch.empty(M, Nn, device="cuda")
    cs = torch.rand(Nn, device="cuda")
    run("cls bwd", Kc, Nn, lambda: ops.linear_dgrad(dy, w, act_out=a if i > 0 else None, slope=0.01, col_scale=cs if i > 0 else None, out=out))
print(f"total {tot*1e3:.1f} us, {totf/tot/1e9:.1f} TF overall")
