"""Debug: d(loss)/d(model_out) of the fused loss kernels vs the oracle's autograd at the scale-sweep shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from weights import mlp_shapes, seeded_state
from oracle import schedule, gaussian, multinomial as mn
from oracle.network import mlp_diffusion
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion
from clavaddpm_amd import training as tr

rng = np.random.RandomState(0)
K = [int(v) for v in np.exp(rng.uniform(np.log(2), np.log(1000), 50)).round().clip(2, 1000)]
DL = [512, 1024, 1024, 1024, 1024, 512]
n_num, sumK, T, B = 100, sum(K), 1000, 256
D = n_num + sumK
state = seeded_state(mlp_shapes(D, DL), 31, scale=0.5)
g = torch.Generator().manual_seed(17)
x = torch.cat([torch.randn(B, n_num, generator=g)] + [torch.randint(0, k, (B, 1), generator=g).float() for k in K], dim=1)
t = torch.randint(0, T, (B,), generator=g); t[:5] = 0
noise = torch.randn(B, n_num, generator=g); u = torch.rand(B, sumK, generator=g)
tb = schedule.tables("cosine", T)
pt = torch.ones_like(t).float() / T
x_num, x_cat = x[:, :n_num], x[:, n_num:]
x_num_t = gaussian.q_sample(tb, x_num, t, noise)
log_cat = mn.index_to_log_onehot(x_cat.long(), K)
idx_t = mn.sample_categorical(mn.q_pred(tb, log_cat, t, K), u, K)
log_cat_t = mn.index_to_log_onehot(idx_t, K)
out = mlp_diffusion(state, torch.cat([x_num_t, log_cat_t], dim=1), t, 128).detach().requires_grad_(True)
lm = mn.multinomial_loss(tb, out[:, n_num:], log_cat, log_cat_t, t, pt, K, T, True) / len(K)
lg = ((noise - out[:, :n_num]) ** 2).mean(dim=1)
(lm.mean() + lg.mean()).backward()
ref = out.grad

d = GaussianMultinomialDiffusion(np.array(K), n_num, None, num_timesteps=T, device=torch.device("cuda")).to("cuda")
dev = "cuda"
idx0 = x_cat.to(torch.int32).to(dev).contiguous()
idxt = idx_t.to(torch.int32).to(dev).contiguous()
l_m, l_g, d_out = tr.loss_and_grad(d, out.detach().to(dev).contiguous(), noise.to(dev).contiguous(), idx0, idxt, t.to(dev))
d_out = d_out.cpu()
print("loss multi", float(l_m), float(lm.mean()), "gauss", float(l_g), float(lg.mean()))
off = mn.offsets(K)
err = (d_out - ref).abs()
print("numeric part max err / max ref", float(err[:, :n_num].max() / ref[:, :n_num].abs().max()))
worst = []
for f in range(len(K)):
    e = err[:, n_num + off[f]:n_num + off[f + 1]]
    r = ref[:, n_num + off[f]:n_num + off[f + 1]]
    worst.append((float(e.max() / ref[:, n_num:].abs().max()), K[f], f, int(e.max(dim=1).values.argmax())))
worst.sort(reverse=True)
for w in worst[:12]:
    row = w[3]
    print("feature %d K=%d err/maxref %.3e worst row %d t=%d idx0=%d idxt=%d" % (w[2], w[1], w[0], row, int(t[row]), int(x_cat[row, w[2]]), int(idx_t[row, w[2]])))
print("per-t0 rows err", float(err[:5].max()), "other rows", float(err[5:].max()), "max ref", float(ref.abs().max()))

# ---- detail of the worst elements, against a float64 run of the oracle loss
tb2 = {k: torch.as_tensor(v).double() for k, v in tb.items()}
o64 = out.detach().clone().double().requires_grad_(True)
lm64 = mn.multinomial_loss(tb2, o64[:, n_num:], log_cat.double(), log_cat_t.double(), t, pt.double(), K, T, True) / len(K)
lm64.mean().backward()
r64 = o64.grad
e64 = (d_out.double() - r64).abs()
print("HIP vs f64 oracle: max err / max", float(e64[:, n_num:].max() / r64[:, n_num:].abs().max()),
      " f32 oracle vs f64:", float((ref.double() - r64)[:, n_num:].abs().max() / r64[:, n_num:].abs().max()))
flat = e64[:, n_num:].flatten().topk(8)
for v, i in zip(flat.values.tolist(), flat.indices.tolist()):
    row, col = divmod(i, sumK)
    f = int(np.searchsorted(off, col, side="right") - 1)
    c = col - off[f]
    print("row %d t=%d feature %d K=%d class %d (hot0=%d hott=%d): hip %.6e f32 %.6e f64 %.6e logit %.4f" % (
        row, int(t[row]), f, K[f], c, int(x_cat[row, f]), int(idx_t[row, f]), float(d_out[row, n_num + col]), float(ref[row, n_num + col]),
        float(r64[row, n_num + col]), float(out[row, n_num + col])))

# ---- what torch's own float32 device ops (same hardware exp2 / log2 underneath) give for the same loss
_orig = torch.from_numpy
torch.from_numpy = lambda a: _orig(a).to(dev)
tbd = {k: torch.as_tensor(v).to(dev) for k, v in tb.items()}
og = out.detach().clone().to(dev).requires_grad_(True)
lmg = mn.multinomial_loss(tbd, og[:, n_num:], log_cat.to(dev), log_cat_t.to(dev), t.to(dev), pt.to(dev), K, T, True) / len(K)
lmg.mean().backward()
torch.from_numpy = _orig
eg_ = (og.grad.cpu().double() - r64).abs()
print("torch float32 ON THE GPU vs f64 oracle: max err / max", float(eg_[:, n_num:].max() / r64[:, n_num:].abs().max()))
