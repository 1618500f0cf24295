import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, math
from conftest import golden
from clavaddpm_amd.tab_ddpm import timestep_embedding
g = golden("g2_mlp.npz")
t = g["t"]
mine = timestep_embedding(t.cuda(), 128).cpu()
d = (mine - g["temb"]).abs()
i = d.argmax().item(); r, c = divmod(i, 128)
half = 64
freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
k = c % 64
arg = t[r].float() * freqs[k]
print("max", d.max().item(), "at", r, c, "t", t[r].item(), "freq", freqs[k].item(), "arg", arg.item())
print("mine", mine[r, c].item(), "ref", g["temb"][r, c].item(), "f64", math.cos(float(arg)) if c < 64 else math.sin(float(arg)))
a = (t[:, None].float() * freqs[None]).cuda()
tg = torch.cat([torch.cos(a), torch.sin(a)], -1).cpu()
print("torch-gpu vs ref", (tg - g["temb"]).abs().max().item(), "mine vs torch-gpu", (tg - mine).abs().max().item())
