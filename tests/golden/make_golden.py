#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference; the GPU box has neither
the reference nor this need).  `tab_ddpm` is imported unmodified.  `Classifier`,
the `cond_fn` closure, `Trainer` and `update_ema` live in modules whose top-level
imports need packages absent from this image (icecream, faiss, ...), so their
class/function definitions are compiled straight from the reference source text
with `ast` and executed with the real `torch` / reference `timestep_embedding` in
scope - no stand-in modules, no edited code, nothing copied into this repo.

Every random draw the reference makes is recorded (torch RNG entry points are
wrapped while the reference runs) and stored next to the outputs, so the HIP
path and the oracle can be fed identical noise.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import ast
import os
import sys
from collections import OrderedDict
from copy import deepcopy

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tab_ddpm.gaussian_multinomial_diffsuion import GaussianMultinomialDiffusion  # noqa: E402
from tab_ddpm.modules import MLPDiffusion, timestep_embedding  # noqa: E402
from tab_ddpm.utils import index_to_log_onehot  # noqa: E402
from weights import classifier_shapes, mlp_shapes, seeded_state  # noqa: E402

torch.set_num_threads(1)  # what `import lib` forces in the reference (lib/__init__.py:4)


# --------------------------------------------------------------------------- helpers
def _extract(path, name, inside=None):
    """Return the ast node of class/function `name` (optionally nested in function `inside`)."""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    scope = tree.body
    if inside is not None:
        scope = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == inside).body
    return next(n for n in scope if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == name)


def _compile(node, ns, path):
    mod = ast.Module(body=[node], type_ignores=[])
    exec(compile(mod, os.path.join(REF, path), "exec"), ns)
    return ns[node.name]


def ref_classifier_cls():
    ns = {"nn": nn, "torch": torch, "timestep_embedding": timestep_embedding}
    return _compile(_extract("pipeline_utils.py", "Classifier"), ns, "pipeline_utils.py")


def ref_cond_fn(classifier, classifier_scale):
    ns = {"torch": torch, "F": F, "classifier": classifier, "classifier_scale": classifier_scale}
    node = _extract("pipeline_utils.py", "cond_fn", inside="conditional_sampling_by_group_size")
    return _compile(node, ns, "pipeline_utils.py")


def ref_classifier_train_step(logger):
    """`numerical_forward_backward_log` and the helpers it calls, compiled from pipeline_utils.py:632-700."""
    ns = {"torch": torch, "F": F, "logger": logger}
    for name in ("split_microbatches", "compute_top_k", "log_loss_dict", "numerical_forward_backward_log"):
        _compile(_extract("pipeline_utils.py", name), ns, "pipeline_utils.py")
    return ns["numerical_forward_backward_log"]


def ref_trainer_cls():
    import pandas as pd
    ns = {"torch": torch, "np": np, "pd": pd, "deepcopy": deepcopy}
    _compile(_extract("scripts/utils_train.py", "update_ema"), ns, "scripts/utils_train.py")
    return _compile(_extract("scripts/train.py", "Trainer"), ns, "scripts/train.py")


class Tape:
    """Record every tensor the torch RNG entry points hand to the reference."""
    NAMES = ("randn", "randn_like", "rand_like", "randint", "multinomial")

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._orig = {n: getattr(torch, n) for n in self.NAMES}
        for n, fn in self._orig.items():
            setattr(torch, n, self._wrap(n, fn))
        return self

    def __exit__(self, *a):
        for n, fn in self._orig.items():
            setattr(torch, n, fn)

    def _wrap(self, name, fn):
        def w(*a, **k):
            out = fn(*a, **k)
            self.log.append((name, out.detach().clone()))
            return out
        return w

    def take(self, name):
        kind, v = self.log.pop(0)
        assert kind == name, (kind, name)
        return v


def load_seeded(module, shapes, seed, scale=1.0):
    st = seeded_state(shapes, seed, scale)
    missing, unexpected = module.load_state_dict(st, strict=False)
    assert not unexpected, unexpected
    return st


def P(st):
    return {"p:" + k: v.detach().cpu().numpy() for k, v in st.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def make_diffusion(K, n_num, net, T, sched="cosine"):
    d = GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=T, scheduler=sched)
    d.eval()
    return d


def make_mlp(d_in, d_layers, seed, scale=1.0):
    net = MLPDiffusion(d_in, 0, "none", {"d_layers": list(d_layers), "dropout": 0.0})
    st = load_seeded(net, mlp_shapes(d_in, d_layers), seed, scale)
    net.eval()
    return net, st


# --------------------------------------------------------------------------- G1
def g1_schedule():
    out = {}
    names = ["alphas", "log_alpha", "log_1_min_alpha", "log_1_min_cumprod_alpha", "log_cumprod_alpha",
             "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    for sched in ("cosine", "linear"):
        for T in (50, 1000, 2000):
            d = make_diffusion([0], 3, None, T, sched)
            for n in names:
                out[f"{sched}_{T}_{n}"] = getattr(d, n)
            out[f"{sched}_{T}_posterior_variance_f64"] = d.posterior_variance
            mv = torch.cat([d.posterior_variance[1].unsqueeze(0), (1. - d.alphas)[1:]], dim=0)  # as :209
            out[f"{sched}_{T}_model_variance_f64"] = mv
            out[f"{sched}_{T}_model_log_variance_f64"] = torch.log(mv)
    save("g1_schedule.npz", **out)


# --------------------------------------------------------------------------- G2
def g2_mlp():
    g = torch.Generator().manual_seed(2)
    net, st = make_mlp(14, [64, 64], seed=21)
    x = torch.randn(16, 14, generator=g)
    t = torch.randint(0, 1000, (16,), generator=g)
    with torch.no_grad():
        out = net(x, t)
    big, _ = make_mlp(15, [512, 1024, 1024, 1024, 1024, 512], seed=22)
    xb = torch.randn(8, 15, generator=g)
    tb = torch.randint(0, 1000, (8,), generator=g)
    with torch.no_grad():
        outb = big(xb, tb)
        emb = timestep_embedding(t, 128)
    save("g2_mlp.npz", x=x, t=t, out=out, temb=emb, big_x=xb, big_t=tb, big_out=outb, **P(st))


# --------------------------------------------------------------------------- G3
def g3_gauss_step():
    T = 1000
    d = make_diffusion([0], 6, None, T)
    g = torch.Generator().manual_seed(3)
    t = torch.tensor([999, 500, 1, 0] * 4)
    B = len(t)
    out = {"t": t}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        x = (2.0 * torch.randn(B, 6, generator=g, dtype=torch.float64)).to(dt)
        eps = torch.randn(B, 6, generator=g)
        grad = 0.3 * torch.randn(B, 6, generator=g)
        for gtag, cf in (("plain", None), ("guided", lambda x_, t_, **kw: grad)):
            torch.manual_seed(33)
            with Tape() as tape:
                r = d.gaussian_p_sample(eps, x, t, model_kwargs={}, cond_fn=cf)
            out[f"{tag}_{gtag}_noise"] = tape.take("randn_like")
            out[f"{tag}_{gtag}_sample"] = r["sample"]
            out[f"{tag}_{gtag}_pred_xstart"] = r["pred_xstart"]
        out[f"{tag}_x"], out[f"{tag}_eps"], out[f"{tag}_grad"] = x, eps, grad
    x0 = torch.randn(B, 6, generator=g)
    nz = torch.randn(B, 6, generator=g)
    out["q_x0"], out["q_noise"], out["q_xt"] = x0, nz, d.gaussian_q_sample(x0, t, noise=nz)
    save("g3_gauss_step.npz", **out)


# --------------------------------------------------------------------------- G4
def g4_multinomial_step():
    T = 1000
    g = torch.Generator().manual_seed(4)
    out = {}
    for tag, K in (("small", [3, 5, 2]), ("wide", [2, 7, 64, 65, 200, 1000, 33])):
        d = make_diffusion(K, 0, None, T)
        t = torch.tensor([999, 500, 1, 0] * 4)
        B, S = len(t), int(sum(K))
        logits = 3.0 * torch.randn(B, S, generator=g)
        idx_t = torch.stack([torch.randint(0, k, (B,), generator=g) for k in K], dim=1)
        idx_0 = torch.stack([torch.randint(0, k, (B,), generator=g) for k in K], dim=1)
        log_xt = index_to_log_onehot(idx_t, d.num_classes)
        log_x0 = index_to_log_onehot(idx_0, d.num_classes)
        torch.manual_seed(44)
        with Tape() as tape:
            lp = d.p_pred(logits, log_xt, t, {})
            new = d.log_sample_categorical(lp)
        u = torch.cat([tape.take("rand_like") for _ in K], dim=1)
        with Tape() as tape:
            fwd = d.q_sample(log_x0, t)
        uq = torch.cat([tape.take("rand_like") for _ in K], dim=1)
        pt = torch.ones(B) / T
        out.update({
            f"{tag}_K": np.array(K), f"{tag}_t": t, f"{tag}_logits": logits, f"{tag}_idx_t": idx_t,
            f"{tag}_idx_0": idx_0, f"{tag}_u": u, f"{tag}_log_model_prob": lp,
            f"{tag}_idx_new": torch.stack([new[:, a:b].argmax(1) for a, b in
                                           zip(d.offsets[:-1].tolist(), d.offsets[1:].tolist())], dim=1),
            f"{tag}_uq": uq,
            f"{tag}_idx_q": torch.stack([fwd[:, a:b].argmax(1) for a, b in
                                         zip(d.offsets[:-1].tolist(), d.offsets[1:].tolist())], dim=1),
            f"{tag}_Lt": d.compute_Lt(logits, log_x0, log_xt, t, {}),
            f"{tag}_kl_prior": d.kl_prior(log_x0),
            f"{tag}_vb": d._multinomial_loss(logits, log_x0, log_xt, t, pt, {}),
        })
    save("g4_multinomial_step.npz", **out)


# --------------------------------------------------------------------------- G5
def _tape_chain(tape, T, K, n_num, conditional):
    has_cat = K[0] != 0
    nz = {"z0": tape.take("randn")}
    if has_cat and not conditional:
        nz["u_init"] = torch.cat([tape.take("rand_like") for _ in K], dim=1)
    if not conditional:
        nz["y"] = tape.take("multinomial")
    noise, us = [], []
    for _ in range(T):
        noise.append(tape.take("randn_like").double())
        if has_cat:
            us.append(torch.cat([tape.take("rand_like") for _ in K], dim=1))
    nz["noise"] = torch.stack(noise)
    if has_cat:
        nz["u"] = torch.stack(us)
    assert not tape.log
    return nz


def g5_chains():
    T, B = 50, 64
    K, n_num = [3, 5, 2], 4
    net, st = make_mlp(n_num + sum(K), [64, 64], seed=51, scale=0.5)
    d = make_diffusion(K, n_num, net, T)
    torch.manual_seed(55)
    with Tape() as tape:
        x, _ = d.sample(B, torch.ones(1))
    nz = _tape_chain(tape, T, K, n_num, conditional=False)
    save("g5_chain_mixed.npz", K=np.array(K), n_num=n_num, T=T, sample=x, **{"nz:" + k: v for k, v in nz.items()}, **P(st))

    n_num, C, hidden = 6, 7, [32, 64, 32]
    net, st = make_mlp(n_num, [64, 64], seed=52, scale=0.5)
    clf = ref_classifier_cls()(d_in=n_num, d_out=C, dim_t=128, hidden_sizes=hidden)
    cst = load_seeded(clf, classifier_shapes(n_num, C, hidden), seed=53)
    clf.eval()
    d = make_diffusion([0], n_num, net, T)
    ys = torch.randint(0, C, (B, 1), generator=torch.Generator().manual_seed(5))
    cond_fn = ref_cond_fn(clf, 1.0)
    torch.manual_seed(56)
    with Tape() as tape:
        x, _ = d.conditional_sample(ys=ys, model_kwargs={"y": ys}, cond_fn=cond_fn)
    nz = _tape_chain(tape, T, [0], n_num, conditional=True)
    save("g5_chain_cond.npz", n_num=n_num, T=T, C=C, hidden=np.array(hidden), ys=ys, sample=x,
         **{"nz:" + k: v for k, v in nz.items()}, **P(st), **{"c:" + k: v.numpy() for k, v in cst.items()})


# --------------------------------------------------------------------------- G6
def g6_loss():
    T, B = 1000, 32
    g = torch.Generator().manual_seed(6)
    out = {}
    for tag, K, n_num in (("mixed", [3, 5, 2], 4), ("gauss", [0], 5)):
        m = len(K) if K[0] else 0
        net, st = make_mlp(n_num + (sum(K) if m else 0), [64, 64], seed=61)
        net.train()
        d = make_diffusion(K, n_num, net, T)
        d.train()
        cols = [torch.randn(B, n_num, generator=g)]
        if m:
            cols.append(torch.stack([torch.randint(0, k, (B,), generator=g) for k in K], dim=1).float())
        x = torch.cat(cols, dim=1)
        torch.manual_seed(66)
        with Tape() as tape:
            lm, lg = d.mixed_loss(x, {"y": torch.zeros(B, dtype=torch.long)})
        (lm + lg).backward()
        t = tape.take("randint")
        noise = tape.take("randn_like")
        out[f"{tag}_x"], out[f"{tag}_t"], out[f"{tag}_noise"] = x, t, noise
        if m:
            out[f"{tag}_u"] = torch.cat([tape.take("rand_like") for _ in K], dim=1)
        assert not tape.log
        out[f"{tag}_K"], out[f"{tag}_n_num"] = np.array(K), n_num
        out[f"{tag}_loss_multi"], out[f"{tag}_loss_gauss"] = lm.detach(), lg.detach()
        for k, v in net.named_parameters():
            out[f"{tag}_p:{k}"] = v.detach()
            out[f"{tag}_g:{k}"] = v.grad
    save("g6_loss.npz", **out)


# --------------------------------------------------------------------------- G7
def g7_classifier():
    g = torch.Generator().manual_seed(7)
    Classifier = ref_classifier_cls()
    out = {}
    for tag, d_in, C, hidden, B, store in (("small", 6, 7, [32, 64, 32], 16, True),
                                           ("real", 15, 25, [128, 256, 512, 1024, 512, 256, 128], 8, False)):
        clf = Classifier(d_in=d_in, d_out=C, dim_t=128, hidden_sizes=hidden)
        st = load_seeded(clf, classifier_shapes(d_in, C, hidden), seed=71)
        clf.eval()
        x = torch.randn(B, d_in, generator=g, dtype=torch.float64)
        t = torch.randint(0, 1000, (B,), generator=g)
        y = torch.randint(0, C, (B, 1), generator=g)
        with torch.no_grad():
            logits = clf(x.float(), t)
        grad = ref_cond_fn(clf, 1.5)(x, t, y=y)
        out.update({f"{tag}_x": x, f"{tag}_t": t, f"{tag}_y": y, f"{tag}_logits": logits, f"{tag}_grad": grad,
                    f"{tag}_hidden": np.array(hidden), f"{tag}_C": C})
        if store:
            out.update({f"{tag}_p:{k}": v for k, v in st.items()})
    save("g7_classifier.npz", **out)


# --------------------------------------------------------------------------- G8
def g8_trainer():
    T, B, steps = 1000, 32, 5
    K, n_num = [3, 5, 2], 4
    g = torch.Generator().manual_seed(8)
    net, st = make_mlp(n_num + sum(K), [64, 64], seed=81)
    net.train()
    d = make_diffusion(K, n_num, net, T)
    d.train()
    batches = []
    for _ in range(steps):
        xb = torch.cat([torch.randn(B, n_num, generator=g),
                        torch.stack([torch.randint(0, k, (B,), generator=g) for k in K], dim=1).float()], dim=1)
        batches.append(xb)

    def it():
        for xb in batches:
            yield xb.clone(), torch.zeros(B, dtype=torch.long)

    tr = ref_trainer_cls()(d, it(), lr=6e-4, weight_decay=1e-5, steps=steps, device=torch.device("cpu"))
    tr.log_every = 1
    torch.manual_seed(88)
    with Tape() as tape:
        tr.run_loop()
    out = {"K": np.array(K), "n_num": n_num, "T": T, "steps": steps, "lr": 6e-4, "wd": 1e-5,
           "x": torch.stack(batches), "loss_history": tr.loss_history.to_numpy().astype(np.float64)}
    ts, ns, us = [], [], []
    for _ in range(steps):
        ts.append(tape.take("randint"))
        ns.append(tape.take("randn_like"))
        us.append(torch.cat([tape.take("rand_like") for _ in K], dim=1))
    assert not tape.log
    out["t"], out["noise"], out["u"] = torch.stack(ts), torch.stack(ns), torch.stack(us)
    out.update({"p0:" + k: v for k, v in st.items()})
    out.update({"p1:" + k: v.detach() for k, v in net.named_parameters()})
    out.update({"ema:" + k: v.detach() for k, v in tr.ema_model.named_parameters()})
    save("g8_trainer.npz", **out)


# --------------------------------------------------------------------------- G9
def g9_gmm():
    from sklearn.mixture import GaussianMixture
    rng = np.random.RandomState(9)
    N, k = 600, 6
    num = rng.uniform(0, 1, (N, 5))
    cat = np.eye(4)[rng.randint(0, 4, N)]
    key = 0.01 * rng.uniform(0, 1, (N, 1))
    X = np.concatenate([num + 0.3 * (rng.randint(0, 3, (N, 1))), 2.0 * cat, key], axis=1)
    gmm = GaussianMixture(n_components=k, covariance_type="diag", init_params="k-means++", tol=1e-4, random_state=0)
    gmm.fit(X)
    lpn, lr = gmm._estimate_log_prob_resp(X)
    save("g9_gmm.npz", X=X, weights=gmm.weights_, means=gmm.means_, precisions_chol=gmm.precisions_cholesky_,
         log_prob_norm=lpn, log_resp=lr, labels=gmm.predict(X).astype(np.int64))


# --------------------------------------------------------------------------- G10
def g10_classifier_training():
    """The classifier training loop of `train_classifier` (pipeline_utils.py:783-817) on the reference's own
    `numerical_forward_backward_log`, `Classifier` (train mode), uniform schedule sampler, logger and torch AdamW;
    every draw (timesteps from np.random.choice, q-sample noise, Dropout keep masks) is recorded."""
    import tempfile
    import types
    from tab_ddpm import logger
    from tab_ddpm.resample import create_named_schedule_sampler
    logger.configure(dir=tempfile.mkdtemp(), format_strs=[])
    fwd_bwd = ref_classifier_train_step(logger)
    d_in, C, hidden, B, steps, T, lr = 6, 7, [32, 64, 32], 64, 6, 1000, 1e-3
    g = torch.Generator().manual_seed(10)
    clf = ref_classifier_cls()(d_in=d_in, d_out=C, dim_t=128, hidden_sizes=hidden)
    st = load_seeded(clf, classifier_shapes(d_in, C, hidden), seed=101)
    opt = torch.optim.AdamW(clf.parameters(), lr=lr)                      # :766
    diffusion = GaussianMultinomialDiffusion(num_classes=np.array([0]), num_numerical_features=d_in, denoise_fn=None,
                                             gaussian_loss_type="mse", num_timesteps=T, scheduler="cosine",
                                             device=torch.device("cpu"))       # :768-776
    sampler = create_named_schedule_sampler("uniform", diffusion)
    dataset = types.SimpleNamespace(n_num_features=d_in)
    xs = torch.randn(steps, B, d_in, generator=g)
    ys = torch.randint(0, C, (steps, B), generator=g)
    n_val = (steps + 4) // 5
    vxs = torch.randn(n_val, B, d_in, generator=g)
    vys = torch.randint(0, C, (n_val, B), generator=g)
    train_loader = iter([(xs[i].clone(), ys[i].clone()) for i in range(steps)])
    val_loader = iter([(vxs[i].clone(), vys[i].clone()) for i in range(n_val)])

    masks, ts = [], []
    drops = [m for m in clf.model if isinstance(m, nn.Dropout)]

    def hook(mod, inp, out):
        if mod.training:
            assert not bool((inp[0] == 0).any())
            masks.append((out != 0).to(torch.uint8))
    for m in drops:
        m.register_forward_hook(hook)
    orig_choice = np.random.choice

    def choice(*a, **k):
        r = orig_choice(*a, **k)
        ts.append(torch.from_numpy(np.asarray(r)).long())
        return r
    np.random.choice = choice
    np.random.seed(1010)
    torch.manual_seed(1011)
    logs, grads = [], {}
    live = [k for k, _ in clf.named_parameters() if not k.startswith("transformer_layer")]
    try:
        with Tape() as tape:
            clf.train()                                                      # :783
            for step in range(steps):                                        # :785-817
                if step == steps - 1:       # full state in front of the last step: lets a test check that step alone
                    pd_ = dict(clf.named_parameters())
                    snap = {"ps:" + k: v.detach().clone() for k, v in clf.state_dict().items()
                            if not k.startswith("transformer_layer")}
                    snap.update({"ms:" + k: opt.state[pd_[k]]["exp_avg"].clone() for k in live})
                    snap.update({"vs:" + k: opt.state[pd_[k]]["exp_avg_sq"].clone() for k in live})
                fwd_bwd(clf, opt, train_loader, dataset, sampler, diffusion, prefix="train", device="cpu")
                if step in (0, steps - 1):
                    pd_ = dict(clf.named_parameters())
                    grads[step] = {k: pd_[k].grad.detach().clone() for k in live}
                opt.step()
                if not step % 5:
                    with torch.no_grad():
                        clf.eval()
                        fwd_bwd(clf, opt, val_loader, dataset, sampler, diffusion, prefix="val", device="cpu")
                        clf.train()
                logs.append(dict(logger.getkvs()))
                logger.dumpkvs()
    finally:
        np.random.choice = orig_choice
    noise = [tape.take("randn_like") for _ in range(steps + n_val)]
    assert not tape.log and len(ts) == steps + n_val and len(masks) == steps * len(drops)
    is_val = []
    for step in range(steps):
        is_val.append(0)
        if not step % 5:
            is_val.append(1)
    tr = [i for i, v in enumerate(is_val) if not v]
    va = [i for i, v in enumerate(is_val) if v]
    out = {"d_in": d_in, "C": C, "hidden": np.array(hidden), "T": T, "lr": lr, "steps": steps,
           "x": xs, "y": ys, "val_x": vxs, "val_y": vys,
           "t": torch.stack([ts[i] for i in tr]), "noise": torch.stack([noise[i] for i in tr]),
           "val_t": torch.stack([ts[i] for i in va]), "val_noise": torch.stack([noise[i] for i in va]),
           "keep": torch.stack([torch.cat(masks[s * len(drops):(s + 1) * len(drops)], dim=1) for s in range(steps)]),
           "train_loss": np.array([l["train_loss"] for l in logs]), "train_acc1": np.array([l["train_acc@1"] for l in logs]),
           "train_acc5": np.array([l["train_acc@5"] for l in logs]),
           "val_loss": np.array([l["val_loss"] for l in logs if "val_loss" in l]),
           "val_acc1": np.array([l["val_acc@1"] for l in logs if "val_acc@1" in l])}
    out.update({"p0:" + k: v for k, v in st.items()})
    out.update(snap)
    out.update({"p1:" + k: v.detach() for k, v in clf.state_dict().items() if not k.startswith("transformer_layer")})
    for s_, gd in grads.items():
        out.update({f"g{s_}:" + k: v for k, v in gd.items()})
    save("g10_classifier_training.npz", **out)


# --------------------------------------------------------------------------- G11
def g11_gmm_fit():
    """scikit-learn's own M-step and EM loop (the GaussianMixture the reference builds at pipeline_modules.py:268-277):
    one `_m_step` from the responsibilities of the G9 mixture, and a whole `fit_predict` from recorded k-means++
    initial parameters (weights / means / precisions the M-step of `_initialize` produced)."""
    from sklearn.mixture import GaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky, _estimate_gaussian_parameters
    g9 = np.load(os.path.join(HERE, "g9_gmm.npz"))
    X = g9["X"]
    nk, means, cov = _estimate_gaussian_parameters(X, np.exp(g9["log_resp"]), 1e-6, "diag")
    out = {"m_weights": nk / nk.sum(), "m_means": means, "m_cov": cov, "m_prec_chol": _compute_precision_cholesky(cov, "diag")}
    rng = np.random.RandomState(11)
    N, k = 4000, 9
    centers = rng.uniform(-3, 3, (k, 7))
    lab = rng.randint(0, k, N)
    X2 = centers[lab] + rng.standard_normal((N, 7)) * rng.uniform(0.2, 0.9, (k, 7))[lab]
    X2 = np.concatenate([X2, 2.0 * np.eye(3)[rng.randint(0, 3, N)], 0.01 * rng.uniform(0, 1, (N, 1))], axis=1)
    init = GaussianMixture(n_components=k, covariance_type="diag", init_params="k-means++", tol=1e-4, random_state=3, max_iter=0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        init.fit(X2)
    w0, m0, p0 = init.weights_.copy(), init.means_.copy(), init.precisions_cholesky_.copy()
    # the run itself: same seed, hence the same k-means++ start (w0 sums to k/N: `_initialize` divides nk by n_samples)
    gm = GaussianMixture(n_components=k, covariance_type="diag", init_params="k-means++", tol=1e-4, random_state=3)
    labels = gm.fit_predict(X2)
    out.update({"X": X2, "w0": w0, "m0": m0, "pc0": p0, "weights": gm.weights_, "means": gm.means_, "cov": gm.covariances_,
                "prec_chol": gm.precisions_cholesky_, "n_iter": gm.n_iter_, "lower_bound": gm.lower_bound_,
                "converged": int(gm.converged_), "labels": labels.astype(np.int64)})
    save("g11_gmm_fit.npz", **out)


# --------------------------------------------------------------------------- G12
def g12_postprocess():
    """scikit-learn's `QuantileTransformer.inverse_transform` for the normaliser the reference fits (lib/data.py:217-223)
    and the reference's own `round_columns` (lib/data.py:627-632, compiled from its source)."""
    from scipy.spatial.distance import cdist
    from sklearn.preprocessing import QuantileTransformer
    round_columns = _compile(_extract("lib/data.py", "round_columns"), {"np": np, "cdist": cdist}, "lib/data.py")
    rng = np.random.RandomState(12)
    N = 6000
    real = np.stack([rng.lognormal(0, 1, N), rng.randint(0, 7, N).astype(float), rng.standard_normal(N) * 50 + 10,
                     np.round(rng.exponential(3, N)), rng.choice([1.0, 2.0, 4.0, 8.0], N)], axis=1)
    qt = QuantileTransformer(output_distribution="normal", n_quantiles=max(min(N // 30, 1000), 10), subsample=int(1e9),
                             random_state=0).fit(real)
    x = rng.standard_normal((3000, 5)) * 1.3
    x[:40] = rng.standard_normal((40, 5)) * 6                 # far tails: the 1e-7 clamps
    x[40, :] = [0.0, -0.0, 5.1993375821928165, -5.1993375821928165, 8.3]
    x[41, 2] = np.nan
    x[42, :] = qt.transform(real[:1])[0]                       # lands on a quantile knot
    inv = qt.inverse_transform(x.copy())
    cols = [1, 3, 4]
    rounded = round_columns(real, inv.copy(), cols)
    save("g12_postprocess.npz", quantiles=qt.quantiles_, references=qt.references_, x=x, inverse=inv, real=real,
         cols=np.array(cols), rounded=rounded)


def g13_group_vote():
    """The reference's own voting statements (pipeline_modules.py:297-344, the non-variational branch and the group-size
    tables that follow it), compiled from its source text and run on synthetic labels / group lengths."""
    from collections import defaultdict
    path = "pipeline_modules.py"
    tree = ast.parse(open(os.path.join(REF, path)).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "pair_clustering_keep_id")
    idx = next(i for i, n in enumerate(fn.body) if isinstance(n, ast.If) and isinstance(n.test, ast.Compare)
               and getattr(n.test.left, "id", "") == "clustering_method" and any(isinstance(m, ast.For) for m in n.orelse))
    stmts = list(fn.body[idx].orelse)
    for n in fn.body[idx + 1:]:
        if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "child_df_with_cluster":
            break
        stmts.append(n)
    freq_to_prob = _compile(_extract("pipeline_utils.py", "freq_to_prob"), {}, "pipeline_utils.py")
    code = compile(ast.Module(body=stmts, type_ignores=[]), os.path.join(REF, path), "exec")
    rng = np.random.RandomState(13)
    out = {}
    for tag, G, k, mean_len in (("a", 400, 7, 2.5), ("b", 150, 50, 40.0), ("c", 60, 3, 1.2)):
        lengths = rng.poisson(mean_len, G) + 1
        if tag == "b":
            lengths[::17] = 300 + rng.randint(0, 50, len(lengths[::17]))      # a few long groups
        N = int(lengths.sum())
        home = np.repeat(rng.randint(0, k, G), lengths)
        labels = np.where(rng.rand(N) < 0.6, home, rng.randint(0, k, N)).astype(np.int64)   # ties and split votes happen
        ns = {"np": np, "defaultdict": defaultdict, "freq_to_prob": freq_to_prob, "cluster_labels": labels,
              "child_group_lengths": lengths.astype(int), "sorted_child_data": np.zeros((N, 1)), "print": lambda *a, **k: None}
        exec(code, ns)
        prob = ns["group_lengths_prob_dicts"]
        flat = np.array([[lab, ln, p] for lab, d in prob.items() for ln, p in d.items()], dtype=np.float64)
        out.update({f"{tag}_labels": labels, f"{tag}_lengths": lengths.astype(np.int64),
                    f"{tag}_group_labels": np.array(ns["group_cluster_labels"], dtype=np.int64),
                    f"{tag}_agree": np.array(ns["agree_rates"], dtype=np.float64),
                    f"{tag}_assignment": np.asarray(ns["group_assignment"], dtype=np.int64), f"{tag}_prob": flat,
                    f"{tag}_mean_agree": np.array(ns["average_agree_rate"])})
    save("g13_group_vote.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g13", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12"]
    fns = {"g1": g1_schedule, "g2": g2_mlp, "g3": g3_gauss_step, "g4": g4_multinomial_step, "g5": g5_chains,
           "g6": g6_loss, "g7": g7_classifier, "g8": g8_trainer, "g9": g9_gmm, "g10": g10_classifier_training, "g11": g11_gmm_fit, "g12": g12_postprocess, "g13": g13_group_vote}
    for w in which:
        fns[w]()
