"""`GaussianMultinomialDiffusion` with the reference's constructor, attributes and method contracts, running its
hot loops on the gfx950 HIP library.

Mirrors `tab_ddpm/gaussian_multinomial_diffsuion.py` of the reference (module name kept, typo included, so that
`from tab_ddpm.gaussian_multinomial_diffsuion import ...` call sites only change their package prefix):
constructor and buffers (`:63-159`), `gaussian_q_sample` (`:172-180`), `mixed_loss` (`:636-675`), `sample`
(`:1026-1070`), `conditional_sample` (`:987-1022`), `sample_all` (`:1072-1099`).

What differs by design:
  * the T-step reverse loop, denoiser, guidance and both fused step kernels run inside ONE C call per batch
    (`clava_sample_chain_f32`); the per-step `print`, the `torch.cat`/`torch.log` rebuild of the variance table and
    the per-step device->host sync of the reference are gone;
  * categorical state is carried as int32 indices [B, m] instead of a [B, sum K] log-one-hot;
  * noise comes from a counter-based generator keyed by (seed, step, global row, column) unless a `noise` dict is
    injected (tests), so samples do not depend on how rows are split over GPUs;
  * the numeric chain state stays float64 like the reference's silently-promoted chain.
There is no CPU path: the module must be on a GPU to compute.
"""
import ctypes

import numpy as np
import torch

from .. import _lib, _native, split_nets
from .._lib import c_int, c_u64, check, lib, ptr, require_cuda, stream_ptr
from . import schedule as _schedule
from .modules import run_stack, time_embed_forward, timestep_embedding
from .schedule import betas_for_alpha_bar, get_named_beta_schedule  # noqa: F401  (public names of the reference module)
from .utils import FoundNANsError

eps = 1e-8

_F32_BUFFERS = ["alphas", "log_alpha", "log_1_min_alpha", "log_1_min_cumprod_alpha", "log_cumprod_alpha",
                "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
                "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"]


def _fresh_seed():
    """64-bit seed drawn from torch's default CPU generator, so torch.manual_seed() controls sampling."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class GaussianMultinomialDiffusion(torch.nn.Module):
    def __init__(self, num_classes: np.array, num_numerical_features: int, denoise_fn, num_timesteps=1000,
                 gaussian_loss_type='mse', gaussian_parametrization='eps', multinomial_loss_type='vb_stochastic',
                 parametrization='x0', scheduler='cosine', device=torch.device('cpu')):
        super().__init__()
        assert multinomial_loss_type in ('vb_stochastic', 'vb_all')
        assert parametrization in ('x0', 'direct')
        num_classes = np.asarray(num_classes)
        self.num_numerical_features = num_numerical_features
        self.num_classes = num_classes                      # [K_1..K_m]; np.array([0]) means "no categoricals"
        self.num_classes_expanded = torch.from_numpy(
            np.concatenate([num_classes[i].repeat(num_classes[i]) for i in range(len(num_classes))])).to(device)
        edges = np.append([0], np.cumsum(num_classes))
        self.slices_for_classes = [np.arange(edges[i], edges[i + 1]) for i in range(len(num_classes))]
        self.offsets = torch.from_numpy(edges).to(device)
        self._denoise_fn = denoise_fn
        self.gaussian_loss_type = gaussian_loss_type
        self.gaussian_parametrization = gaussian_parametrization
        self.multinomial_loss_type = multinomial_loss_type
        self.num_timesteps = num_timesteps
        self.parametrization = parametrization
        self.scheduler = scheduler

        f32, f64 = _schedule.build_tables(scheduler, num_timesteps)
        # left as a float64 CPU attribute by the reference too (:125-127)
        self.posterior_variance = torch.from_numpy(f64["posterior_variance"])
        self.posterior_log_variance_clipped = torch.from_numpy(f32["posterior_log_variance_clipped"]).to(device)
        self.posterior_mean_coef1 = torch.from_numpy(f32["posterior_mean_coef1"]).to(device)
        self.posterior_mean_coef2 = torch.from_numpy(f32["posterior_mean_coef2"]).to(device)
        for name in _F32_BUFFERS:
            self.register_buffer(name, torch.from_numpy(f32[name]).to(device))
        self.register_buffer('Lt_history', torch.zeros(num_timesteps))
        self.register_buffer('Lt_count', torch.zeros(num_timesteps))
        # extra device tables of the HIP path; non-persistent so the state_dict stays the reference's
        self.register_buffer('_coef1', torch.from_numpy(f32["posterior_mean_coef1"]).to(device), persistent=False)
        self.register_buffer('_coef2', torch.from_numpy(f32["posterior_mean_coef2"]).to(device), persistent=False)
        self.register_buffer('_model_variance', torch.from_numpy(f64["model_variance"]).to(device), persistent=False)
        self.register_buffer('_model_sigma', torch.from_numpy(f64["model_sigma"]).to(device), persistent=False)
        self.register_buffer('_offsets_i32', torch.from_numpy(edges.astype(np.int32)).to(device), persistent=False)
        ks = torch.from_numpy(num_classes.astype(np.int64))
        self.register_buffer('_log_k', torch.log(ks).float().to(device), persistent=False)   # torch.log(int64) -> float32 (:398)

    # ------------------------------------------------------------------ helpers
    @property
    def has_cat(self):
        return int(self.num_classes[0]) != 0

    @property
    def _m(self):
        return len(self.num_classes) if self.has_cat else 0

    @property
    def _sum_k(self):
        return int(self.num_classes.sum()) if self.has_cat else 0

    def _device(self):
        return self.log_alpha.device

    def _c_schedule(self):
        tabs = {"sqrt_recip_alphas_cumprod": self.sqrt_recip_alphas_cumprod,
                "sqrt_recipm1_alphas_cumprod": self.sqrt_recipm1_alphas_cumprod,
                "posterior_mean_coef1": self._coef1, "posterior_mean_coef2": self._coef2,
                "model_variance": self._model_variance, "model_sigma": self._model_sigma,
                "log_alpha": self.log_alpha, "log_1_min_alpha": self.log_1_min_alpha,
                "log_cumprod_alpha": self.log_cumprod_alpha, "log_1_min_cumprod_alpha": self.log_1_min_cumprod_alpha,
                "sqrt_alphas_cumprod": self.sqrt_alphas_cumprod,
                "sqrt_one_minus_alphas_cumprod": self.sqrt_one_minus_alphas_cumprod}
        require_cuda(self.log_alpha)
        return _native.make_schedule(self.num_timesteps, tabs)

    def _emb_table(self, time_embed, dim_t):
        """time_embed(timestep_embedding(t)) for every t in [0, T): the batch-constant term of `proj(x) + emb`."""
        t = torch.arange(self.num_timesteps, device=self._device(), dtype=torch.long)
        return time_embed_forward(time_embed, t, dim_t).contiguous()

    # ------------------------------------------------------------------ Gaussian forward process
    def gaussian_q_sample(self, x_start, t, noise=None):
        """sqrt(abar_t) * x_0 + sqrt(1 - abar_t) * noise (`:172-180`)."""
        require_cuda(x_start, t)
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        x0 = x_start.float().contiguous()
        nz = noise.to(x0.device).float().contiguous()
        tt = t.to(x0.device).long().contiguous()
        out = torch.empty_like(x0)
        B, n = x0.shape
        if B * n:
            sched = self._c_schedule()
            check(lib().clava_gauss_qsample_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(x0), ptr(nz), ptr(tt), ptr(out),
                                                c_int(n), stream_ptr()), "clava_gauss_qsample_f32")
        return out

    # ------------------------------------------------------------------ single reverse / forward steps
    @torch.no_grad()
    def gaussian_p_sample(self, model_out, x, t, clip_denoised=False, denoised_fn=None, model_kwargs=None, cond_fn=None,
                          *, noise=None, seed=0, step_index=0):
        """One fused Gaussian reverse step with per-row `t` (`:354-383`); returns {'sample', 'pred_xstart'} (float64).
        `cond_fn(x, t, **model_kwargs)` is evaluated first, as `condition_mean` does (`:315-328`)."""
        require_cuda(model_out, x, t)
        B, n = x.shape
        grad = None
        if cond_fn is not None:
            grad = cond_fn(x, t, **(model_kwargs or {})).float().contiguous()
        xs = x.double().clone().contiguous()
        mo = model_out.float().contiguous()
        pred = torch.empty_like(xs)
        nz = noise.to(x.device).double().contiguous() if noise is not None else None
        sched = self._c_schedule()
        check(lib().clava_gauss_step_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(mo), c_int(mo.stride(0)), ptr(xs), ptr(grad),
                                         c_int(n), ptr(nz), None, c_int(0), ptr(pred), ptr(t.long().contiguous()), c_int(0),
                                         c_int(1 if x.dtype == torch.float32 else 0), c_u64(seed), c_u64(0),
                                         ctypes.c_uint(step_index), stream_ptr()), "clava_gauss_step_f32")
        return {"sample": xs, "pred_xstart": pred}

    @torch.no_grad()
    def _categorical_step(self, mode, logits, idx_in, t, uniforms, want_log_prob, seed=0, step_index=0):
        dev = self._device()
        B = (idx_in if idx_in is not None else uniforms).shape[0]
        m, sk = self._m, self._sum_k
        out = torch.empty((B, m), device=dev, dtype=torch.int32)
        lp = torch.empty((B, sk), device=dev, dtype=torch.float32) if want_log_prob else None
        lg = logits.float().contiguous() if logits is not None else None
        ii = idx_in.to(dev).to(torch.int32).contiguous() if idx_in is not None else None
        u = uniforms.to(dev).float().contiguous() if uniforms is not None else None
        tt = t.to(dev).long().contiguous() if t is not None else None
        sched = self._c_schedule()
        check(lib().clava_multinomial_step_f32(ctypes.byref(sched), c_int(mode), c_int(B), c_int(m), c_int(sk),
                                               ptr(self._offsets_i32), ptr(self._log_k), ptr(lg), c_int(lg.stride(0) if lg is not None else 0),
                                               ptr(ii), ptr(out), ptr(u), c_int(sk), None, c_int(0), ptr(lp), c_int(sk), ptr(tt), c_int(0),
                                               c_u64(seed), c_u64(0), ctypes.c_uint(step_index), stream_ptr()),
              "clava_multinomial_step_f32")
        return out, lp

    def p_sample_index(self, model_out_cat, idx_t, t, uniforms=None, return_log_prob=False):
        """`p_sample` (`:468-472`) on index state: x_{t-1} indices (and optionally log p(x_{t-1} | x_t), `p_pred` `:457-466`)."""
        return self._categorical_step(0, model_out_cat, idx_t, t, uniforms, return_log_prob)

    def q_sample_index(self, idx_0, t, uniforms=None):
        """`q_sample` (`:518-523`) on index state."""
        return self._categorical_step(2, None, idx_0, t, uniforms, False)[0]

    # ------------------------------------------------------------------ training loss
    def mixed_loss(self, x, out_dict, *, draws=None):
        """(loss_multi.mean(), loss_gauss.mean()), differentiable w.r.t. the denoiser (`:636-675`).
        `draws` = dict(t, noise, u) injects the randomness (tests); otherwise it is drawn on the device."""
        from ..training import mixed_loss as _mixed_loss
        return _mixed_loss(self, x, out_dict, draws)

    # ------------------------------------------------------------------ sampling
    def _denoiser_stack(self):
        fn = self._denoise_fn
        if not hasattr(fn, "stack") or not hasattr(fn, "time_embed"):
            raise _lib.ClavaHipError("the fused chain needs a clavaddpm_amd MLPDiffusion as denoise_fn")
        if getattr(fn, "is_y_cond", "none") == "embedding":
            raise NotImplementedError("is_y_cond='embedding' is not wired into the fused chain (ClavaDDPM uses 'none')")
        ws, bs, acts = fn.stack()
        return [w.detach() for w in ws], [b.detach() for b in bs], acts

    def _run_chain(self, x, idx, guidance, ys, noise, seed, row_offset, n_steps=None, t_start=None):
        """x float64 [B, n_num] and idx int32 [B, m] are updated in place by clava_sample_chain_f32.
        `t_start` (default T-1) resumes a chain at a later timestep: x / idx are then the state at t_start, and a float64
        `noise["z0"]` marks a state that is already past the reference's float32 first step."""
        dev = self._device()
        B = x.shape[0]
        T = self.num_timesteps
        t_start = T - 1 if t_start is None else int(t_start)
        if not 0 <= t_start < T:
            raise ValueError(f"t_start {t_start} outside [0, {T})")
        n_steps = t_start + 1 if n_steps is None else int(n_steps)
        if n_steps > t_start + 1:
            raise ValueError(f"{n_steps} steps from t = {t_start} run past t = 0")
        ws, bs, acts = self._denoiser_stack()
        den, keep_d = _native.make_net(ws, bs, acts)
        den_emb = self._emb_table(self._denoise_fn.time_embed, self._denoise_fn.dim_t)
        c = _native.CChain()
        c.sched = self._c_schedule()
        c.t_start = t_start
        c.n_steps = n_steps
        c.n_num, c.m, c.sumK = self.num_numerical_features, self._m, self._sum_k
        c.offsets = self._offsets_i32.data_ptr() if self.has_cat else None
        c.log_k = self._log_k.data_ptr() if self.has_cat else None
        c.denoiser = ctypes.pointer(den)
        c.denoiser_emb = den_emb.data_ptr()
        keep = [keep_d, den_emb]
        cls_emb = None
        if guidance is not None:
            clf = guidance.classifier
            if clf.training:
                raise _lib.ClavaHipError("guidance needs classifier.eval()")
            cw, cb, cacts, sc, sh = clf.stack()
            cnet, keep_c = _native.make_net([w.detach() for w in cw], [b.detach() for b in cb], cacts, sc, sh)
            cls_emb = self._emb_table(clf.time_embed, clf.dim_t)
            y32 = ys.reshape(-1).to(dev).to(torch.int32).contiguous()
            c.classifier = ctypes.pointer(cnet)
            c.classifier_emb = cls_emb.data_ptr()
            c.y = y32.data_ptr()
            c.guidance_scale = guidance.classifier_scale
            keep += [keep_c, cls_emb, y32, cnet]
        c.seed, c.row_offset = int(seed), int(row_offset)
        c.first_step_f32 = 0 if (noise is not None and noise["z0"].dtype == torch.float64) else 1
        if noise is not None:
            nz = noise["noise"][:n_steps].to(dev).double().contiguous()
            c.noise = nz.data_ptr()
            keep.append(nz)
            if self.has_cat:
                u = noise["u"][:n_steps].to(dev).float().contiguous()
                c.uniforms = u.data_ptr()
                keep.append(u)
        if split_nets.get_gemm_mode() == "f16x3":
            done = self._run_chain_f16x3(c, x, idx, ws, bs, acts, den_emb, guidance, cls_emb, keep)
            if done:
                return keep
        need = lib().clava_chain_workspace_floats
        need.restype = ctypes.c_longlong
        n_ws = need(ctypes.byref(c), c_int(B))
        if n_ws < 0:
            check(-1, "clava_chain_workspace_floats")
        wsbuf = torch.empty(int(n_ws), device=dev, dtype=torch.float32)
        check(lib().clava_sample_chain_f32(ctypes.byref(c), c_int(B), ptr(x), ptr(idx) if idx is not None else None, ptr(wsbuf),
                                           ctypes.c_longlong(int(n_ws)), stream_ptr()), "clava_sample_chain_f32")
        keep.append(wsbuf)
        return keep  # caller keeps these alive until it synchronises

    def _run_chain_f16x3(self, c, x, idx, ws, bs, acts, den_emb, guidance, cls_emb, keep):
        """The same chain with every Linear on the f16 matrix cores (split-f16, float32-level accuracy: csrc/split.h).
        Returns False - with x / idx restored - when an intermediate left the float16 range, so that the caller reruns
        the chain on the float32 kernels."""
        dev = self._device()
        B = x.shape[0]
        offs = self._offsets_i32.cpu().tolist() if self.has_cat else None
        x_cal, cal_max = split_nets.calibration_input(self.num_numerical_features, offs, dev)
        shift_in = split_nets.shift_for(cal_max, split_nets.IN_TARGET_LOG2)
        den = split_nets.prepare(ws, bs, acts, None, None, den_emb, x_cal, shift_in, watch=list(self._denoise_fn.parameters()),
                                 cat=(self.num_numerical_features, self._sum_k) if self.has_cat else None)
        cls = None
        if guidance is not None:
            clf = guidance.classifier
            cw, cb, cacts, sc, sh = clf.stack()
            n_cls = cw[-1].shape[0]
            y_cal = (torch.arange(split_nets.CAL_ROWS, device=dev) % n_cls).to(torch.int32)
            cls = split_nets.prepare([w.detach() for w in cw], [b.detach() for b in cb], cacts, sc, sh, cls_emb, x_cal, shift_in,
                                     y_cal=y_cal, need_grad=True, watch=list(clf.parameters()) + list(clf.buffers()))
        x0 = x.clone()
        idx0 = idx.clone() if idx is not None else None
        flag = torch.zeros(1, device=dev, dtype=torch.int32)
        L = lib()
        L.clava_chain_workspace_bytes_f16x3.restype = ctypes.c_longlong
        cp = ctypes.byref(cls.c) if cls is not None else None
        n_ws = L.clava_chain_workspace_bytes_f16x3(ctypes.byref(c), ctypes.byref(den.c), cp, c_int(B))
        if n_ws < 0:
            check(-1, "clava_chain_workspace_bytes_f16x3")
        wsbuf = torch.zeros(int(n_ws) + 256, device=dev, dtype=torch.uint8)
        base = wsbuf.data_ptr()
        aligned = (base + 255) // 256 * 256
        check(L.clava_sample_chain_f16x3(ctypes.byref(c), ctypes.byref(den.c), cp, c_int(B), ptr(x),
                                         ptr(idx) if idx is not None else None, ctypes.c_void_p(aligned),
                                         ctypes.c_longlong(int(n_ws)), ptr(flag), stream_ptr()), "clava_sample_chain_f16x3")
        keep += [den, cls, wsbuf, flag]
        if int(flag.item()) != 0:       # synchronises; rare: a magnitude 64x beyond the calibrated range
            import warnings
            split_nets.fallback_count += 1
            warnings.warn("split-f16 chain left the float16 range; rerunning this batch on the float32 MFMA kernels")
            x.copy_(x0)
            if idx is not None:
                idx.copy_(idx0)
            return False
        return True

    def _initial_state(self, b, noise, seed, row_offset):
        require_cuda(self.log_alpha)          # no CPU path: fail before anything is computed
        dev = self._device()
        n = self.num_numerical_features
        if noise is not None:
            x = noise["z0"].to(dev).double().contiguous()
        else:
            x = torch.empty((b, n), device=dev, dtype=torch.float64)
            if b * n:
                check(lib().clava_gauss_init_f64(ptr(x), c_int(b), c_int(n), c_u64(seed), c_u64(row_offset), stream_ptr()),
                      "clava_gauss_init_f64")
        idx = None
        if self.has_cat:
            idx = torch.empty((b, self._m), device=dev, dtype=torch.int32)
            u = noise["u_init"].to(dev).float().contiguous() if noise is not None else None
            sched = self._c_schedule()
            check(lib().clava_multinomial_step_f32(ctypes.byref(sched), c_int(1), c_int(b), c_int(self._m), c_int(self._sum_k),
                                                   ptr(self._offsets_i32), ptr(self._log_k), None, c_int(0), None, ptr(idx),
                                                   ptr(u), c_int(self._sum_k), None, c_int(0), None, c_int(0), None, c_int(0),
                                                   c_u64(seed), c_u64(row_offset), ctypes.c_uint(0), stream_ptr()),
                  "clava_multinomial_step_f32(init)")
        return x, idx

    def _finish(self, x, idx):
        cols = [x]
        if self.has_cat:
            cols.append(idx.to(torch.float64))
        return torch.cat(cols, dim=1).cpu()

    @torch.no_grad()
    def conditional_sample(self, ys, model_kwargs=None, cond_fn=None, *, noise=None, seed=None, row_offset=0, n_steps=None, t_start=None):
        """Cluster-guided chain for labels `ys` [b, 1] (`:987-1022`); returns (sample float64 CPU [b, n_num], out_dict)."""
        dev = self._device()
        b = len(ys)
        if self.has_cat:
            # the reference never initialises the categorical state here (:987-993), so its denoiser input has the
            # wrong width with categories; refuse instead of mis-sampling
            raise NotImplementedError("conditional_sample is Gaussian-only in ClavaDDPM (num_classes must be [0])")
        out_dict = {'y': ys.long().to(dev)}
        seed = _fresh_seed() if seed is None else seed
        x, idx = self._initial_state(b, noise, seed, row_offset)
        if b == 0:
            return self._finish(x, idx), out_dict
        from ..pipeline_utils import ClassifierGuidance
        if cond_fn is None or isinstance(cond_fn, ClassifierGuidance):
            labels = self._guidance_labels(model_kwargs, b) if cond_fn is not None else out_dict['y']
            keep = self._run_chain(x, idx, cond_fn, labels, noise, seed, row_offset, n_steps, t_start)
        else:
            if t_start is not None:
                raise NotImplementedError("t_start with a Python cond_fn")
            keep = self._run_chain_python_guidance(x, cond_fn, model_kwargs or {}, noise, seed, row_offset, n_steps)
        sample = self._finish(x, idx)   # .cpu() synchronises; `keep` may go now
        del keep
        return sample, out_dict

    @torch.no_grad()
    def sample(self, num_samples, y_dist, model_kwargs=None, cond_fn=None, *, noise=None, seed=None, row_offset=0, n_steps=None, t_start=None,
               idx_start=None):
        """Unconditional chain (`:1026-1070`); returns (sample float64 CPU [b, n_num + m], {'y': labels})."""
        dev = self._device()
        b = num_samples
        seed = _fresh_seed() if seed is None else seed
        x, idx = self._initial_state(b, noise, seed, row_offset)
        if idx_start is not None:                 # resumed chain (tests): the categorical state at t_start
            idx.copy_(torch.as_tensor(idx_start).to(dev).to(torch.int32))
        if noise is not None and "y" in noise:
            y = noise["y"]
        else:
            y = torch.multinomial(y_dist, num_samples=b, replacement=True)   # label draw of :1037-1041 (unused by the MLP)
        out_dict = {'y': y.long().to(dev)}
        from ..pipeline_utils import ClassifierGuidance
        if cond_fn is None or isinstance(cond_fn, ClassifierGuidance):
            if cond_fn is not None and self.has_cat:
                raise NotImplementedError("guidance is Gaussian-only")
            # the guidance target is model_kwargs['y'] (`cond_fn(x, t, **model_kwargs)`, :315-328), NOT the label drawn
            # from y_dist above: `conditional_sampling` (pipeline_utils.py:1001-1012) passes its own `classes` there and
            # records them as the rows' clusters
            labels = self._guidance_labels(model_kwargs, b) if cond_fn is not None else out_dict['y']
            keep = self._run_chain(x, idx, cond_fn, labels, noise, seed, row_offset, n_steps, t_start)
        else:
            if self.has_cat:
                raise NotImplementedError("a Python cond_fn with categorical columns is not supported")
            keep = self._run_chain_python_guidance(x, cond_fn, model_kwargs or {}, noise, seed, row_offset, n_steps)
        sample = self._finish(x, idx)
        del keep
        return sample, out_dict

    def _guidance_labels(self, model_kwargs, b):
        """The fused chain's guidance target: `model_kwargs['y']`, exactly what the reference hands to its closure
        (`cond_fn(x, t, **model_kwargs)`, `:315-328`; the closure asserts `y is not None`, pipeline_utils.py:855)."""
        kw = model_kwargs or {}
        extra = set(kw) - {"y", "remove_first_col"}
        if extra:
            raise TypeError(f"cond_fn() got an unexpected keyword argument {sorted(extra)[0]!r}")
        if kw.get("remove_first_col", False):
            raise NotImplementedError("remove_first_col=True is not wired into the fused guided chain (ClavaDDPM never "
                                      "passes it); wrap the guidance in a plain Python cond_fn to use it")
        y = kw.get("y", None)
        assert y is not None                                  # the reference closure's own assertion
        y = torch.as_tensor(y)
        if y.numel() != b:
            raise ValueError(f"model_kwargs['y'] holds {y.numel()} labels for {b} rows")
        return y.reshape(-1).long().to(self._device())

    def _run_chain_python_guidance(self, x, cond_fn, model_kwargs, noise, seed, row_offset, n_steps=None):
        """Per-step loop for an arbitrary Python `cond_fn(x, t, **model_kwargs)` (the reference's closure contract):
        denoiser and step kernels still run on the HIP library, only the gradient comes from the callback."""
        dev = self._device()
        T = self.num_timesteps
        B, n = x.shape
        n_steps = T if n_steps is None else int(n_steps)
        ws, bs, acts = self._denoiser_stack()
        emb = self._emb_table(self._denoise_fn.time_embed, self._denoise_fn.dim_t)
        sched = self._c_schedule()
        nz = noise["noise"].to(dev).double().contiguous() if noise is not None else None
        for s in range(n_steps):
            i = T - 1 - s
            t = torch.full((B,), i, device=dev, dtype=torch.long)
            out = run_stack(ws, bs, acts, x.float(), rowadd0=emb[i])[-1]
            grad = cond_fn(x, t, **model_kwargs).float().contiguous()
            check(lib().clava_gauss_step_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(out), c_int(out.stride(0)), ptr(x),
                                             ptr(grad), c_int(n), ptr(nz[s]) if nz is not None else None, None, c_int(0), None,
                                             None, c_int(i), c_int(1 if s == 0 else 0), c_u64(seed), c_u64(row_offset),
                                             ctypes.c_uint(s), stream_ptr()), "clava_gauss_step_f32")
        return [nz, emb]

    def sample_all(self, num_samples, batch_size, y_dist, ddim=False, model_kwargs=None, cond_fn=None, *, seed=None):
        """Batches of `batch_size` until `num_samples` rows exist (`:1072-1099`): returns (x_gen CPU [N, n_num+m], y_gen CPU [N]);
        raises FoundNANsError if any sampled row has a NaN."""
        if ddim:
            raise NotImplementedError("DDIM sampling is unused by the pipeline (ddim=False, pipeline_utils.py:429)")
        b = batch_size
        seed = _fresh_seed() if seed is None else seed
        all_y, all_samples = [], []
        num_generated = 0
        while num_generated < num_samples:
            sample, out_dict = self.sample(b, y_dist, model_kwargs=model_kwargs, cond_fn=cond_fn, seed=seed,
                                           row_offset=num_generated)
            mask_nan = torch.any(sample.isnan(), dim=1)
            sample = sample[~mask_nan]
            out_dict['y'] = out_dict['y'].cpu()[~mask_nan]
            all_samples.append(sample)
            all_y.append(out_dict['y'])
            if sample.shape[0] != b:
                raise FoundNANsError
            num_generated += sample.shape[0]
        x_gen = torch.cat(all_samples, dim=0)[:num_samples]
        y_gen = torch.cat(all_y, dim=0)[:num_samples]
        return x_gen, y_gen
