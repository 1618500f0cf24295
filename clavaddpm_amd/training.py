"""Denoiser training on the HIP library: forward with saved activations, fused losses, backward GEMMs.

Two callers share this core:
  * the compat path - `GaussianMultinomialDiffusion.mixed_loss` / `MLPDiffusion.forward` under autograd return ordinary
    differentiable tensors (two `torch.autograd.Function`s whose backward runs the HIP kernels), so the reference's
    own training loop (`scripts/train.py:30-40`: zero_grad, mixed_loss, backward, torch AdamW) works unchanged;
  * the fused path - `clavaddpm_amd.scripts.train.Trainer` calls `forward_backward` directly and applies
    `clava_adamw_ema_f32` to one flat parameter buffer (optionally after an RCCL all-reduce of one flat gradient).

Mirrors `mixed_loss` (`tab_ddpm/gaussian_multinomial_diffsuion.py:636-675`) and autograd's backward through
`MLPDiffusion.forward` (`tab_ddpm/modules.py:451-460`).
"""
import ctypes
import os

import numpy as np

import torch

from . import _lib, ops
from ._lib import c_float, c_int, check, lib, ptr, require_cuda, stream_ptr
from .tab_ddpm.modules import timestep_embedding

c_ll = ctypes.c_longlong
WGRAD_F16X3 = os.environ.get("CLAVA_WGRAD_F16X3", "1") != "0"   # hidden-block weight gradients on the split-f16 kernels
WIDE_END_LAYERS = int(os.environ.get("CLAVA_WIDE_END_LAYERS", "256"))   # D_in from which proj / head run split-f16 too (0 = never)
FRONT_FUSED = os.environ.get("CLAVA_TRAIN_FRONT", "1") != "0"   # time MLP + proj + split pack of the training forward in one launch
DEVICE_DRAWS = os.environ.get("CLAVA_TRAIN_DRAWS", "device") != "torch"   # captured step: t / noise from the in-kernel generator
NARROW_ENDS = os.environ.get("CLAVA_TRAIN_NARROW", "1") != "0"   # head / proj of tables with <= 16 columns on the memory-bound kernels
WGRAD_DEFER = os.environ.get("CLAVA_WGRAD_DEFER", "0") != "0"   # hidden-block weight gradients after the input-gradient chain (measured: slower)
CAT_GATHER = os.environ.get("CLAVA_TRAIN_CAT_GATHER", "1") != "0"   # wide tables: first layer and its weight gradient from (numeric columns, category indices)
TAIL_BRANCHES = os.environ.get("CLAVA_TAIL_BRANCHES", "1") != "0"   # proj / time-embedding gradients on parallel streams
GRAD_SPLITS = int(os.environ.get("CLAVA_GRAD_SPLITS", "8"))   # batch ranges per weight-gradient product (4096-row batches -> 512 rows per slab, >= 512 workgroups)


def param_list(net):
    """Parameters of an MLPDiffusion in `net.parameters()` order (the order torch.optim sees, scripts/train.py:17)."""
    return list(net.parameters())


class FlatLayout:
    """Offsets of every parameter of a denoiser inside one flat float32 buffer."""

    def __init__(self, net, skip_prefixes=()):
        self.names, self.shapes, self.offsets = [], [], []
        off = 0
        for name, p in net.named_parameters():
            if name.startswith(tuple(skip_prefixes)) if skip_prefixes else False:
                continue
            self.names.append(name)
            self.shapes.append(tuple(p.shape))
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned inside the flat buffer
        self.total = off
        self.index = {n: i for i, n in enumerate(self.names)}

    def view(self, flat, name):
        i = self.index[name]
        n = 1
        for s in self.shapes[i]:
            n *= s
        return flat[self.offsets[i]:self.offsets[i] + n].view(self.shapes[i])


def _silu(z):
    a = torch.empty_like(z)
    check(lib().clava_silu_f32(ptr(z), ptr(a), c_ll(z.numel()), stream_ptr()), "clava_silu_f32")
    return a


def _silu_backward(z, da):
    dz = torch.empty_like(z)
    check(lib().clava_silu_backward_f32(ptr(z), ptr(da), ptr(dz), c_ll(z.numel()), stream_ptr()), "clava_silu_backward_f32")
    return dz


def denoiser_forward(net, x_in, t):
    """MLPDiffusion forward keeping what the backward needs.  Returns (model_out, saved)."""
    te = timestep_embedding(t, net.dim_t)
    w0, b0 = net.time_embed[0].weight.detach(), net.time_embed[0].bias.detach()
    w2, b2 = net.time_embed[2].weight.detach(), net.time_embed[2].bias.detach()
    z1 = ops.linear(te, w0, b0)
    a1 = _silu(z1)
    emb = ops.linear(a1, w2, b2)
    ws, bs, acts = net.stack()
    h = ops.linear(x_in, ws[0].detach(), bs[0].detach(), rowadd=emb)            # proj(x) + emb (modules.py:459)
    hs = [h]
    for l in range(1, len(ws)):
        h = ops.linear(h, ws[l].detach(), bs[l].detach(), act="relu" if l < len(ws) - 1 else None)
        hs.append(h)
    return hs[-1], (x_in, te, z1, a1, hs)


def denoiser_backward(net, saved, d_out, layout, slabs, splits=GRAD_SPLITS, want_dx=False, collect=None):
    """Backward through the denoiser: fills `slabs` [splits, layout.total] with partial parameter gradients
    (weight slabs from clava_gemm_wgrad_f32, bias slabs from clava_colsum_slabs_f32); returns dx or None."""
    x_in, te, z1, a1, hs = saved
    total = layout.total
    M = d_out.shape[0]

    def param_grads(prefix, dy, inp):
        wi = layout.index[prefix + ".weight"]
        n_out, k_in = layout.shapes[wi]
        wslab = slabs[:, layout.offsets[wi]:]
        ops.linear_wgrad(dy, inp, splits=splits, out=wslab, slab_stride=total)
        bi = layout.index[prefix + ".bias"]
        bslab = slabs[:, layout.offsets[bi]:]
        check(lib().clava_colsum_slabs_f32(ptr(dy), c_int(dy.stride(0)), c_int(M), c_int(n_out), c_int(splits), ptr(bslab),
                                           c_ll(total), stream_ptr()), "clava_colsum_slabs_f32")

    n_blocks = len(net.mlp.blocks)
    names = ["proj"] + [f"mlp.blocks.{i}.linear" for i in range(n_blocks)] + ["mlp.head"]
    ws, _, _ = net.stack()
    g = d_out                                   # gradient w.r.t. the pre-activation of layer l (head has no activation)
    for l in range(len(ws) - 1, 0, -1):
        param_grads(names[l], g, hs[l - 1])
        # input gradient, with ReLU' of the previous block folded in (layer 0 = proj has no activation)
        g = ops.linear_dgrad(g, ws[l].detach(), act_out=hs[l - 1] if l - 1 >= 1 else None, slope=0.0)
        if collect is not None:
            collect[l - 1] = g                  # gradient w.r.t. the pre-activation of layer l-1
    param_grads("proj", g, x_in)
    # time-embedding branch: emb was added to proj's output, so d emb = g
    param_grads("time_embed.2", g, a1)
    d_a1 = ops.linear_dgrad(g, net.time_embed[2].weight.detach())
    d_z1 = _silu_backward(z1, d_a1)
    param_grads("time_embed.0", d_z1, te)
    if want_dx:
        return ops.linear_dgrad(g, ws[0].detach())
    return None


def narrow_ok(w, inp_cols_first):
    """Whether an end layer with weight `w` ([out, in]) runs on csrc/narrow.hip: its narrow side (out for the head, in
    for proj) has at most 16 columns and the wide side meets the kernels' alignment."""
    out_f, in_f = w.shape
    if not NARROW_ENDS:
        return False
    if inp_cols_first:            # proj: narrow input
        return in_f <= 16
    return out_f <= 16 and in_f % 4 == 0 and out_f * in_f <= 16384


def narrow_head(h, w, b):
    """h [B, K] @ w[D, K]^T + b on clava_narrow_head_f32."""
    B, K = h.shape
    D = w.shape[0]
    out = torch.empty((B, D), device=h.device, dtype=torch.float32)
    check(lib().clava_narrow_head_f32(ptr(h), c_ll(h.stride(0)), c_int(B), c_int(K), ptr(w), ptr(b), c_int(D), ptr(out), c_int(D),
                                      stream_ptr()), "clava_narrow_head_f32")
    return out


def narrow_dgrad(dy, w, act, slope, want_f32, out_split, shift, flag):
    """(dy [B, D] @ w [D, J]) * act'(act) as split rows (`out_split`, may be None) and / or float32."""
    B, D = dy.shape
    J = w.shape[1]
    g = torch.empty((B, J), device=dy.device, dtype=torch.float32) if want_f32 else None
    if out_split is not None:
        out_split.shift = shift
    check(lib().clava_narrow_dgrad_f32(ptr(dy), c_int(dy.stride(0)), c_int(B), c_int(D), ptr(w), c_int(J), ptr(act),
                                       c_ll(act.stride(0) if act is not None else 0), c_float(slope), ptr(g), c_ll(J),
                                       ptr(out_split.buf) if out_split is not None else None,
                                       c_ll(out_split.pitch if out_split is not None else 0), c_int(shift), ptr(flag), stream_ptr()),
          "clava_narrow_dgrad_f32")
    return g


def narrow_wgrad(A, Bm, splits, dw, stride_d, stride_j, db_narrow, db_wide, slab_stride):
    B, D = A.shape
    check(lib().clava_narrow_wgrad_f32(ptr(A), c_ll(A.stride(0)), c_int(D), ptr(Bm), c_ll(Bm.stride(0)), c_int(Bm.shape[1]), c_int(B),
                                       c_int(splits), ptr(dw), c_ll(stride_d), c_ll(stride_j), ptr(db_narrow), ptr(db_wide),
                                       c_ll(slab_stride), stream_ptr()), "clava_narrow_wgrad_f32")


# ------------------------------------------------------------------------- split-f16 forward / input gradients
class SplitTrainState:
    """State of the split-f16 training path of one Trainer (csrc/split.h): the hidden blocks' forward products and
    their input-gradient products run on the f16 matrix cores at float32-level accuracy; weight gradients, the 2- and
    D_in-wide end layers and the time-embedding branch stay on the float32-input MFMA kernels.

    Weights change every step, so they are re-packed every step (17 MB).  The power-of-two plane scales are taken from
    a float32 step: every `period`-th step runs entirely on the float32 kernels and records the largest magnitude of
    every weight, activation and gradient tensor; scales place those near 2^6 (activations, gradients) / 2^10
    (weights), i.e. a factor 1000 / 64 below the float16 limit, far more than anything moves in `period` steps at
    training learning rates.  The device overflow flag is read at the same cadence; a raised flag raises
    SplitOverflowError, which `Trainer.run_loop` answers by restoring its snapshot of the window's start and re-running
    the window on the float32 kernels."""

    def __init__(self, period=200):
        self.period = period
        self.shift_h = self.shift_g = self.shift_w = None
        self.flag = None
        self.steps_since = None
        self.generation = 0       # bumped by every calibration: captured graphs bake the scales in
        self.flag_reduce = None   # data parallel: max of the overflow flag over ranks (set by the Trainer)
        self.bufs = {}

    def due(self):
        return self.shift_h is None or self.steps_since >= self.period

    def side_stream(self, device, which=0):
        """Stream `which` of the step's parallel branches (0: weight gradients / weight packing; 1, 2: the small
        products at the end of the backward)."""
        ss = self.__dict__.setdefault("_sides", {})
        if which not in ss:
            ss[which] = torch.cuda.Stream(device=device)
        return ss[which]

    def buf(self, key, rows, cols, device):
        """The persistent split buffer `key` of this shape.  Activation / gradient buffers are kept PER SHAPE: a captured
        step holds their addresses, and a trainer that alternates batch shapes (the short last batch of every epoch)
        replays one graph per shape -- replacing a buffer would hand its memory back to the allocator under a live graph.
        Weight operands ("w", l) / ("wt", l) do not depend on the batch and are looked up by their plain key."""
        k = key if key[0] in ("w", "wt") else key + (rows, cols)
        b = self.bufs.get(k)
        if b is None or (b.rows, b.cols) != (rows, cols):
            b = ops.split_empty(rows, cols, 0, device)
            self.bufs[k] = b
        return b

    def pack_weights(self, ws, wide, device, gather=False):
        """Packs the step's weights for both products in one launch (ops.PackPlan): layers 1 .. n-2 as the forward operand
        ("w", l) and, transposed, as the input-gradient operand ("wt", l); for wide tables also proj (forward only) and the
        head (both).  The plan is rebuilt when a weight's storage or a scale changed."""
        n = len(ws)
        layers = [(l, True) for l in range(1, n - 1)]
        if wide:
            layers = ([] if gather else [(0, False)]) + layers + [(n - 1, True)]
        key = tuple((l, ws[l].data_ptr(), self.shift_w[l]) for l, _ in layers)
        # Plans are kept for the lifetime of this state, one per (layer set, weight storage, scales): a captured step bakes
        # the address of its plan's device descriptor table into the graph, and the eager steps that run between replays
        # (the warm-up of a new batch shape takes the dense form of the first layer where the captured step takes the gather
        # form, i.e. another layer set) must not hand that table back to the allocator under the live graph.  The set of
        # keys is small: two layer sets x the scale generations of a run (a calibration rarely moves a weight scale).
        plans = self.__dict__.setdefault("_pack_plans", {})
        plan = plans.get(key)
        if plan is None:
            items = []
            for l, both in layers:
                w = ws[l].detach()
                items.append((w, self.shift_w[l], self.buf(("w", l), w.shape[0], w.shape[1], device),
                              self.buf(("wt", l), w.shape[1], w.shape[0], device) if both else None))
            plan = plans[key] = ops.PackPlan(items, device)
        plan.run(self.flag)

    def wide(self, d_in):
        """Whether proj / head of a `d_in`-column table run on the split-f16 kernels too."""
        return WIDE_END_LAYERS > 0 and d_in >= WIDE_END_LAYERS and getattr(self, "shift_x", None) is not None

    # ---- gather form of the first layer (tables with categorical columns, wide input) ---------------------------------
    def gather_ok(self, diffusion, net, batch):
        """Whether a step of `batch` rows runs proj from (numeric columns, category indices) (csrc/cat_input.hip) and its
        weight gradient from csrc/cat_wgrad.hip instead of products over the [B, sum K] log-one-hot image."""
        d_in = diffusion.num_numerical_features + diffusion._sum_k
        w0 = net.stack()[0][0]
        return (CAT_GATHER and DEVICE_DRAWS and diffusion.has_cat and self.wide(d_in) and net.dim_t == 128 and tuple(w0.shape) == (128, d_in)
                and getattr(self, "shift_xn", None) is not None and wgrad_split_ok(batch) and batch % 32 == 0)

    def gather_plan(self, diffusion, device):
        """Per-table constants of the gather form: device offsets, the weight-gradient kernel's workgroup plan, log(1e-30)."""
        gp = getattr(self, "_gather_plan", None)
        if gp is None:
            off = np.ascontiguousarray(diffusion._offsets_i32.cpu().numpy().astype(np.int32))
            m = len(off) - 1
            op = off.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
            nwg = lib().clava_cat_wgrad_workgroups(op, c_int(m))
            wf, wk = np.zeros(nwg, np.int32), np.zeros(nwg, np.int32)
            check(lib().clava_cat_wgrad_plan(op, c_int(m), wf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                             wk.ctypes.data_as(ctypes.POINTER(ctypes.c_int))), "clava_cat_wgrad_plan")
            gp = self._gather_plan = {"m": m, "nwg": nwg, "wf": torch.from_numpy(wf).to(device), "wk": torch.from_numpy(wk).to(device),
                                      "log_tiny": float(torch.log(torch.tensor(1e-30, dtype=torch.float32)))}
        return gp

    def gather_weights(self, w0, b0, n_num):
        """proj's weight transposed ([D_in][128]: rows = input columns) and its bias with the constant part of the
        log-one-hot columns folded in, b + log(1e-30) * (row sums of the categorical columns), into persistent buffers
        (a captured step rebuilds them every replay: the weights change every step)."""
        D = w0.shape[1]
        if getattr(self, "_wt0", None) is None or self._wt0.shape[0] != D:
            self._wt0 = torch.empty((D, w0.shape[0]), device=w0.device, dtype=torch.float32)
            self._cat_bias = torch.empty(w0.shape[0], device=w0.device, dtype=torch.float32)
            self._cat_partial = torch.empty(lib().clava_cat_input_weights_tiles(c_int(D)) * 128, device=w0.device, dtype=torch.float64)
        log_tiny = float(torch.log(torch.tensor(1e-30, dtype=torch.float32)))
        if w0.shape[0] == 128 and w0.stride(1) == 1:
            check(lib().clava_cat_input_weights_f32(ptr(w0), c_ll(w0.stride(0)), c_int(128), c_int(D), c_int(n_num), ptr(b0),
                                                    ctypes.c_double(log_tiny), ptr(self._wt0), ptr(self._cat_partial), ptr(self._cat_bias),
                                                    stream_ptr()), "clava_cat_input_weights_f32")
            return
        self._wt0.copy_(w0.t())
        self._cat_bias.copy_(b0.double() + log_tiny * w0[:, n_num:].sum(dim=1, dtype=torch.float64))

    def start_pack(self, net, d_in, gather_n_num=None):
        """Starts this step's weight packing on stream 0 of the parallel branches, forked from the current stream: the
        packing only depends on the weights, so a step calls this before it draws t and the noise and the packing runs
        beside the whole front of the step.  `denoiser_forward_split` joins it before its first split product.
        `gather_n_num`: the step runs proj in gather form (number of numeric columns); its weight operands are rebuilt
        here instead of proj's packed planes."""
        ws = net.stack()[0]
        dev = ws[0].device
        main = torch.cuda.current_stream()
        side = self.side_stream(dev)
        fork = torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            if gather_n_num is not None:             # first: the first layer is the first consumer
                self.gather_weights(ws[0].detach(), net.stack()[1][0].detach(), gather_n_num)
                self._gathered = torch.cuda.Event()
                self._gathered.record(side)
            self.pack_weights(ws, self.wide(d_in), dev, gather=gather_n_num is not None)   # every weight operand of this step's products
            self._packed = torch.cuda.Event()
            self._packed.record(side)

    def calibrate(self, ws, hs, gs, x_in=None, d_out=None, n_num=None):
        """ws[l], hs[l] (layer outputs), gs[l] (gradient w.r.t. layer l's pre-activation, l <= n-2) from a float32 step;
        x_in / d_out (denoiser input, loss gradient) fix the scales of the wide end layers' operands; `n_num` (numeric
        columns at the front of x_in) fixes the scale of the numeric block alone (gather form of the first layer)."""
        self.calibrate_from(self.calibration_stats(ws, hs, gs, x_in, d_out, n_num), len(ws), x_in is not None and bool(n_num))

    @staticmethod
    def calibration_stats(ws, hs, gs, x_in=None, d_out=None, n_num=None):
        """The largest magnitudes `calibrate_from` turns into scales, as one device tensor (capturable: no host access)."""
        n = len(ws)
        zero = torch.zeros((), device=ws[0].device)
        return torch.stack([w.detach().abs().max() for w in ws] + [h.abs().max() for h in hs] +
                           [gs[l].abs().max() if gs[l] is not None else zero for l in range(n)] +
                           [x_in.abs().max() if x_in is not None else zero, d_out.abs().max() if d_out is not None else zero,
                            x_in[:, :n_num].abs().max() if x_in is not None and n_num else zero])

    def calibrate_from(self, stats, n, has_xn):
        """Reads the overflow flag and the statistics (the window's one host synchronisation) and fixes the scales."""
        from .split_nets import TARGET_LOG2, shift_for
        if self.flag is None:
            self.flag = torch.zeros(1, device=stats.device, dtype=torch.int32)
        elif int((self.flag_reduce(self.flag.clone()) if self.flag_reduce is not None else self.flag).item()) != 0:
            raise _lib.SplitOverflowError("split-f16 training: a tensor left the float16 range since the last calibration "
                                          "(magnitudes grew by more than 60x within %d steps)" % self.period)
        v = stats.cpu().tolist()
        before = (getattr(self, "shift_x", None), getattr(self, "shift_d", None), getattr(self, "shift_xn", None), self.shift_w,
                  self.shift_h, self.shift_g)
        self.shift_x = shift_for(v[3 * n], TARGET_LOG2 + 1)      # the log-one-hot columns are exactly 0 / -69.08
        self.shift_d = shift_for(v[3 * n + 1], TARGET_LOG2)
        self.shift_xn = shift_for(v[3 * n + 2], TARGET_LOG2) if has_xn else None
        self.shift_w = [shift_for(x, 10) for x in v[:n]]
        self.shift_h = [shift_for(x, TARGET_LOG2) for x in v[n:2 * n]]
        self.shift_g = [shift_for(x, TARGET_LOG2) for x in v[2 * n:3 * n]]
        self.steps_since = 0
        # a captured step bakes the scales in: it stays valid while no power-of-two scale moved (the usual case once the
        # magnitudes have settled; a recapture costs an eager step and a capture, ~8 ms per calibration window)
        if before != (self.shift_x, self.shift_d, self.shift_xn, self.shift_w, self.shift_h, self.shift_g):
            self.generation += 1


def wgrad_split_ok(batch, splits=None):
    """Whether the hidden-block weight gradients of a `batch`-row step run on the split-f16 kernels: every gradient slab
    must be written each step, i.e. `splits` non-empty ranges of whole 32-row blocks."""
    splits = GRAD_SPLITS if splits is None else splits
    nkb = (batch + 31) // 32
    return WGRAD_F16X3 and (splits - 1) * ((nkb + splits - 1) // splits) < nkb


def denoiser_forward_split(net, x_in, t, st, cat=None):
    """`denoiser_forward` with the hidden blocks (layers 1 .. n-2 of the stack) on clava_gemm_f16x3.
    `cat` = (idx [B, m] int32 noised categories, diffusion): x_in then holds the NUMERIC columns only and proj runs in
    gather form (csrc/cat_input.hip; the caller checked `st.gather_ok` and started the packing with `gather_n_num`)."""
    ws, bs, acts = net.stack()
    dev = x_in.device
    d_in = ws[0].shape[1] if cat is not None else x_in.shape[1]
    wide = st.wide(d_in)
    # the weight packing runs on a second stream (a parallel branch of the captured graph), started by the caller before
    # the draws or here, and is joined before the first split product
    main = torch.cuda.current_stream()
    if getattr(st, "_packed", None) is None:
        st.start_pack(net, d_in, gather_n_num=x_in.shape[1] if cat is not None else None)
    packed, st._packed = st._packed, None
    w0, b0 = net.time_embed[0].weight.detach(), net.time_embed[0].bias.detach()
    w2, b2 = net.time_embed[2].weight.detach(), net.time_embed[2].bias.detach()
    n = len(ws)
    B = x_in.shape[0]
    planes_only = wgrad_split_ok(B)               # float32 activation copies only feed the float32 weight-gradient kernels
    # D_in-wide end layers (one-hot heavy tables, BASELINE.json configs[4]) go to the f16 matrix cores too; at a handful
    # of columns they are a few MFLOP and stay on the float32 kernels
    xs = None
    catinfo = None
    fused_front = (FRONT_FUSED and not wide and net.dim_t == 128 and tuple(ws[0].shape) == (128, x_in.shape[1])
                   and t.dtype == torch.int64 and x_in.stride(1) == 1)
    if cat is not None:
        # time MLP alone (clava_train_front_f32 without proj), then proj from the numeric columns and m gathered weight rows
        from .tab_ddpm.modules import _freqs
        idx, diffusion = cat
        gp = st.gather_plan(diffusion, dev)
        n_num = x_in.shape[1]
        te, z1, a1, emb = (torch.empty((B, 128), device=dev, dtype=torch.float32) for _ in range(4))
        check(lib().clava_train_front_f32(ptr(t.contiguous()), c_int(B), ptr(_freqs(128, dev)), c_int(128), None, c_ll(0), c_int(0),
                                          ptr(w0), ptr(b0), ptr(w2), ptr(b2), None, None, ptr(te), ptr(z1), ptr(a1), ptr(emb), None,
                                          c_ll(0), c_int(0), None, stream_ptr()), "clava_train_front_f32(emb)")
        main.wait_event(st._gathered)             # the transposed weight and the folded bias come from the packing stream
        h = None if planes_only else torch.empty((B, 128), device=dev, dtype=torch.float32)
        hsp = st.buf(("h", 0), B, 128, dev)
        hsp.shift = st.shift_h[0]
        check(lib().clava_cat_input_f32(ptr(x_in), c_int(x_in.stride(0)), ptr(idx), c_int(B), c_int(n_num), c_int(gp["m"]), c_int(128),
                                        ptr(diffusion._offsets_i32), ptr(st._wt0), ptr(st._wt0[n_num:]), ptr(st._cat_bias), ptr(emb),
                                        c_ll(128), c_float(gp["log_tiny"]), c_int(0), ptr(hsp.buf), c_ll(hsp.pitch), c_int(hsp.shift),
                                        ptr(h), c_ll(128), ptr(st.flag), stream_ptr()), "clava_cat_input_f32")
        main.wait_event(packed)                   # the other layers' operands
        hs = [h]
        hsplit = [hsp]
        catinfo = (idx, diffusion, gp)
    elif fused_front:
        # timestep embedding, time MLP, proj and the split pack of h in one launch (csrc/train_front.hip)
        from .tab_ddpm.modules import _freqs
        te, z1, a1 = (torch.empty((B, 128), device=dev, dtype=torch.float32) for _ in range(3))
        h = None if planes_only else torch.empty((B, 128), device=dev, dtype=torch.float32)
        hsp = st.buf(("h", 0), B, 128, dev)
        hsp.shift = st.shift_h[0]
        check(lib().clava_train_front_f32(ptr(t.contiguous()), c_int(B), ptr(_freqs(128, dev)), c_int(128), ptr(x_in), c_ll(x_in.stride(0)),
                                          c_int(x_in.shape[1]), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(ws[0].detach()), ptr(bs[0].detach()),
                                          ptr(te), ptr(z1), ptr(a1), ptr(h), ptr(hsp.buf), c_ll(hsp.pitch), c_int(hsp.shift),
                                          ptr(st.flag), stream_ptr()), "clava_train_front_f32")
        hs = [h]
        hsplit = [hsp]
    else:
        te = timestep_embedding(t, net.dim_t)
        z1 = ops.linear(te, w0, b0)
        a1 = _silu(z1)
        emb = ops.linear(a1, w2, b2)
        if wide:
            main.wait_event(packed)
            xs = ops.split_pack(x_in, st.shift_x, st.flag, out=st.buf(("x",), B, x_in.shape[1], dev))
            h = ops.linear_f16x3(xs, st.bufs[("w", 0)], bs[0].detach(), out_f32=True, overflow=st.flag)
            h.add_(emb)                           # per-row time embedding (the split kernel's row term is batch-constant)
        else:
            h = ops.linear(x_in, ws[0].detach(), bs[0].detach(), rowadd=emb)    # proj(x) + emb (modules.py:459), float32
        hs = [h]
        hsplit = [ops.split_pack(h, st.shift_h[0], st.flag, out=st.buf(("h", 0), B, h.shape[1], dev))]
    if not wide:
        main.wait_event(packed)
    wsp = []
    for l in range(1, n - 1):
        w = ws[l]
        wsp.append(st.bufs[("w", l)])
        last = l == n - 2
        if last and not wide:                     # the head reads float32: no split copy of the last hidden activation
            hf = ops.linear_f16x3(hsplit[-1], wsp[-1], bs[l].detach(), act="relu", out_f32=True, overflow=st.flag)
            hsplit.append(None)
        elif last or not planes_only:
            hsp, hf = ops.linear_f16x3(hsplit[-1], wsp[-1], bs[l].detach(), act="relu", both=True, out_shift=st.shift_h[l],
                                       out=st.buf(("h", l), B, w.shape[0], dev), overflow=st.flag)
            hsplit.append(hsp)
        else:                                     # the backward only reads the planes (mask, transposed operand): no float32 copy
            hsplit.append(ops.linear_f16x3(hsplit[-1], wsp[-1], bs[l].detach(), act="relu", out_shift=st.shift_h[l],
                                           out=st.buf(("h", l), B, w.shape[0], dev), overflow=st.flag))
            hf = None
        hs.append(hf)
    if wide:
        wh = ws[n - 1]
        whs = st.bufs[("w", n - 1)]
        # rows padded to a multiple of 4 floats: 16-byte stores in the epilogue whatever D_in is
        d_pad = (wh.shape[0] + 3) // 4 * 4
        out = torch.empty((B, d_pad), device=dev, dtype=torch.float32)[:, :wh.shape[0]]
        ops.linear_f16x3(hsplit[-1], whs, bs[n - 1].detach(), out_f32=True, overflow=st.flag, out=out)
    elif narrow_ok(ws[n - 1], False) and hs[-1].stride(1) == 1 and hs[-1].stride(0) % 4 == 0:
        out = narrow_head(hs[-1], ws[n - 1].detach(), bs[n - 1].detach())
    else:
        out = ops.linear(hs[-1], ws[n - 1].detach(), bs[n - 1].detach())
    hs.append(out)
    return out, (x_in, te, z1, a1, hs, hsplit, xs, catinfo)


LOSS_SPLIT = os.environ.get("CLAVA_TRAIN_LOSS_SPLIT", "1") != "0"
QSAMPLE_CLOSED = os.environ.get("CLAVA_TRAIN_QSAMPLE", "closed") != "gumbel"


def loss_split_target(st, saved, model_out, splits=GRAD_SPLITS):
    """Where `loss_and_grad` may write the loss gradient as split rows (wide tables, a batch of whole 32-row blocks: the
    head's backward then consumes nothing but the split operand), or None: the float32 gradient + a pack launch."""
    M, D = model_out.shape
    wide = saved[6] is not None or saved[7] is not None
    if not (LOSS_SPLIT and wide and M % 32 == 0 and wgrad_split_ok(M, splits)):
        return None
    b = st.buf(("d",), M, D, model_out.device)       # zero-initialised: the pad columns of the last block stay zero
    b.shift = st.shift_d
    return b


def denoiser_backward_split(net, saved, d_out, layout, slabs, st, splits=GRAD_SPLITS, on_layer_done=None, after_blocks=None):
    """`denoiser_backward` with the input-gradient products of layers n-2 .. 1 on clava_gemm_f16x3 and the hidden blocks'
    weight / bias gradients on the batch-major split kernel.  `on_layer_done(prefix)` (data parallel) is called with the
    weight-gradient stream current right after a layer's gradient slabs are enqueued: the Trainer reduces them and starts
    that layer's all-reduce there, beside the rest of the input-gradient chain.  `after_blocks()` is called with that
    stream current once the gradients of every hidden block and of the head are enqueued on it (the captured step starts
    their optimizer update there, beside the small products of proj and the time-embedding branch)."""
    x_in, te, z1, a1, hs, hsplit, xs, catinfo = saved
    wide = xs is not None or catinfo is not None
    total = layout.total
    d_split = d_out if isinstance(d_out, ops.SplitTensor) else None        # loss_and_grad(split_out=...): no float32 gradient
    M = d_split.rows if d_split is not None else d_out.shape[0]
    dev = te.device
    if d_split is not None and not (wide and wgrad_split_ok(M, splits) and M % 32 == 0):
        raise _lib.ClavaHipError("denoiser_backward_split: a split-row loss gradient needs the wide-table path and a batch of whole 32-row blocks")

    def param_grads(prefix, dy, inp):
        wi = layout.index[prefix + ".weight"]
        n_out, k_in = layout.shapes[wi]
        ops.linear_wgrad(dy, inp, splits=splits, out=slabs[:, layout.offsets[wi]:], slab_stride=total)
        bi = layout.index[prefix + ".bias"]
        check(lib().clava_colsum_slabs_f32(ptr(dy), c_int(dy.stride(0)), c_int(M), c_int(n_out), c_int(splits),
                                           ptr(slabs[:, layout.offsets[bi]:]), c_ll(total), stream_ptr()), "clava_colsum_slabs_f32")

    n_blocks = len(net.mlp.blocks)
    names = ["proj"] + [f"mlp.blocks.{i}.linear" for i in range(n_blocks)] + ["mlp.head"]
    ws, _, _ = net.stack()
    n = len(ws)
    # Weight / bias gradients only consume (g_l, h_{l-1}); nothing in the input-gradient chain waits for them.  They run
    # on a second stream beside that chain (float32 MFMA wgrad kernels filling the split-f16 kernels' bubbles) and are
    # joined before the optimizer.  Tensors handed to the side stream are kept alive until the next step.
    main = torch.cuda.current_stream()
    side = st.side_stream(dev)
    keep = []
    # every slab must be written each step: the split product needs `splits` non-empty ranges of whole 32-row blocks
    split_wgrad = wgrad_split_ok(M, splits)

    def param_grads_async(prefix, dy, inp, dy_split=None, inp_split=None, side=side, narrow=None):
        ev = torch.cuda.Event()
        ev.record(main)
        keep.append((dy, inp, dy_split, inp_split))
        with torch.cuda.stream(side):
            side.wait_event(ev)
            if narrow is not None:
                # an end layer with a handful of columns on one side: weight and bias slabs from one memory-bound launch
                wi, bi = layout.index[prefix + ".weight"], layout.index[prefix + ".bias"]
                wsl, bsl = slabs[:, layout.offsets[wi]:], slabs[:, layout.offsets[bi]:]
                if narrow == "head":        # dW[d][j] = sum_b dy[b][d] inp[b][j];  db = column sums of dy (the narrow side)
                    narrow_wgrad(dy, inp, splits, wsl, inp.shape[1], 1, bsl, None, total)
                else:                       # proj: dW[c][k] = sum_b dy[b][c] inp[b][k];  db = column sums of dy (the wide side)
                    narrow_wgrad(inp, dy, splits, wsl, 1, inp.shape[1], None, bsl, total)
            elif dy_split is not None and inp_split is not None and split_wgrad and M % 32 == 0:
                # weight AND bias gradient on the f16 matrix cores straight from the batch-major planes (the kernel reads
                # its LDS tiles transposed; the column sums of dy ride along as one more MFMA against a fragment of ones)
                wi, bi = layout.index[prefix + ".weight"], layout.index[prefix + ".bias"]
                ops.linear_wgrad_f16x3_tn(dy_split, inp_split, splits, slabs[:, layout.offsets[wi]:], total,
                                          bias_out=slabs[:, layout.offsets[bi]:])
            elif dy_split is not None and inp_split is not None and split_wgrad:
                # batch not a multiple of 32: the product of the transposed copies (their pad columns are zeros)
                wi = layout.index[prefix + ".weight"]
                dyt = ops.split_transpose(dy_split, out=st.buf(("T", prefix, "dy"), dy_split.cols, dy_split.rows, dev))
                xt = ops.split_transpose(inp_split, out=st.buf(("T", prefix, "x"), inp_split.cols, inp_split.rows, dev))
                ops.linear_wgrad_f16x3(dyt, xt, splits, slabs[:, layout.offsets[wi]:], total)
                bi = layout.index[prefix + ".bias"]
                check(lib().clava_colsum_slabs_split(ptr(dy_split.buf), c_ll(dy_split.pitch), c_int(dy_split.shift), c_int(M),
                                                     c_int(dy_split.cols), c_int(splits), ptr(slabs[:, layout.offsets[bi]:]), c_ll(total),
                                                     stream_ptr()), "clava_colsum_slabs_split")
            else:
                param_grads(prefix, dy, inp)
            if on_layer_done is not None:
                on_layer_done(prefix)

    # head: parameter gradients and the gradient w.r.t. the last hidden pre-activation
    if wide:
        ds = d_split if d_split is not None else ops.split_pack(d_out, st.shift_d, st.flag, out=st.buf(("d",), M, d_out.shape[1], dev))
        param_grads_async(names[n - 1], d_out, hs[n - 2], ds, hsplit[n - 2])
        wh = ws[n - 1]
        wht = st.bufs[("wt", n - 1)]
        gs, g = ops.linear_f16x3(ds, wht, act_out=hsplit[n - 2], slope=0.0, both=True, out_shift=st.shift_g[n - 2],
                                 out=st.buf(("g", n - 2), M, wh.shape[1], dev), overflow=st.flag)
    elif narrow_ok(ws[n - 1], False):
        param_grads_async(names[n - 1], d_out, hs[n - 2], narrow="head")
        gs = st.buf(("g", n - 2), M, ws[n - 1].shape[1], dev)
        # the float32 copy only feeds the float32 weight-gradient kernels of layer n-2
        g = narrow_dgrad(d_out, ws[n - 1].detach(), hs[n - 2], 0.0, not (split_wgrad and n - 2 >= 1), gs, st.shift_g[n - 2], st.flag)
    else:
        param_grads_async(names[n - 1], d_out, hs[n - 2])
        g = ops.linear_dgrad(d_out, ws[n - 1].detach(), act_out=hs[n - 2], slope=0.0)
        gs = ops.split_pack(g, st.shift_g[n - 2], st.flag, out=st.buf(("g", n - 2), M, g.shape[1], dev))
    # A workgroup of either product family fills a CU's LDS, so an input-gradient product and a weight-gradient product in
    # flight together only take turns on the CUs and the input-gradient chain runs at half speed beside the weight
    # gradients.  Enqueueing the weight gradients AFTER the chain (CLAVA_WGRAD_DEFER=1) was measured and is slower all the
    # same (0.660 vs 0.644 ms/step): side by side, each family fills the other's write-drain gaps between dependent kernels.
    deferred = []
    defer = WGRAD_DEFER and on_layer_done is None
    for l in range(n - 2, 0, -1):
        if defer:
            deferred.append((names[l], g, hs[l - 1], gs, hsplit[l - 1]))
        else:
            param_grads_async(names[l], g, hs[l - 1], gs, hsplit[l - 1])
        w = ws[l]
        wt = st.bufs[("wt", l)]                   # W^T for the input gradient, packed with the forward operand (pack_weights)
        if l - 1 >= 1 and split_wgrad:            # weight and bias gradients of layer l-1 read the planes: no float32 copy
            gs = ops.linear_f16x3(gs, wt, act_out=hsplit[l - 1], slope=0.0, out_shift=st.shift_g[l - 1],
                                  out=st.buf(("g", l - 1), M, w.shape[1], dev), overflow=st.flag)
            g = None
        elif l - 1 >= 1:
            gs, g = ops.linear_f16x3(gs, wt, act_out=hsplit[l - 1], slope=0.0, both=True, out_shift=st.shift_g[l - 1],
                                     out=st.buf(("g", l - 1), M, w.shape[1], dev), overflow=st.flag)
        elif wide:                                # layer 0 = proj has no activation; its wide weight gradient wants the planes
            gs, g = ops.linear_f16x3(gs, wt, both=True, out_shift=st.shift_g[0], out=st.buf(("g", 0), M, w.shape[1], dev),
                                     overflow=st.flag)
        else:                                     # only the float32 gradient is needed
            g = ops.linear_f16x3(gs, wt, out_f32=True, overflow=st.flag)
    for args in deferred:
        param_grads_async(*args)
    if after_blocks is not None:
        with torch.cuda.stream(side):
            after_blocks()
    # the end of the backward is five small products that only share g (and d_z1): three parallel branches instead of a
    # queue behind the hidden blocks' weight gradients -- proj on stream 1, time_embed.2 on stream 2, the chain
    # d_a1 -> d_z1 -> time_embed.0 on the main stream
    tail = [st.side_stream(dev, 1), st.side_stream(dev, 2)] if TAIL_BRANCHES else [side, side]
    if catinfo is not None:
        # proj in gather form: numeric columns of dW (and the bias) from the batch-major split kernel on the numeric block,
        # categorical columns from the class-wise row sums of g (csrc/cat_wgrad.hip); the split image of x_in does not exist
        idx, diffusion, gp = catinfo
        n_num = x_in.shape[1]
        wi, bi = layout.index["proj.weight"], layout.index["proj.bias"]
        D = layout.shapes[wi][1]
        wsl, bsl = slabs[:, layout.offsets[wi]:], slabs[:, layout.offsets[bi]:]
        ev = torch.cuda.Event()
        ev.record(main)
        keep.append((g, gs, x_in, idx))
        with torch.cuda.stream(tail[0]):
            tail[0].wait_event(ev)
            if n_num > 0:
                xn = ops.split_pack(x_in, st.shift_xn, st.flag, out=st.buf(("xn",), M, n_num, dev))
                ops.linear_wgrad_f16x3_tn(gs, xn, splits, wsl, total, bias_out=bsl, ldw=D)
            else:
                check(lib().clava_colsum_slabs_split(ptr(gs.buf), c_ll(gs.pitch), c_int(gs.shift), c_int(M), c_int(gs.cols), c_int(splits),
                                                     ptr(bsl), c_ll(total), stream_ptr()), "clava_colsum_slabs_split")
            check(lib().clava_cat_wgrad_f32(ptr(g), c_ll(g.stride(0)), ptr(bsl), c_int(M), c_int(128), ptr(idx), c_int(gp["m"]),
                                            ptr(diffusion._offsets_i32), ptr(gp["wf"]), ptr(gp["wk"]), c_int(gp["nwg"]), c_int(splits),
                                            c_float(gp["log_tiny"]), ptr(wsl), c_ll(D), c_int(n_num), c_ll(total), stream_ptr()),
                  "clava_cat_wgrad_f32")
            if on_layer_done is not None:
                on_layer_done("proj")
    elif wide:
        param_grads_async("proj", g, x_in, gs, xs, side=tail[0])
    else:
        param_grads_async("proj", g, x_in, side=tail[0], narrow="proj" if narrow_ok(ws[0], True) else None)
    param_grads_async("time_embed.2", g, a1, side=tail[1])
    if FRONT_FUSED and net.dim_t == 128 and g.stride(1) == 1:
        d_z1 = torch.empty_like(z1)               # input gradient of time_embed[2] and SiLU' in one launch
        check(lib().clava_train_time_bwd_f32(ptr(g), c_ll(g.stride(0)), c_int(M), c_int(128), ptr(net.time_embed[2].weight.detach()),
                                             ptr(z1), ptr(d_z1), stream_ptr()), "clava_train_time_bwd_f32")
    else:
        d_a1 = ops.linear_dgrad(g, net.time_embed[2].weight.detach())
        d_z1 = _silu_backward(z1, d_a1)
    if TAIL_BRANCHES:
        keep.append((d_z1, te))
        param_grads("time_embed.0", d_z1, te)
        if on_layer_done is not None:
            on_layer_done("time_embed.0")
    else:
        param_grads_async("time_embed.0", d_z1, te)
    for s_ in dict.fromkeys([side] + tail):
        done = torch.cuda.Event()
        done.record(s_)
        main.wait_event(done)
    st.keep = keep            # released by the next step, after the join above is on the main stream's queue
    st.steps_since += 1
    return None


def reduce_slabs(slabs, scale=1.0, out=None, lo=0, hi=None):
    """Flat gradient = scale * sum over slabs (fixed order); `lo:hi` restricts it to a range of the flat buffer."""
    S, total = slabs.shape
    if out is None:
        out = torch.empty(total, device=slabs.device, dtype=torch.float32)
    hi = total if hi is None else hi
    if hi > lo:
        check(lib().clava_slab_reduce_f32(ptr(slabs[:, lo:]), c_int(S), c_ll(total), c_ll(hi - lo), c_float(scale), ptr(out[lo:]),
                                          stream_ptr()), "clava_slab_reduce_f32")
    return out


# ---------------------------------------------------------------------------------------------------- losses
def _device_sum(v, scale, aux_stream=None):
    out = torch.empty((), device=v.device, dtype=torch.float32)
    if aux_stream is None:
        check(lib().clava_sum_f32(ptr(v), c_ll(v.numel()), c_float(scale), ptr(out), stream_ptr()), "clava_sum_f32")
        return out
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(aux_stream):
        aux_stream.wait_event(ev)
        check(lib().clava_sum_f32(ptr(v), c_ll(v.numel()), c_float(scale), ptr(out), stream_ptr()), "clava_sum_f32")
    return out


def noised_input(diffusion, x, t, noise, uniforms, seed, step_counter=None):
    """q-sample both halves straight into the denoiser input buffer.  Returns (x_in, idx0, idxt).  `step_counter` (a
    one-element int32 device tensor, or None) is added to the categorical generator's step key on the device: a captured
    step advances it once per replay instead of taking a new `seed` from the host."""
    dev = x.device
    B = x.shape[0]
    n = diffusion.num_numerical_features
    D = n + diffusion._sum_k
    x_in = torch.empty((B, D), device=dev, dtype=torch.float32)
    sched = diffusion._c_schedule()
    if n > 0:
        x_num = x[:, :n].float().contiguous()
        check(lib().clava_gauss_qsample_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(x_num), ptr(noise), ptr(t), ptr(x_in),
                                            c_int(D), stream_ptr()), "clava_gauss_qsample_f32")
    idx0, idxt = _noised_categories(diffusion, x, x_in, t, uniforms, seed, step_counter)
    return x_in, idx0, idxt


def _noised_categories(diffusion, x, x_in, t, uniforms, seed, step_counter, dense=True):
    """Categorical half of the forward noising (q_sample of :650-652) into x_in[:, n_num:] (`dense`; otherwise only the
    noised class indices are produced); returns (idx0, idxt)."""
    if not diffusion.has_cat:
        return None, None
    B = x.shape[0]
    n = diffusion.num_numerical_features
    D = x_in.shape[1]
    sched = diffusion._c_schedule()
    m, sk = diffusion._m, diffusion._sum_k
    idx0 = x[:, n:].to(torch.int32).contiguous()           # `x_cat.long()` of :650
    idxt = torch.empty_like(idx0)
    xin_cat = x_in[:, n:] if dense else None
    check(lib().clava_multinomial_step_counter_f32(ctypes.byref(sched), c_int(2), c_int(B), c_int(m), c_int(sk),
                                                   ptr(diffusion._offsets_i32), ptr(diffusion._log_k), None, c_int(0), ptr(idx0),
                                                   ptr(idxt), ptr(uniforms), c_int(sk), ptr(xin_cat), c_int(D), None, c_int(0),
                                                   ptr(t), c_int(0), _lib.c_u64(seed), _lib.c_u64(0), ctypes.c_uint(0),
                                                   ptr(step_counter), stream_ptr()),
          "clava_multinomial_step_counter_f32(q_sample)")
    return idx0, idxt


def draw_noised_input(diffusion, x, seed, step_counter, dense=True):
    """`draw` + `noised_input` with the draws made on the device (clava_train_draw_qsample_f32): timesteps, Gaussian
    noise and the numeric q-sample in one launch, keyed by (seed, *step_counter).  This is the captured step's form: it
    keeps torch's generator (two fill kernels before every replay, two draw kernels in it) out of the graph.  Returns
    (t, noise, x_in, idx0, idxt).  `dense=False`: x_in holds the numeric columns only and the log-one-hot image of the
    noised categories is not written (the first layer runs in gather form from idxt)."""
    dev = x.device
    B = x.shape[0]
    n = diffusion.num_numerical_features
    D = n + (diffusion._sum_k if dense else 0)
    x_in = torch.empty((B, D), device=dev, dtype=torch.float32)
    t = torch.empty(B, device=dev, dtype=torch.int64)
    noise = torch.empty((B, n), device=dev, dtype=torch.float32) if n > 0 else None
    xf = x if x.dtype == torch.float32 and x.stride(1) == 1 else x.float().contiguous()
    sched = diffusion._c_schedule()
    if diffusion.has_cat and QSAMPLE_CLOSED:
        # categorical q-sample in closed form inside the same launch (two uniforms per (row, feature) block instead of K
        # Gumbel draws: the same distribution, see clava_train_draw_mixed_f32); CLAVA_TRAIN_QSAMPLE=gumbel keeps the
        # per-class form
        m, sk = diffusion._m, diffusion._sum_k
        idx0 = torch.empty((B, m), device=dev, dtype=torch.int32)
        idxt = torch.empty_like(idx0)
        check(lib().clava_train_draw_mixed_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(xf), c_ll(xf.stride(0)), c_int(m),
                                               ptr(diffusion._offsets_i32), ptr(diffusion._log_k), _lib.c_u64(seed), ctypes.c_uint(0),
                                               ptr(step_counter), ptr(t), ptr(noise), ptr(x_in), c_int(D), ptr(idx0), ptr(idxt),
                                               stream_ptr()), "clava_train_draw_mixed_f32")
        if dense:
            check(lib().clava_cat_onehot_image_f32(c_int(B), c_int(m), c_int(sk), ptr(diffusion._offsets_i32), ptr(idxt), ptr(x_in[:, n:]),
                                                   c_ll(D), stream_ptr()), "clava_cat_onehot_image_f32")
        return t, noise, x_in, idx0, idxt
    check(lib().clava_train_draw_qsample_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(xf), c_ll(xf.stride(0)), _lib.c_u64(seed),
                                             ctypes.c_uint(0), ptr(step_counter), ptr(t), ptr(noise), ptr(x_in), c_int(D),
                                             stream_ptr()), "clava_train_draw_qsample_f32")
    idx0, idxt = _noised_categories(diffusion, x, x_in, t, None, seed, step_counter, dense=dense)
    return t, noise, x_in, idx0, idxt


def loss_and_grad(diffusion, model_out, noise, idx0, idxt, t, aux_stream=None, split_out=None, overflow=None):
    """Both loss means (0-dim device tensors, or None for an absent half) and d(loss_multi + loss_gauss)/d(model_out).
    `aux_stream`: the reductions to the two means (nothing in the step waits for them) run there, behind an event; the
    caller joins that stream before the results are read (the captured step: a stream the backward joins at its end).
    `split_out` (a zero-initialised ops.SplitTensor [B, D] with its shift set; wide tables): the two kernels write the
    gradient there as split rows, the operand of the head's backward products, and no float32 gradient exists (the
    third result is `split_out`): saves the 137 MB float32 image and its pack launch at the scale-sweep shape."""
    B, D = model_out.shape
    n = diffusion.num_numerical_features
    dev = model_out.device
    if split_out is not None:
        if (split_out.rows, split_out.cols) != (B, D):
            raise _lib.ClavaHipError("loss_and_grad: `split_out` has another shape")
        d_out, d_num, d_cat, ld_d = split_out, None, None, 0
    else:
        # rows padded to 16 bytes (the pad columns are never read as data): float4 loads in the bias-gradient column sums
        d_out = torch.empty((B, (D + 3) // 4 * 4), device=dev, dtype=torch.float32)[:, :D]
        d_num, d_cat, ld_d = d_out, d_out[:, n:], d_out.stride(0)
    sp = (ptr(split_out.buf), c_ll(split_out.pitch), c_int(split_out.shift)) if split_out is not None else (ptr(None), c_ll(0), c_int(0))
    ld_out = model_out.stride(0)                  # the wide-table head writes rows padded to 16 bytes
    loss_multi = loss_gauss = None
    if n > 0:
        if diffusion.gaussian_loss_type != "mse":
            raise NotImplementedError("gaussian_loss_type='kl' is not implemented on the HIP path (every config uses 'mse')")
        rows = torch.empty(B, device=dev, dtype=torch.float32)
        check(lib().clava_gauss_mse_loss_split_f32(c_int(B), c_int(n), ptr(model_out), c_int(ld_out), ptr(noise), ptr(rows), ptr(d_num), c_int(ld_d),
                                                   *sp, ptr(overflow), stream_ptr()), "clava_gauss_mse_loss_split_f32")
        loss_gauss = _device_sum(rows, 1.0 / B, aux_stream)
    if diffusion.has_cat:
        if diffusion.multinomial_loss_type != "vb_stochastic" or diffusion.parametrization != "x0":
            raise NotImplementedError("only multinomial_loss_type='vb_stochastic' with parametrization='x0' is implemented")
        m, sk = diffusion._m, diffusion._sum_k
        terms = torch.empty((B, m), device=dev, dtype=torch.float32)
        sched = diffusion._c_schedule()
        pt = torch.tensor(1.0, dtype=torch.float32) / diffusion.num_timesteps        # ones_like(t).float() / T  (:592)
        check(lib().clava_multinomial_loss_split_f32(ctypes.byref(sched), c_int(B), c_int(m), c_int(sk), ptr(diffusion._offsets_i32),
                                                     ptr(diffusion._log_k), ptr(model_out[:, n:]), c_int(ld_out), ptr(idx0), ptr(idxt), ptr(t),
                                                     c_float(float(pt)), ptr(terms), ptr(d_cat), c_int(ld_d), *sp, c_int(n), ptr(overflow),
                                                     stream_ptr()), "clava_multinomial_loss_split_f32")
        loss_multi = _device_sum(terms, 1.0 / (m * B), aux_stream)
    return loss_multi, loss_gauss, d_out


class _MixedLossFn(torch.autograd.Function):
    """(loss_multi, loss_gauss) from the denoiser output; backward hands the fused gradient back to autograd."""

    @staticmethod
    def forward(ctx, model_out, diffusion, noise, idx0, idxt, t):
        lm, lg, d_out = loss_and_grad(diffusion, model_out.detach().contiguous(), noise, idx0, idxt, t)
        ctx.save_for_backward(d_out)
        ctx.n = diffusion.num_numerical_features
        zero = torch.zeros((), device=model_out.device)
        return (lm if lm is not None else zero), (lg if lg is not None else zero)

    @staticmethod
    def backward(ctx, g_multi, g_gauss):
        (d_out,) = ctx.saved_tensors
        # incoming scales applied on the device (the training loop sums the two losses unweighted, so both are 1):
        # reading them back would cost two host synchronisations per backward
        d = torch.empty_like(d_out)
        torch.mul(d_out[:, :ctx.n], g_gauss, out=d[:, :ctx.n])
        torch.mul(d_out[:, ctx.n:], g_multi, out=d[:, ctx.n:])
        return d, None, None, None, None, None


class _DenoiserFn(torch.autograd.Function):
    """MLPDiffusion.forward under autograd: backward produces every parameter gradient with the HIP GEMMs."""

    @staticmethod
    def forward(ctx, net, x_in, t, *params):
        out, saved = denoiser_forward(net, x_in.detach().float().contiguous(), t)
        ctx.net, ctx.saved = net, saved
        ctx.want_dx = x_in.requires_grad
        return out

    @staticmethod
    def backward(ctx, d_out):
        net = ctx.net
        layout = FlatLayout(net)
        slabs = torch.empty((GRAD_SPLITS, layout.total), device=d_out.device, dtype=torch.float32)
        dx = denoiser_backward(net, ctx.saved, d_out.contiguous(), layout, slabs, want_dx=ctx.want_dx)
        flat = reduce_slabs(slabs)
        grads = [layout.view(flat, name) for name in layout.names]
        return (None, dx, None, *grads)


def mlp_diffusion_autograd(net, x, timesteps, y=None):
    if getattr(net, "is_y_cond", "none") == "embedding" and y is not None:
        raise NotImplementedError("is_y_cond='embedding' has no HIP training path (ClavaDDPM trains with 'none')")
    require_cuda(x, timesteps)
    return _DenoiserFn.apply(net, x, timesteps.long().contiguous(), *param_list(net))


def draw(diffusion, x, draws):
    """Timesteps, Gaussian noise, categorical uniforms and the Philox seed of one training step.
    Injected (`draws`) for parity tests; otherwise t and noise come from torch's device generator (the reference's
    `torch.randint` / `torch.randn_like`, :590, :647) and the categorical uniforms from the in-kernel generator."""
    B = x.shape[0]
    n = diffusion.num_numerical_features
    dev = x.device
    if draws is not None:
        t = draws["t"].to(dev).long().contiguous()
        noise = draws["noise"].to(dev).float().contiguous() if n > 0 else None
        u = draws["u"].to(dev).float().contiguous() if diffusion.has_cat else None
        return t, noise, u, 0
    t = torch.randint(0, diffusion.num_timesteps, (B,), device=dev).long()
    noise = torch.randn((B, n), device=dev, dtype=torch.float32) if n > 0 else None
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if diffusion.has_cat else 0
    return t, noise, None, seed


def mixed_loss(diffusion, x, out_dict, draws=None):
    """Compat path of `mixed_loss` (`:636-675`): differentiable (loss_multi.mean(), loss_gauss.mean())."""
    require_cuda(x)
    t, noise, u, seed = draw(diffusion, x, draws)
    x_in, idx0, idxt = noised_input(diffusion, x, t, noise, u, seed)
    model_out = diffusion._denoise_fn(x_in, t, **out_dict)
    lm, lg = _MixedLossFn.apply(model_out, diffusion, noise, idx0, idxt, t)
    # an absent half is a CPU `zeros((1,)).mean()` in the reference (:664-665); keep the 0-dim shape, stay on the device
    return lm, lg
