"""Training of the cluster-guidance classifier on the HIP library.

Mirrors the loop of `train_classifier` (`pipeline_utils.py:783-817`) around `numerical_forward_backward_log`
(`:655-700`) for the reference `Classifier` in train mode (`:584-630`):

    t ~ uniform{0..T-1};  x_t = gaussian_q_sample(x, t)                                     (:676-677)
    logits = classifier(x_t, t)     Linear -> ReLU/LeakyReLU -> BatchNorm1d(batch stats) -> Dropout(0.5) blocks
    loss = cross_entropy(logits, labels, 'none');  acc@1, acc@5                             (:683-693)
    loss.mean().backward(loss.mean())       <- the reference back-propagates the mean loss scaled by ITSELF (:696-700)
    AdamW(lr, weight_decay=0.01).step()                                                     (:766, :801)

One step = q-sample kernel -> GEMMs with fused bias/activation -> BatchNorm+Dropout kernels -> cross-entropy kernels
-> backward (BatchNorm/Dropout/activation in two kernels per block, input / weight / bias gradients on the GEMM and
column-sum kernels, all writing gradient slabs) -> ONE fused AdamW kernel over a flat parameter buffer.  Nothing is read
back to the host inside a step: the mean loss stays on the device (also as the gradient scale), Dropout masks come from
the counter-based generator keyed by a device-resident step counter, so the whole step replays as one hipGraph.
"""
import contextlib
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib, ops
from . import parallel as _par
from . import training as _tr
from ._lib import ACT_LEAKY, ACT_RELU, c_float, c_int, check, lib, ptr, stream_ptr
from .tab_ddpm.modules import timestep_embedding

c_ll = ctypes.c_longlong
# Batch ranges per weight-gradient product of the classifier step.  Its narrow blocks (128-256 wide) give one product only a
# handful of output tiles, so the batch ranges are the parallelism: 16 ranges of 256 rows at batch 4096 run those products
# in half the time of 8 (0.804 vs 0.840 ms per step; 32: 0.826, the optimizer then reads 32 partial gradients).
GRAD_SPLITS = int(os.environ.get("CLAVA_CLF_GRAD_SPLITS", "16"))
CLF_WGRAD_STREAMS = int(os.environ.get("CLAVA_CLF_WGRAD_STREAMS", "2"))     # streams the weight-gradient products take turns on
# BatchNorm launches write the split rows the next wide product reads (no packing launch in between); 0: separate packs
BN_WRITES_SPLIT = os.environ.get("CLAVA_BN_SPLIT", "1") != "0"
c_u64 = ctypes.c_ulonglong
c_u32 = ctypes.c_uint
BN_MOMENTUM = 0.1


def _blocks(clf):
    """[(linear, act code, slope, batchnorm, dropout p)] of the hidden blocks and the output Linear of `clf.model`."""
    mods = list(clf.model)
    out, j = [], 0
    while j + 3 < len(mods):
        lin, act, bn, dr = mods[j:j + 4]
        relu = isinstance(act, nn.ReLU)
        out.append((j, lin, ACT_RELU if relu else ACT_LEAKY, 0.0 if relu else float(getattr(act, "negative_slope", 0.01)), bn,
                    float(dr.p)))
        j += 4
    return out, j, mods[j]


def _bn_dropout(a, bn, p, keep, seed, step, step_dev, layer, ws, update_running=True, count=True, split_out=None):
    """`split_out` = (SplitTensor, shift, overflow flag): the output also goes out as split rows (next block's operand)."""
    B, C = a.shape
    dev = a.device
    mean = torch.empty(C, device=dev, dtype=torch.float32)
    invstd = torch.empty(C, device=dev, dtype=torch.float32)
    h = torch.empty((B, C), device=dev, dtype=torch.float32)
    xs, xsh, xflag = split_out if split_out is not None else (None, 0, None)
    if xs is not None:
        xs.shift = xsh
    check(lib().clava_bn_dropout_train_split_f32(ptr(a), c_int(a.stride(0)), c_int(B), c_int(C), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                           ptr(bn.running_mean) if update_running else None,
                                           ptr(bn.running_var) if update_running else None, c_float(BN_MOMENTUM if bn.momentum is None else bn.momentum),
                                           c_float(bn.eps), c_float(p), ptr(keep), c_int(keep.stride(0) if keep is not None else 0),
                                           c_u64(seed), c_u32(step), ptr(step_dev), c_u32(layer), ptr(ws), ptr(mean), ptr(invstd),
                                           ptr(h), c_int(h.stride(0)),
                                                 ptr(xs.buf) if xs is not None else None, c_ll(xs.pitch if xs is not None else 0), c_int(xsh),
                                                 ptr(xflag), stream_ptr()), "clava_bn_dropout_train_split_f32")
    if update_running and count:
        bn.num_batches_tracked += 1
    return h, mean, invstd


class ClassifierSplitState:
    """Split-f16 state of one ClassifierTrainer (csrc/split.h; the denoiser's counterpart is training.SplitTrainState).

    The wide blocks of the classifier (input and output width >= 256: 256 -> 512 -> 1024 -> 512 -> 256, 94 % of the step's
    multiply-adds) run their forward, input-gradient and weight-gradient products on the f16 matrix cores at float32-level
    accuracy (three f16 MFMAs per product, 22-bit significands); BatchNorm, Dropout, the narrow blocks and the cross
    entropy stay as they are.  A wide block's input (the previous BatchNorm's output) and its pre-activation gradient are
    packed into split rows by one launch each; all weights of the wide blocks (W and W^T) by one launch per step.
    Plane scales come from a float32 step: every `period`-th step runs on the float32 kernels and records the largest
    magnitude of every weight, block input and gradient; the device overflow flag is read at the same cadence."""

    def __init__(self, period=200):
        self.period = period
        self.shift_w = self.shift_h = self.shift_g = None
        self.since = 0
        self.flag = None
        self.generation = 0
        self.bufs = {}
        self.plans = {}
        self.disabled = False

    def due(self):
        return self.shift_w is None or self.since >= self.period

    @staticmethod
    def wide(blocks):
        return [i for i, b in enumerate(blocks) if b[1].in_features >= 256 and b[1].out_features >= 256 and b[1].in_features % 32 == 0]

    def buf(self, key, rows, cols, device):
        b = self.bufs.get(key + (rows, cols))
        if b is None:
            b = self.bufs[key + (rows, cols)] = ops.split_empty(rows, cols, 0, device)
        return b

    def pack_weights(self, blocks, device):
        """W (forward operand) and W^T (input-gradient operand) of every wide block, one launch (ops.PackPlan).  Plans are
        kept per (weight storage, scale): a captured step holds the address of its plan's descriptor table."""
        idx = self.wide(blocks)
        key = tuple((i, blocks[i][1].weight.data_ptr(), self.shift_w[i]) for i in idx)
        plan = self.plans.get(key)
        if plan is None:
            items = []
            for i in idx:
                w = blocks[i][1].weight.detach()
                items.append((w, self.shift_w[i], self.buf(("w", i), w.shape[0], w.shape[1], device),
                              self.buf(("wt", i), w.shape[1], w.shape[0], device)))
            plan = self.plans[key] = ops.PackPlan(items, device)
        plan.run(self.flag)

    def calibrate_from(self, stats, idx):
        """stats: device vector [max |W_i|..., max |block input_i|..., max |dz_i|...] over the wide blocks `idx`."""
        from .split_nets import TARGET_LOG2, shift_for
        v = stats.cpu().tolist()                 # the window's one host synchronisation
        n = len(idx)
        before = (self.shift_w, self.shift_h, self.shift_g)
        self.shift_w = {i: shift_for(v[k], 10) for k, i in enumerate(idx)}
        self.shift_h = {i: shift_for(v[n + k], TARGET_LOG2) for k, i in enumerate(idx)}
        self.shift_g = {i: shift_for(v[2 * n + k], TARGET_LOG2) for k, i in enumerate(idx)}
        self.since = 0
        if before != (self.shift_w, self.shift_h, self.shift_g):
            self.generation += 1


def classifier_forward_train(clf, x_t, t, keeps=None, seed=0, step=0, step_dev=None, split=None):
    """Train-mode `Classifier.forward` keeping what the backward needs.  `keeps`: per-block uint8 [B, C_i] Dropout keep
    masks (parity tests) or None for the in-kernel generator keyed by (seed, step or *step_dev, block)."""
    w0, b0 = clf.time_embed[0].weight.detach(), clf.time_embed[0].bias.detach()
    w2, b2 = clf.time_embed[2].weight.detach(), clf.time_embed[2].bias.detach()
    B = x_t.shape[0]
    dev = x_t.device
    if _tr.FRONT_FUSED and clf.dim_t == 128 and t.dtype == torch.int64 and x_t.dtype == torch.float32 and x_t.stride(1) == 1:
        # timestep embedding, time MLP and proj(x) + emb in one launch (csrc/train_front.hip: five launches otherwise, each
        # with a dependent-launch gap on the step's one critical chain)
        from .tab_ddpm.modules import _freqs
        te, z1, a1, h = (torch.empty((B, 128), device=dev, dtype=torch.float32) for _ in range(4))
        check(lib().clava_train_front_f32(ptr(t.contiguous()), c_int(B), ptr(_freqs(128, dev)), c_int(128), ptr(x_t), c_ll(x_t.stride(0)),
                                          c_int(x_t.shape[1]), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(clf.proj.weight.detach()),
                                          ptr(clf.proj.bias.detach()), ptr(te), ptr(z1), ptr(a1), ptr(h), None, c_ll(0), c_int(0), None,
                                          stream_ptr()), "clava_train_front_f32")
    else:
        te = timestep_embedding(t, clf.dim_t)
        z1 = ops.linear(te, w0, b0)
        a1 = _tr._silu(z1)
        emb = ops.linear(a1, w2, b2)
        h = ops.linear(x_t, clf.proj.weight.detach(), clf.proj.bias.detach(), rowadd=emb)
    blocks, _, head = _blocks(clf)
    ws = _bn_workspace(max(b[1].out_features for b in blocks), x_t.device)
    hs, acts, stats = [h], [], []
    wide = ClassifierSplitState.wide(blocks) if split is not None else []
    if wide:
        split.pack_weights(blocks, dev)
    xsplits, prepacked = {}, {}
    for i, (j, lin, act, slope, bn, p) in enumerate(blocks):
        if i in wide:
            xs = prepacked.pop(i, None)            # written by the previous block's BatchNorm launch
            if xs is None:
                xs = ops.split_pack(hs[-1], split.shift_h[i], split.flag, out=split.buf(("h", i), B, lin.in_features, dev))
            a = ops.linear_f16x3(xs, split.buf(("w", i), lin.out_features, lin.in_features, dev), lin.bias.detach(),
                                 act="relu" if act == ACT_RELU else "leaky_relu", out_f32=True, overflow=split.flag)
            xsplits[i] = xs
        else:
            a = ops.linear(hs[-1], lin.weight.detach(), lin.bias.detach(), act="relu" if act == ACT_RELU else "leaky_relu")
        keep = keeps[i] if keeps is not None else None
        nxt = None
        if BN_WRITES_SPLIT and i + 1 in wide:
            nxt = (split.buf(("h", i + 1), B, lin.out_features, dev), split.shift_h[i + 1], split.flag)
            prepacked[i + 1] = nxt[0]
        h, mean, invstd = _bn_dropout(a, bn, p, keep, seed, step, step_dev, i, ws, count=False, split_out=nxt)
        acts.append(a)
        stats.append((mean, invstd))
        hs.append(h)
    # `num_batches_tracked += 1` of every BatchNorm (torch's train-mode forward) as ONE multi-tensor launch instead of one
    # tiny kernel per layer on the critical chain
    torch._foreach_add_([b[4].num_batches_tracked for b in blocks], 1)
    logits = ops.linear(hs[-1], head.weight.detach(), head.bias.detach())
    return logits, (x_t, te, z1, a1, hs, acts, stats, xsplits)


def cross_entropy(logits, labels, want_acc=True):
    """Per-row loss, logsumexp, acc@1 and acc@5 indicators (acc@5 only when there are >= 5 classes, `:690`)."""
    B, C = logits.shape
    dev = logits.device
    rows = torch.empty(B, device=dev, dtype=torch.float32)
    lse = torch.empty(B, device=dev, dtype=torch.float32)
    acc1 = torch.empty(B, device=dev, dtype=torch.float32) if want_acc else None
    acc5 = torch.empty(B, device=dev, dtype=torch.float32) if want_acc and C >= 5 else None
    check(lib().clava_cross_entropy_rows_f32(ptr(logits), c_int(logits.stride(0)), ptr(labels), c_int(B), c_int(C), ptr(rows),
                                             ptr(acc1), ptr(acc5), ptr(lse), stream_ptr()), "clava_cross_entropy_rows_f32")
    return rows, lse, acc1, acc5


def _bn_workspace(C, device):
    need = lib().clava_bn_workspace_floats
    need.restype = ctypes.c_longlong
    return torch.empty(int(need(c_int(C))), device=device, dtype=torch.float32)


def classifier_backward(clf, saved, logits, labels, lse, scale, layout, slabs, keeps=None, seed=0, step=0, step_dev=None,
                        splits=GRAD_SPLITS, side=None, split=None, collect=None):
    """Fills `slabs` [splits, layout.total] with the partial gradients of every live parameter for the upstream gradient
    `scale[0] / B` on each row's cross-entropy (`scale`: 0-dim device tensor, the batch-mean loss in the reference)."""
    x_t, te, z1, a1, hs, acts, stats, xsplits = saved
    B, C = logits.shape
    dev = logits.device
    total = layout.total
    dl = torch.empty((B, C), device=dev, dtype=torch.float32)
    check(lib().clava_cross_entropy_grad_f32(ptr(logits), c_int(logits.stride(0)), ptr(labels), ptr(lse), c_int(B), c_int(C), ptr(scale),
                                             c_float(1.0 / B), ptr(dl), c_int(dl.stride(0)), stream_ptr()), "clava_cross_entropy_grad_f32")

    # Weight / bias gradients only consume (dz_l, h_{l-1}); nothing in the input-gradient chain waits for them, so they
    # run on a second stream beside that chain (they are small, latency-bound launches) and are joined at the end.
    main = torch.cuda.current_stream()
    keep_alive = []

    # `side`: one stream or a list of them.  The weight-gradient products of different layers are independent of each other,
    # and none of them fills the chip: taken round robin over two streams they overlap with each other as well (on ONE side
    # stream their sum, 335 us of an 800 us step, was the backward's long pole beside a 290 us input-gradient chain).
    sides = [] if side is None else (list(side) if isinstance(side, (list, tuple)) else [side])
    turn = [0]

    def next_side():
        s_ = sides[turn[0] % len(sides)]
        turn[0] += 1
        return s_

    def param_grads(prefix, dy, inp):
        side = next_side() if sides else None
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            keep_alive.append((dy, inp))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if side is not None:
                side.wait_event(ev)
            wi = layout.index[prefix + ".weight"]
            ops.linear_wgrad(dy, inp, splits=splits, out=slabs[:, layout.offsets[wi]:], slab_stride=total)
            bi = layout.index[prefix + ".bias"]
            check(lib().clava_colsum_slabs_f32(ptr(dy), c_int(dy.stride(0)), c_int(dy.shape[0]), c_int(dy.shape[1]), c_int(splits),
                                               ptr(slabs[:, layout.offsets[bi]:]), c_ll(total), stream_ptr()), "clava_colsum_slabs_f32")

    blocks, jl, head = _blocks(clf)
    ws = _bn_workspace(max(b[1].out_features for b in blocks), dev)
    param_grads(f"model.{jl}", dl, hs[-1])
    dh = ops.linear_dgrad(dl, head.weight.detach())
    for i in range(len(blocks) - 1, -1, -1):
        j, lin, act, slope, bn, p = blocks[i]
        a = acts[i]
        mean, invstd = stats[i]
        keep = keeps[i] if keeps is not None else None
        dz = torch.empty_like(a)
        gi, bi = layout.index[f"model.{j + 2}.weight"], layout.index[f"model.{j + 2}.bias"]
        dzs = None
        if i in xsplits and BN_WRITES_SPLIT:
            dzs = split.buf(("g", i), B, lin.out_features, dev)
            dzs.shift = split.shift_g[i]
        check(lib().clava_bn_dropout_backward_split_f32(
            ptr(dh), c_int(dh.stride(0)), ptr(a), c_int(a.stride(0)), c_int(B), c_int(a.shape[1]), ptr(mean), ptr(invstd),
            ptr(bn.weight.detach()), c_float(p), ptr(keep), c_int(keep.stride(0) if keep is not None else 0), c_u64(seed), c_u32(step),
            ptr(step_dev), c_u32(i), c_int(act), c_float(slope), ptr(ws), c_int(splits), ptr(slabs[:, layout.offsets[gi]:]),
            ptr(slabs[:, layout.offsets[bi]:]), c_ll(total), ptr(dz), c_int(dz.stride(0)), ptr(dzs.buf) if dzs is not None else None,
            c_ll(dzs.pitch if dzs is not None else 0), c_int(dzs.shift if dzs is not None else 0),
            ptr(split.flag) if dzs is not None else None, stream_ptr()), "clava_bn_dropout_backward_split_f32")
        if collect is not None:
            collect[i] = dz.abs().max()
        if i in xsplits:
            # wide block on the f16 matrix cores: dz packed once, read by the weight-gradient product (batch-major operands,
            # bias gradient riding along, on the side stream) and by the input-gradient product
            if dzs is None:
                dzs = ops.split_pack(dz, split.shift_g[i], split.flag, out=split.buf(("g", i), B, lin.out_features, dev))
            side = next_side() if sides else None
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if side is not None:
                    side.wait_event(ev)
                wi, bi_ = layout.index[f"model.{j}.weight"], layout.index[f"model.{j}.bias"]
                ops.linear_wgrad_f16x3_tn(dzs, xsplits[i], splits, slabs[:, layout.offsets[wi]:], total,
                                          bias_out=slabs[:, layout.offsets[bi_]:])
            dh = ops.linear_f16x3(dzs, split.buf(("wt", i), lin.in_features, lin.out_features, dev), out_f32=True, overflow=split.flag)
        else:
            param_grads(f"model.{j}", dz, hs[i])
            dh = ops.linear_dgrad(dz, lin.weight.detach())
    g = dh                                             # gradient w.r.t. proj(x) + emb
    param_grads("proj", g, x_t)
    param_grads("time_embed.2", g, a1)
    if _tr.FRONT_FUSED and clf.dim_t == 128 and g.stride(1) == 1:
        d_z1 = torch.empty_like(z1)               # input gradient of time_embed[2] and SiLU' in one launch
        check(lib().clava_train_time_bwd_f32(ptr(g), c_ll(g.stride(0)), c_int(B), c_int(128), ptr(clf.time_embed[2].weight.detach()),
                                             ptr(z1), ptr(d_z1), stream_ptr()), "clava_train_time_bwd_f32")
    else:
        d_a1 = ops.linear_dgrad(g, clf.time_embed[2].weight.detach())
        d_z1 = _tr._silu_backward(z1, d_a1)
    param_grads("time_embed.0", d_z1, te)
    for s_ in sides:
        done = torch.cuda.Event()
        done.record(s_)
        main.wait_event(done)
    return keep_alive          # tensors the side stream reads: the caller keeps them until the next step


class ClassifierTrainer:
    """The training loop of `train_classifier` (`pipeline_utils.py:759-817`) for an already constructed `Classifier`
    and a schedule-only `GaussianMultinomialDiffusion` (`denoise_fn=None`, `:768-777`).

    `run_loop(train_iter, steps, val_iter=None)` consumes `(x float[B, >= n_num], labels long[B])` batches like the
    reference's data loaders; `history` collects (step, train loss, acc@1, acc@5[, val loss, val acc@1]) rows every
    `log_interval` steps (the reference's logger means, `:816-817`) without synchronising in between."""

    def __init__(self, classifier, diffusion, lr=0.0001, weight_decay=0.01, device=None, remove_first_col=False,
                 data_parallel=True):
        self.classifier = classifier
        self.diffusion = diffusion
        self.lr, self.weight_decay = lr, weight_decay
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self.remove_first_col = remove_first_col
        self.eval_interval, self.log_interval = 5, 10
        self.log_to_logger = False
        dev = next(classifier.parameters()).device if device is None else torch.device(device)
        _lib.require_cuda(next(classifier.parameters()))
        self.device = dev
        # the reference hands classifier.parameters() to AdamW, including the never-called nn.Transformer (`:591-595`);
        # its gradients stay None, so torch skips it: only the live parameters are optimised here
        self.layout = _tr.FlatLayout(classifier, skip_prefixes=("transformer_layer",))
        total = self.layout.total
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self._flattened = False
        self._flatten()
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.slabs = torch.zeros((GRAD_SPLITS, total), device=dev, dtype=torch.float32)
        self.opt_step = 0
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)      # Dropout step counter, advanced on the device
        self._host_step = 0           # host mirror of step_dev (its parity picks the captured step's pinned input buffer)
        self.history = []
        self._use_graph = os.environ.get("CLAVA_TRAIN_GRAPH", "1") != "0"
        self.split_with_draws = False     # tests: take the split-f16 path with injected draws too
        self._window = []
        self._graph = None
        self._side = None
        self._sides = None
        self._keep = None
        self.world = 1
        self.dist = None
        # data parallel (every rank must construct the trainer and call step()): parameters broadcast once, one all-reduce
        # of the flat gradient per step.  data_parallel=False keeps a trainer local to its rank (per-relation sharding,
        # parallel.assign_relations: each rank trains other classifiers).
        if data_parallel and _par.data_parallel_active():
            self.dist = torch.distributed
            self.world = self.dist.get_world_size()
            _par.broadcast_(self.flat, src=0)
            self.grad_flat = torch.zeros(total, device=dev, dtype=torch.float32)

    def _flatten(self):
        if self._flattened:
            return
        live = dict(self.classifier.named_parameters())
        for name in self.layout.names:
            v = self.layout.view(self.flat, name)
            v.copy_(live[name].data)
            live[name].data = v
        self._flattened = True

    def finalize(self):
        """Give every parameter its own storage again (the classifier is pickled with the run, complex_pipeline.py:64)."""
        if self._flattened:
            live = dict(self.classifier.named_parameters())
            for name in self.layout.names:
                live[name].data = live[name].data.clone()
            self._flattened = False

    def load_adam_state(self, exp_avg, exp_avg_sq, steps_done):
        """Resume from recorded Adam moments ({parameter name: tensor}) after `steps_done` optimizer steps."""
        for name in self.layout.names:
            self.layout.view(self.exp_avg, name).copy_(exp_avg[name].to(self.device))
            self.layout.view(self.exp_avg_sq, name).copy_(exp_avg_sq[name].to(self.device))
        self.opt_step = int(steps_done)

    # ------------------------------------------------------------------------------------------------ one step
    def _inputs(self, x, labels, draws):
        d = self.diffusion
        n = d.num_numerical_features
        dev = self.device
        x = x.to(dev)
        if self.remove_first_col:
            x = x[:, 1:]
        x_num = x[:, :n].float().contiguous()
        if not labels.is_cuda and labels.numel():          # the loaders yield CPU labels: check them where it costs nothing
            lo, hi = int(labels.min()), int(labels.max())
            n_out = _blocks(self.classifier)[2].out_features
            if lo < 0 or hi >= n_out:
                raise _lib.ClavaHipError(f"labels must lie in [0, {n_out}): got [{lo}, {hi}]")
        labels = labels.to(dev).long().contiguous()
        return x_num, labels

    def _noised(self, x_num, draws):
        d = self.diffusion
        B, n = x_num.shape
        dev = x_num.device
        if draws is not None:
            t = draws["t"].to(dev).long().contiguous()
            noise = draws["noise"].to(dev).float().contiguous()
        elif _tr.DEVICE_DRAWS and getattr(self, "step_dev", None) is not None:
            # t, noise and the q-sample in one launch from the counter-based generator, keyed like the Dropout masks by
            # (seed, device step counter): torch's generator stays out of the captured step
            t = torch.empty(B, device=dev, dtype=torch.int64)
            noise = torch.empty((B, n), device=dev, dtype=torch.float32)
            x_t = torch.empty((B, n), device=dev, dtype=torch.float32)
            sched = d._c_schedule()
            check(lib().clava_train_draw_qsample_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(x_num), c_ll(x_num.stride(0)),
                                                     c_u64(self.seed ^ 0x5DEECE66D), ctypes.c_uint(0), ptr(self.step_dev), ptr(t), ptr(noise),
                                                     ptr(x_t), c_int(n), stream_ptr()), "clava_train_draw_qsample_f32")
            return x_t, t
        else:
            t = torch.randint(0, d.num_timesteps, (B,), device=dev).long()      # UniformSampler.sample (:676)
            noise = torch.randn((B, n), device=dev, dtype=torch.float32)
        x_t = torch.empty((B, n), device=dev, dtype=torch.float32)
        sched = d._c_schedule()
        check(lib().clava_gauss_qsample_f32(ctypes.byref(sched), c_int(B), c_int(n), ptr(x_num), ptr(noise), ptr(t), ptr(x_t), c_int(n),
                                            stream_ptr()), "clava_gauss_qsample_f32")
        return x_t, t

    def _forward_backward(self, x_num, labels, draws, split=None, collect=None):
        clf = self.classifier
        keeps = None
        if draws is not None and draws.get("keep") is not None:
            keeps, off = [], 0
            k = draws["keep"].to(self.device).to(torch.uint8)
            for b in _blocks(clf)[0]:
                C = b[1].out_features
                keeps.append(k[:, off:off + C].contiguous())
                off += C
        x_t, t = self._noised(x_num, draws)
        B = x_t.shape[0]
        logits, saved = classifier_forward_train(clf, x_t, t, keeps, self.seed, 0, self.step_dev, split=split)
        rows, lse, acc1, acc5 = cross_entropy(logits, labels)
        if self._side is None and os.environ.get("CLAVA_TRAIN_STREAMS", "2") != "1":
            self._side = torch.cuda.Stream(device=self.device)
            self._sides = [self._side] + [torch.cuda.Stream(device=self.device) for _ in range(max(0, CLF_WGRAD_STREAMS - 1))]
        loss = _tr._device_sum(rows, 1.0 / B)          # the backward starts from it (`loss.backward(loss)`, :700)
        # the two accuracies are only reported: summed on the weight-gradient stream, which the backward joins at its end
        a1 = _tr._device_sum(acc1, 1.0 / B, self._side)
        a5 = _tr._device_sum(acc5, 1.0 / B, self._side) if acc5 is not None else None
        self._keep = classifier_backward(clf, saved, logits, labels, lse, loss, self.layout, self.slabs, keeps, self.seed, 0,
                                         self.step_dev, side=self._sides if self._side is not None else None, split=split, collect=collect)
        self.step_dev += 1
        return loss, a1, a5, (saved, logits, rows, lse, acc1, acc5, x_t, t)

    def _host_inputs(self, x, labels):
        """The batch's numeric columns (float32) and labels without moving them to the device (the captured step fetches
        them itself); labels on the host are range-checked like `_inputs` does."""
        n = self.diffusion.num_numerical_features
        if self.remove_first_col:
            x = x[:, 1:]
        x_num = x[:, :n]
        if not labels.is_cuda and labels.numel():
            lo, hi = int(labels.min()), int(labels.max())
            n_out = _blocks(self.classifier)[2].out_features
            if lo < 0 or hi >= n_out:
                raise _lib.ClavaHipError(f"labels must lie in [0, {n_out}): got [{lo}, {hi}]")
        return x_num, labels

    def _adamw_hyper(self, n):
        """The fused optimizer over the whole flat buffer, step scalars read from `_hyper_dev` (a captured step's form)."""
        check(lib().clava_adamw_ema_hyper_f32(ptr(self.flat), None, ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.slabs),
                                              c_int(GRAD_SPLITS), c_ll(self.layout.total), c_float(1.0), c_ll(n), ptr(self._hyper_dev),
                                              ctypes.c_double(0.0), stream_ptr()), "clava_adamw_ema_hyper_f32")

    def _graphed(self, x, labels):
        """The whole step (input fetch, draws, forward, loss, backward, optimizer) as ONE hipGraph launch.  Inputs go through
        two pinned host buffers used alternately (parity of the device step counter) and are fetched by the graph's first
        node (`clava_stage_fetch_f32`): no host-synchronous copy sits between two replays (scripts/train.py has the
        measurements).  Layout of a buffer, in 4-byte words: x_num [B, n] float32 (+ 1 pad word if odd) | labels [B] int64 |
        8 optimizer scalars.  (Labels travel as the int64 the loss kernels read: an int32 -> int64 cast node in front of the
        draws showed up as 40 us of the step's timeline.)"""
        x_num, labels = self._host_inputs(x, labels)
        B, n = x_num.shape
        on_dev = x_num.is_cuda
        sp = self._split_for(B)
        key = (B, n, on_dev, sp.generation if sp is not None else -1)     # a captured step bakes the plane scales in
        graphs = self.__dict__.setdefault("_graphs", {})          # one captured step per batch shape (an epoch's short last batch)
        gr = graphs.get(key)
        if gr is None:
            warm = self.__dict__.setdefault("_graph_warm", set())
            if key not in warm:                                    # first step of a shape runs eagerly (allocator warm-up)
                warm.add(key)
                return None
            if len(graphs) >= 4:
                # least recently used shape out, after its queued replays (their fetch nodes read the pinned buffers from a
                # kernel, which torch's host allocator does not track)
                old = graphs.pop(next(iter(graphs)))
                for ev in old["ev"]:
                    if ev is not None:
                        ev.synchronize()
                if getattr(self, "_graph", None) is old:
                    self._graph = None
            dev = self.device
            y0 = (B * n + 1) & ~1                                   # labels start 8-byte aligned
            h0 = y0 + 2 * B
            nw = h0 + 8
            stage_dev = torch.zeros(nw, device=dev, dtype=torch.float32)
            sx = stage_dev[:B * n].view(B, n)
            sy = stage_dev[y0:h0].view(torch.int64)
            self._hyper_dev = stage_dev[h0:]
            pins = [torch.zeros(nw, dtype=torch.float32).pin_memory() for _ in range(2)]
            lo = h0 if on_dev else 0
            graph = torch.cuda.CUDAGraph()
            # data parallel: the optimizer stays outside (the all-reduce sits between the backward and the update) and the
            # capture tolerates the process group's watchdog thread (thread-local error mode), as in scripts/train.py
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if self.dist is not None else "global"):
                check(lib().clava_stage_fetch_f32(ptr(pins[0][lo:]), ptr(pins[1][lo:]), ptr(self.step_dev), ptr(stage_dev[lo:]),
                                                  c_ll(nw - lo), stream_ptr()), "clava_stage_fetch_f32")
                out = self._forward_backward(sx, sy, None, split=sp)
                if self.dist is None:
                    self._adamw_hyper(self.layout.total)
            gr = graphs[key] = {"key": key, "graph": graph, "x": sx, "y": sy, "y0": y0, "out": out, "pins": pins, "ev": [None, None],
                                "stage": stage_dev, "hyper": self._hyper_dev,
                                "hyper_host": [(ctypes.c_float * 8).from_address(p.data_ptr() + 4 * h0) for p in pins]}
        else:
            graphs[key] = graphs.pop(key)        # most recently used last
        self._graph = gr
        self._hyper_dev = gr["hyper"]
        par = self._host_step & 1
        if gr["ev"][par] is not None:
            gr["ev"][par].synchronize()          # the replay two steps back, the last reader of this pinned buffer
        check(lib().clava_adamw_hyper_f32(ctypes.c_double(self.lr), ctypes.c_double(self.betas[0]), ctypes.c_double(self.betas[1]),
                                          ctypes.c_double(self.eps), ctypes.c_double(self.weight_decay), c_ll(self.opt_step + 1),
                                          gr["hyper_host"][par]), "clava_adamw_hyper_f32")
        if on_dev:
            gr["x"].copy_(x_num)
            gr["y"].copy_(labels)
        else:
            pin = gr["pins"][par]
            pin[:B * n].view(B, n).copy_(x_num)
            pin[gr["y0"]:gr["y0"] + 2 * B].view(torch.int64).copy_(labels)
        gr["graph"].replay()
        ev = torch.cuda.Event()
        ev.record()
        gr["ev"][par] = ev
        self._host_step += 1
        return gr["out"][:3]

    def step(self, x, labels, draws=None):
        """One training iteration (`:791-801`).  Returns device scalars (mean loss, acc@1, acc@5 or None)."""
        if not self.classifier.training:
            raise _lib.ClavaHipError("ClassifierTrainer.step needs classifier.train()")
        if not self._flattened:
            self._flatten()
        from . import split_nets
        out = None
        B_ = len(x)
        sp = self._split_for(B_) if (draws is None or self.split_with_draws) else None
        calib = sp is not None and sp.due()
        if sp is not None:
            self._window.append((x, labels, draws))
        if calib:
            self._calibration_step(x, labels, draws, sp)
            return self._last_out
        if sp is not None:
            sp.since += 1
        if draws is None and self._use_graph and (self.dist is None or os.environ.get("CLAVA_TRAIN_GRAPH_DP", "1") != "0"):
            try:
                out = self._graphed(x, labels)
            except RuntimeError as e:
                if self.dist is None:
                    raise
                import warnings
                warnings.warn(f"graph capture next to the process group failed ({e}); data-parallel steps stay eager")
                self._use_graph = False
                torch.cuda.synchronize()
            if out is not None and self.dist is None:      # the optimizer ran inside the graph
                self.opt_step += 1
                split_nets.invalidate()          # parameters and BatchNorm statistics were rewritten through raw pointers
                return out
        if out is None:
            x_num, labels = self._inputs(x, labels, draws)
            out = self._forward_backward(x_num, labels, draws, split=sp)[:3]
            self._host_step += 1
        return self._finish_step(out)

    def _finish_step(self, out):
        """All-reduce (data parallel) and the fused AdamW over the gradient slabs of an eager step."""
        from . import split_nets
        self.opt_step += 1
        if self.dist is not None:
            _tr.reduce_slabs(self.slabs, 1.0 / self.world, out=self.grad_flat)
            _par.allreduce_sum_(self.grad_flat)
            g, S = self.grad_flat, 1
        else:
            g, S = self.slabs, GRAD_SPLITS
        check(lib().clava_adamw_ema_f32(ptr(self.flat), None, ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(g), c_int(S),
                                        c_ll(self.layout.total), c_float(1.0), c_ll(self.layout.total), ctypes.c_double(self.lr),
                                        ctypes.c_double(self.betas[0]), ctypes.c_double(self.betas[1]), ctypes.c_double(self.eps),
                                        ctypes.c_double(self.weight_decay), c_ll(self.opt_step), ctypes.c_double(0.0), stream_ptr()),
              "clava_adamw_ema_f32")
        split_nets.invalidate()      # parameters and BatchNorm statistics were rewritten through raw pointers
        return out

    # ---- split-f16 products of the wide blocks -----------------------------------------------------------------------------
    def _split_for(self, B):
        """The split state when a `B`-row step may use it: split-f16 layer kernels selected (CLAVA_GEMM), whole 32-row blocks
        for every gradient slab, default LeakyReLU slope (the split kernel's), not switched off after an overflow."""
        from . import split_nets
        sp = self.__dict__.get("_split")
        if sp is None:
            sp = self._split = ClassifierSplitState(int(os.environ.get("CLAVA_SPLIT_PERIOD", "200")))
            self._window = []
            self._snap = None
            blocks = _blocks(self.classifier)[0]
            sp.usable = bool(ClassifierSplitState.wide(blocks)) and all(b[2] == ACT_RELU or abs(b[3] - 0.01) < 1e-12 for b in blocks)
        ok = (sp.usable and not sp.disabled and split_nets.get_gemm_mode() == "f16x3" and os.environ.get("CLAVA_CLF_SPLIT", "1") != "0"
              and B % 32 == 0 and _tr.wgrad_split_ok(B, GRAD_SPLITS))
        return sp if ok else None

    def _state_snapshot(self):
        clf = self.classifier
        return {"flat": self.flat.clone(), "m": self.exp_avg.clone(), "v": self.exp_avg_sq.clone(), "opt_step": self.opt_step,
                "step_dev": self.step_dev.clone(), "host_step": self._host_step,
                "buffers": [b.detach().clone() for b in clf.buffers()]}

    def _state_restore(self, snap):
        self.flat.copy_(snap["flat"]); self.exp_avg.copy_(snap["m"]); self.exp_avg_sq.copy_(snap["v"])
        self.opt_step = snap["opt_step"]
        self.step_dev.copy_(snap["step_dev"])
        self._host_step = snap["host_step"]
        for b, v in zip(self.classifier.buffers(), snap["buffers"]):
            b.copy_(v)
        from . import split_nets
        split_nets.invalidate()

    def _calibration_step(self, x, labels, draws, sp):
        """Every `period`-th step: read the overflow flag of the window that ends here, then run this step on the float32
        kernels and take the plane scales of the next window from its tensors.  A raised flag means a magnitude grew more
        than 60-fold within one window: the window is re-run on the float32 kernels from the snapshot taken at its start
        (same batches, same Dropout / noise draws: they are keyed by the restored step counter) and the split path stays
        off for this trainer."""
        import warnings
        if sp.flag is None:
            sp.flag = ops.overflow_flag(self.device)
        elif int((_par.allreduce_max_(sp.flag.clone()) if self.dist is not None else sp.flag).item()) != 0 and self._snap is not None:
            warnings.warn("split-f16 classifier training left the float16 range; re-running the window on the float32 kernels")
            sp.disabled = True
            self._state_restore(self._snap)
            window, self._window = self._window, []
            self.overflow_recoveries = getattr(self, "overflow_recoveries", 0) + 1
            for (wx, wl, wd) in window:
                self._last_out = self.step(wx, wl, wd)
            return
        self._snap = self._state_snapshot()
        self._window = [(x, labels, draws)]
        blocks = _blocks(self.classifier)[0]
        idx = ClassifierSplitState.wide(blocks)
        collect = {}
        x_num, lab = self._inputs(x, labels, draws)
        out = self._forward_backward(x_num, lab, draws, split=None, collect=collect)
        self._host_step += 1
        saved = out[3][0]
        hs = saved[4]
        stats = torch.stack([blocks[i][1].weight.detach().abs().max() for i in idx] + [hs[i].abs().max() for i in idx] +
                            [collect[i] for i in idx])
        self._last_out = self._finish_step(out[:3])
        sp.calibrate_from(stats, idx)
        if len(self.__dict__.get("_graphs", {})) and sp.generation:      # captured steps of older scale generations are dead
            for k in [k for k in self._graphs if k[3] != sp.generation]:
                old = self._graphs.pop(k)
                for ev in old["ev"]:
                    if ev is not None:
                        ev.synchronize()

    def gradients(self):
        """{name: gradient} of the last step (sum of the slabs), for tests."""
        flat = _tr.reduce_slabs(self.slabs)
        return {n: self.layout.view(flat, n).clone() for n in self.layout.names}

    def evaluate(self, x, labels, draws=None):
        """The `prefix='val'` call (`:803-814`): eval-mode forward on a noised batch; (mean loss, acc@1, acc@5 or None)."""
        clf = self.classifier
        was = clf.training
        clf.eval()
        try:
            x_num, labels = self._inputs(x, labels, draws)
            x_t, t = self._noised(x_num, draws)
            logits = clf(x_t, t)
            rows, _, acc1, acc5 = cross_entropy(logits, labels)
            B = x_t.shape[0]
            return (_tr._device_sum(rows, 1.0 / B), _tr._device_sum(acc1, 1.0 / B),
                    _tr._device_sum(acc5, 1.0 / B) if acc5 is not None else None)
        finally:
            clf.train(was)

    def run_loop(self, train_iter, steps, val_iter=None):
        """`for step in range(classifier_steps)` of `:785-817`; leaves the classifier in eval mode like `:820`."""
        self.classifier.train()
        dev = self.device
        sums = torch.zeros(3, device=dev, dtype=torch.float64)
        vsums = torch.zeros(2, device=dev, dtype=torch.float64)
        n = nv = 0
        for step in range(steps):
            x, y = next(train_iter)
            loss, a1, a5 = self.step(x, y)
            sums[0] += loss.double()
            sums[1] += a1.double()
            if a5 is not None:
                sums[2] += a5.double()
            n += 1
            if val_iter is not None and not step % self.eval_interval:
                vx, vy = next(val_iter)
                vl, va, _ = self.evaluate(vx, vy)
                vsums[0] += vl.double()
                vsums[1] += va.double()
                nv += 1
            if not step % self.log_interval:                       # logger.dumpkvs(): means since the last dump
                row = [step] + (sums / n).tolist()
                if nv:
                    row += (vsums / nv).tolist()
                self.history.append(row)
                if self.log_to_logger:                             # the reference's table (`:786-790, 816-817`), on request
                    from .tab_ddpm import logger
                    logger.logkv("step", step)
                    logger.logkv("samples", (step + 1) * len(x))
                    for k, v in zip(("train_loss", "train_acc@1", "train_acc@5", "val_loss", "val_acc@1"), row[1:]):
                        logger.logkv(k, v)
                    logger.dumpkvs()
                sums.zero_()
                vsums.zero_()
                n = nv = 0
        self.finalize()
        self.classifier.eval()
        return self.classifier
