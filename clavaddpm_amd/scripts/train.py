"""`Trainer` with the reference's constructor, attributes and loop semantics (`scripts/train.py:7-71`), fused on the HIP library.

Same observable behaviour: AdamW(lr, weight_decay) on the denoiser, linear LR anneal applied AFTER each step (so
step 0 runs at the full lr, `:24-28, 53`), EMA(0.999) of the denoiser updated every step (`:69`), `loss_history`
rows every `log_every` steps with losses rounded to 4 decimals (`:59-67`).

What differs by design:
  * one step = q-sample kernels -> denoiser forward -> fused loss/gradient kernels -> backward GEMMs writing gradient
    slabs -> ONE fused AdamW+EMA kernel over a flat parameter buffer (the reference: autograd graph, ~60 optimizer
    launches, 40 EMA launches);
  * no host synchronisation per step: the reference calls `.item()` twice per step (`:56-57`); here the running loss
    sums stay on the device and are read once every `log_every` steps;
  * split-f16 overflow (csrc/split.h): the overflow flag of the hidden-block kernels is read once per calibration window
    (200 steps); if it is raised, `run_loop` restores the snapshot it took at the window's start (parameters, EMA, Adam
    moments, step counters, RNG states, loss sums) and re-runs the window's batches on the float32 kernels;
  * data parallelism (new): when torch.distributed is initialised, the flat gradient is averaged
    with one RCCL all-reduce per step; parameters are broadcast from rank 0 at construction.
"""
import ctypes
import os
from copy import deepcopy

import numpy as np
import pandas as pd
import torch

from .. import _lib
from .. import parallel as _par
from .. import training as _tr
from .._lib import c_float, c_int, check, lib, ptr, stream_ptr
from ..training import c_ll


class Trainer:
    def __init__(self, diffusion, train_iter, lr, weight_decay, steps, device=torch.device('cuda:0')):
        self.diffusion = diffusion
        self.train_iter = train_iter
        self.steps = steps
        self.init_lr = lr
        self.lr = lr
        self.weight_decay = weight_decay
        self.device = device
        self.loss_history = pd.DataFrame(columns=['step', 'mloss', 'gloss', 'loss'])
        self.log_every = 100
        self.print_every = 500
        self.ema_every = 1000
        self.betas = (0.9, 0.999)
        self.eps = 1e-8
        self.ema_rate = 0.999
        self.opt_step = 0
        import os
        self._use_graph = os.environ.get("CLAVA_TRAIN_GRAPH", "1") != "0"

        net = diffusion._denoise_fn
        # this loop calls the HIP forward / backward directly, so MLPDiffusion.forward's own guards never run: refuse
        # here what the fused step does not compute instead of training silently without it
        if any(b.dropout.p > 0 for b in net.mlp.blocks):
            raise NotImplementedError("dropout > 0 is not implemented on the HIP training path (every ClavaDDPM config uses 0.0)")
        if getattr(net, "is_y_cond", "none") == "embedding" or hasattr(net, "label_emb"):
            raise NotImplementedError("is_y_cond='embedding' (label_emb) has no HIP training path (ClavaDDPM trains with 'none', "
                                      "pipeline_modules.py:52)")
        self.layout = _tr.FlatLayout(net)
        dev = next(net.parameters()).device
        # training draws (t, Gaussian noise, categorical uniforms) come from the library's counter-based generator, keyed by
        # a seed taken once from torch's CPU generator (torch.manual_seed makes a run repeatable) and a device step word
        # advanced once per step, eager or replayed; CLAVA_TRAIN_DRAWS=torch draws t and the noise with torch instead
        self._draw_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._step_word = torch.zeros(1, device=dev, dtype=torch.int32)
        self._host_word = 0          # host mirror of the step word (its parity picks the captured step's pinned input buffer)
        total = self.layout.total
        # one flat buffer per role; while training, the module's parameters are views into `flat`
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self._flattened = False
        self._flatten()
        self.world = 1
        self.dist = None
        if _par.data_parallel_active():       # not inside parallel.independent_ranks() (relation-parallel training)
            self.dist = torch.distributed
            self.world = self.dist.get_world_size()
            _par.broadcast_(self.flat, src=0)
            # ranks hold different rows: they must not noise them with the same draws
            self._draw_seed = (self._draw_seed + 0x9E3779B97F4A7C15 * self.dist.get_rank()) % (1 << 62)
        self.ema_model = deepcopy(net)
        self.ema_flat = self.flat.clone()
        for name, p in self.ema_model.named_parameters():
            p.detach_()
            p.data = self.layout.view(self.ema_flat, name)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.slabs = torch.zeros((_tr.GRAD_SPLITS, total), device=dev, dtype=torch.float32)
        self.grad_flat = torch.zeros(total, device=dev, dtype=torch.float32) if self.dist is not None else None

    def _flatten(self):
        if self._flattened:
            return
        for name, p in self.diffusion._denoise_fn.named_parameters():
            v = self.layout.view(self.flat, name)
            v.copy_(p.data)
            p.data = v
        self._flattened = True

    def finalize(self):
        """Give every parameter its own storage again (the whole diffusion object is pickled after training,
        complex_pipeline.py:64,78; views into one 17 MB buffer would be serialised once per parameter)."""
        if not self._flattened:
            return
        for p in list(self.diffusion._denoise_fn.parameters()) + list(self.ema_model.parameters()):
            p.data = p.data.clone()
        self._flattened = False

    def _split_state(self):
        """The split-f16 state when that layer-kernel family is selected and the stack has hidden blocks to run on it."""
        from .. import split_nets
        if split_nets.get_gemm_mode() != "f16x3" or len(self.diffusion._denoise_fn.mlp.blocks) < 2:
            return None
        if getattr(self, "_force_f32", 0) > 0:
            return None
        if getattr(self, "_split", None) is None:
            self._split = _tr.SplitTrainState(getattr(self, "split_period", 200))
            if self.dist is not None:                  # every rank must take the same branch at a calibration
                self._split.flag_reduce = _par.allreduce_max_
        return self._split

    # ---- data parallel: per-layer all-reduce started from the weight-gradient stream -------------
    def _dp_layer_done(self, prefix):
        """Called by the split backward with the weight-gradient stream current, right after layer `prefix`'s gradient
        slabs were enqueued: sum the slabs of that layer's (weight, bias) range into the flat gradient (pre-scaled by
        1/world) and start its all-reduce.  The input-gradient chain of the earlier layers runs meanwhile on the main
        stream; `_optimizer_step` waits for the handles."""
        lo, hi = self._layer_range(prefix)
        _tr.reduce_slabs(self.slabs, 1.0 / self.world, out=self.grad_flat, lo=lo, hi=hi)
        self._dp_pending.append((lo, hi, _par.allreduce_sum_async(self.grad_flat[lo:hi])))

    def _layer_range(self, prefix):
        """[lo, hi) of layer `prefix`'s weight and bias in the flat buffers (the layers tile it exactly)."""
        lay = self.layout
        wi, bi = lay.index[prefix + ".weight"], lay.index[prefix + ".bias"]
        lo = min(lay.offsets[wi], lay.offsets[bi])
        hi = max(lay.offsets[i] + int(np.prod(lay.shapes[i])) for i in (wi, bi))
        return lo, min((hi + 3) // 4 * 4, lay.total)

    # ---- optimizer inside the captured step -----------------------------------------------------------------------------
    def _adamw_range(self, lo, hi):
        """AdamW + EMA over flat[lo:hi) from the gradient slabs, step scalars read from the device (`_hyper_dev`)."""
        check(lib().clava_adamw_ema_hyper_f32(ptr(self.flat[lo:]), ptr(self.ema_flat[lo:]), ptr(self.exp_avg[lo:]), ptr(self.exp_avg_sq[lo:]),
                                              ptr(self.slabs[:, lo:]), c_int(_tr.GRAD_SPLITS), c_ll(self.layout.total), c_float(1.0),
                                              c_ll(hi - lo), ptr(self._hyper_dev), ctypes.c_double(self.ema_rate), stream_ptr()),
              "clava_adamw_ema_hyper_f32")

    def _anneal_lr(self, step):
        frac_done = step / self.steps
        self.lr = self.init_lr * (1 - frac_done)

    def _run_step(self, x, out_dict, draws=None):
        d = self.diffusion
        net = d._denoise_fn
        if not self._flattened:
            self._flatten()
            for name, p in self.ema_model.named_parameters():
                v = self.layout.view(self.ema_flat, name)
                v.copy_(p.data)
                p.data = v
        st = self._split_state()
        # Data parallel: the step up to the gradient slabs replays as ONE graph launch (fetch, draws, forward, losses, backward),
        # then three eager calls: slab reduction into the flat gradient, its all-reduce, AdamW + EMA.  Eager, a step is ~100
        # host calls for ~0.65 ms of GPU work and every rank would run at the host's pace.  The collective stays OUTSIDE the
        # captured region; the capture itself uses the thread-local error mode, because the process group's watchdog thread
        # queries events while this thread captures (under the default global mode such a call invalidates the capture).  A
        # capture that fails anyway switches this Trainer to eager steps for good (CLAVA_TRAIN_GRAPH_DP=0: never capture).
        use_graph = self._use_graph and (self.dist is None or os.environ.get("CLAVA_TRAIN_GRAPH_DP", "1") != "0")
        if (st is not None and not st.due() and draws is None and use_graph):
            try:
                done = self._graphed_split_step(x, st)
            except RuntimeError as e:
                if self.dist is None:
                    raise
                import warnings
                warnings.warn(f"graph capture next to the process group failed ({e}); data-parallel steps stay eager")
                self._use_graph = False
                torch.cuda.synchronize()
                done = None
            if done is not None and done[2]:                  # the optimizer ran inside the graph
                return self._loss_pair(done[0], done[1])
            if done is not None:
                return self._optimizer_step(x, done[0], done[1])
        if st is not None and st.due() and draws is None and use_graph and st.shift_h is not None:
            done = self._graphed_calibration_step(x, st)
            if done is not None:
                return self._loss_pair(done[0], done[1])
        x = x.to(self.device)
        if st is not None and st.due() and getattr(self, "_graph", None) is not None:
            # a calibration step synchronises anyway: check the host's mirror of the step word against the device
            assert int(self._step_word.item()) == self._host_word, (int(self._step_word.item()), self._host_word)
        if draws is None and _tr.DEVICE_DRAWS:
            t, noise, x_in, idx0, idxt = _tr.draw_noised_input(d, x, self._draw_seed, self._step_word)
            self._step_word.add_(1)
            self._host_word += 1
        else:
            t, noise, u, seed = _tr.draw(d, x, draws)
            x_in, idx0, idxt = _tr.noised_input(d, x, t, noise, u, seed)
        if st is None or st.due():
            # float32 step (always when CLAVA_GEMM=f32; every `period`-th step otherwise, to re-fix the plane scales)
            model_out, saved = _tr.denoiser_forward(net, x_in, t)
            loss_multi, loss_gauss, d_out = _tr.loss_and_grad(d, model_out, noise, idx0, idxt, t)
            gs = [None] * len(saved[4]) if st is not None else None
            _tr.denoiser_backward(net, saved, d_out, self.layout, self.slabs, collect=gs)
            if st is not None:
                st.calibrate(net.stack()[0], saved[4], gs, x_in=x_in, d_out=d_out, n_num=d.num_numerical_features)
        else:
            model_out, saved = _tr.denoiser_forward_split(net, x_in, t, st)
            loss_multi, loss_gauss, d_out = _tr.loss_and_grad(d, model_out, noise, idx0, idxt, t,
                                                              split_out=_tr.loss_split_target(st, saved, model_out), overflow=st.flag)
            overlap = self.dist is not None and os.environ.get("CLAVA_DP_OVERLAP", "1") != "0"
            self._dp_pending = [] if overlap else None
            _tr.denoiser_backward_split(net, saved, d_out, self.layout, self.slabs, st,
                                        on_layer_done=self._dp_layer_done if overlap else None)
        return self._optimizer_step(x, loss_multi, loss_gauss)

    # ---- captured steps ---------------------------------------------------------------------------------------------------
    def _staging(self, x):
        """Input staging shared by the captured steps of one batch shape.  The step's inputs -- the batch (as float32) and
        the optimizer's eight step scalars -- live in one device buffer.  The host writes them into one of two PINNED
        buffers, alternating with the parity of the step word, and a graph's first node fetches from the one its parity
        names (clava_stage_fetch_f32).  No host-synchronous copy sits between two replays: the host prepares step i+1
        while step i runs (a pageable `copy_` waited for step i to finish and left the GPU idle ~40 us per step;
        asynchronous copies from pinned memory through the copy engine were slower still, 1.03 vs 0.77 ms/step).  A batch
        already on the device is copied device-to-device in stream order and only the scalars go through the pinned
        buffers."""
        key = (tuple(x.shape), x.is_cuda)
        stages = self.__dict__.setdefault("_stages", {})
        sg = stages.get(key)
        if sg is None:
            # one staging buffer and one set of captured graphs PER batch shape (the reference's loader ends every epoch with
            # a short batch: a single slot would re-capture twice per epoch); at most four shapes are kept
            if len(stages) >= 4:
                # least recently used shape out (`stages` is kept in order of last use).  Its replays may still be queued: the
                # fetch node reads the pinned buffers from a kernel, which torch's host allocator does not track, so the last
                # replays are awaited before the buffers and the graphs go.
                old = stages.pop(next(iter(stages)))
                for ev in old["ev"]:
                    if ev is not None:
                        ev.synchronize()
                if getattr(self, "_graph", None) is old["graph"]:
                    self._graph = None
                if getattr(self, "_cal_graph", None) is old["cal"]:
                    self._cal_graph = None
            B, Dx = x.shape
            n_x = B * Dx
            stage_dev = torch.zeros(n_x + 8, device=self.flat.device, dtype=torch.float32)
            pins = [torch.zeros(n_x + 8, dtype=torch.float32).pin_memory() for _ in range(2)]
            sg = stages[key] = {"key": key, "dev": stage_dev, "x": stage_dev[:n_x].view(B, Dx), "pins": pins, "ev": [None, None],
                                "lo": n_x if x.is_cuda else 0, "n": n_x + 8, "hyper": stage_dev[n_x:], "graph": None, "cal": None,
                                "hyper_host": [(ctypes.c_float * 8).from_address(p.data_ptr() + 4 * n_x) for p in pins]}
            sg["x"].copy_(x)
        else:
            stages[key] = stages.pop(key)                 # most recently used last
        self._hyper_dev = sg["hyper"]                     # what `_adamw_range` bakes into a capture
        return sg

    def _fetch_node(self, sg):
        """The captured step's first node: stage buffer <- the pinned buffer of the step word's parity."""
        lo = sg["lo"]
        check(lib().clava_stage_fetch_f32(ptr(sg["pins"][0][lo:]), ptr(sg["pins"][1][lo:]), ptr(self._step_word), ptr(sg["dev"][lo:]),
                                          c_ll(sg["n"] - lo), stream_ptr()), "clava_stage_fetch_f32")

    def _replay(self, sg, graph, x, opt):
        """Fills this step's pinned buffer (batch + optimizer scalars), replays `graph`, advances the host's step mirror."""
        par = self._host_word & 1
        if sg["ev"][par] is not None:
            sg["ev"][par].synchronize()          # the replay two steps back, the last reader of this pinned buffer
        if opt:
            check(lib().clava_adamw_hyper_f32(ctypes.c_double(self.lr), ctypes.c_double(self.betas[0]), ctypes.c_double(self.betas[1]),
                                              ctypes.c_double(self.eps), ctypes.c_double(self.weight_decay), c_ll(self.opt_step + 1),
                                              sg["hyper_host"][par]), "clava_adamw_hyper_f32")
        if x.is_cuda:
            sg["x"].copy_(x)
        else:
            sg["pins"][par][:x.numel()].view(x.shape).copy_(x)
        graph.replay()
        ev = torch.cuda.Event()
        ev.record()
        sg["ev"][par] = ev
        self._host_word += 1
        if opt:
            self.opt_step += 1
            from .. import split_nets
            split_nets.invalidate()      # the optimizer kernels rewrote the parameters through raw pointers

    def _graphed_split_step(self, x, st):
        """The split-f16 step (input fetch, draws, q-sample, forward, losses, backward into the gradient slabs, optimizer) as
        ONE hipGraph launch.  The step is ~100 small host calls; at 0.65 ms of GPU work per step the Python side is the
        bottleneck otherwise.  The graph is re-captured when a calibration moved a plane scale or the batch shape changed;
        the first split step after that runs eagerly and allocates the buffers the capture then reuses.  Not used with
        injected draws or CLAVA_TRAIN_GRAPH=0.  The captured step draws t, the Gaussian noise and the categorical uniforms
        from the library's counter-based generator, keyed by the Trainer's seed and a device step word that every step
        advances (clava_train_draw_qsample_f32, clava_multinomial_step_counter_f32), like the eager steps;
        CLAVA_TRAIN_DRAWS=torch keeps torch's generator in the graph.  Returns (loss_multi, loss_gauss, optimizer ran)."""
        sg = self._staging(x)
        key = (sg["key"], st.generation)
        gr = sg["graph"]
        if gr is None or gr["key"] != key:
            if sg.get("warm") != key:
                sg["warm"] = key
                return None
            d = self.diffusion
            net = d._denoise_fn
            dev = self.flat.device
            B = x.shape[0]
            static_x = sg["x"]
            # optimizer inside the graph (single process only: data parallel puts an all-reduce between the two halves)
            graph_opt = self.dist is None and os.environ.get("CLAVA_TRAIN_GRAPH_OPT", "1") != "0"
            # wide tables with categorical columns: proj and its weight gradient from (numeric columns, class indices),
            # the [B, sum K] log-one-hot image is never built (csrc/cat_input.hip, csrc/cat_wgrad.hip)
            gather = st.gather_ok(d, net, B)
            if gather:
                st.gather_plan(d, dev)
                st.gather_weights(net.stack()[0][0].detach(), net.stack()[1][0].detach(), d.num_numerical_features)   # allocates
                st.pack_weights(net.stack()[0], True, dev, gather=True)      # builds the packing plan (an upload: not capturable)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if self.dist is not None else "global"):
                st.start_pack(net, d.num_numerical_features + d._sum_k,      # denoiser input width
                              gather_n_num=d.num_numerical_features if gather else None)
                self._fetch_node(sg)
                if _tr.DEVICE_DRAWS:
                    # every draw of the step from the counter-based generator, keyed by (the Trainer's seed, device step word
                    # advanced once per step): the same draws as the eager step, and torch's generator stays out of the graph
                    t, noise, x_in, idx0, idxt = _tr.draw_noised_input(d, static_x, self._draw_seed, self._step_word, dense=not gather)
                else:
                    # t and the Gaussian noise from torch's graph-safe generator; the categorical draws keyed as above
                    t, noise, u, seed = _tr.draw(d, static_x, None)
                    x_in, idx0, idxt = _tr.noised_input(d, static_x, t, noise, u, seed,
                                                        step_counter=self._step_word if d.has_cat else None)
                # the step word advances beside the forward, on the packing stream (joined at the end of the backward)
                drawn = torch.cuda.Event()
                drawn.record()
                side = st.side_stream(dev)
                with torch.cuda.stream(side):
                    side.wait_event(drawn)
                    self._step_word.add_(1)
                model_out, saved = _tr.denoiser_forward_split(net, x_in, t, st, cat=(idxt, d) if gather else None)
                # the loss means are summed on a stream the backward joins at its end: nothing in the step waits for them
                lm, lg, d_out = _tr.loss_and_grad(d, model_out, noise, idx0, idxt, t,
                                                  aux_stream=st.side_stream(dev, 1) if _tr.TAIL_BRANCHES else None,
                                                  split_out=_tr.loss_split_target(st, saved, model_out), overflow=st.flag)
                # The optimizer inside the graph, in two launches: the hidden blocks and the head (99.2 % of the parameters)
                # as soon as their gradients are complete, on the weight-gradient stream -- an HBM-bound kernel beside the
                # small latency-bound products of proj and the time-embedding branch that end the backward -- and those
                # three layers after the join.  (Per-LAYER updates beside the matrix products were measured and are slower,
                # 0.78-0.82 vs 0.74 ms/step: they slow the L2-bound products by more than their own time.)
                cut = self.layout.offsets[self.layout.index["proj.weight"]] if graph_opt and _tr.TAIL_BRANCHES else 0
                assert cut == 0 or all(self.layout.offsets[self.layout.index[n_]] >= cut for n_ in self.layout.names
                                       if n_.startswith(("proj.", "time_embed."))), "parameter order changed"
                _tr.denoiser_backward_split(net, saved, d_out, self.layout, self.slabs, st,
                                            after_blocks=(lambda: self._adamw_range(0, cut)) if cut > 0 else None)
                if graph_opt:
                    self._adamw_range(cut, self.layout.total)
            gr = sg["graph"] = {"key": key, "graph": graph, "x": static_x, "lm": lm, "lg": lg, "keep": (saved, d_out, x_in, t, noise),
                                "opt": graph_opt}
        self._graph = gr                                  # the split graph last replayed (tests look at its tensors)
        self._replay(sg, gr["graph"], x, gr["opt"])
        st.steps_since += 1
        return gr["lm"], gr["lg"], gr["opt"]

    def _graphed_calibration_step(self, x, st):
        """The float32 step that re-fixes the plane scales every `period` steps, as a second captured graph (eager it is ~90
        launches and ~35 torch reductions, ~7 ms per window against 1.5 ms of GPU work): fetch, draws, float32 forward /
        loss / backward, the largest magnitudes of every tensor as one device vector, optimizer.  The window's one host
        synchronisation -- overflow flag and that vector -- follows the replay.  The first calibrations of a shape run
        eagerly.  Returns (loss_multi, loss_gauss) or None."""
        if self.dist is not None or os.environ.get("CLAVA_TRAIN_GRAPH_OPT", "1") == "0" or not _tr.DEVICE_DRAWS or st.flag is None:
            return None
        sg = self._staging(x)
        gr = sg["cal"]
        d = self.diffusion
        net = d._denoise_fn
        if gr is None:
            if not sg.get("cal_warm"):
                sg["cal_warm"] = True
                return None
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._fetch_node(sg)
                t, noise, x_in, idx0, idxt = _tr.draw_noised_input(d, sg["x"], self._draw_seed, self._step_word)
                self._step_word.add_(1)
                model_out, saved = _tr.denoiser_forward(net, x_in, t)
                lm, lg, d_out = _tr.loss_and_grad(d, model_out, noise, idx0, idxt, t)
                gs = [None] * len(saved[4])
                _tr.denoiser_backward(net, saved, d_out, self.layout, self.slabs, collect=gs)
                stats = st.calibration_stats(net.stack()[0], saved[4], gs, x_in, d_out, d.num_numerical_features)
                self._adamw_range(0, self.layout.total)
            gr = sg["cal"] = {"key": sg["key"], "graph": graph, "lm": lm, "lg": lg, "stats": stats, "n": len(saved[4]),
                              "keep": (saved, d_out, x_in, t, noise, gs)}
        self._cal_graph = gr
        self._replay(sg, gr["graph"], x, True)
        st.calibrate_from(gr["stats"], gr["n"], bool(d.num_numerical_features))
        return gr["lm"], gr["lg"]

    def _optimizer_step(self, x, loss_multi, loss_gauss):
        self.opt_step += 1
        pending = getattr(self, "_dp_pending", None)
        self._dp_pending = None
        if self.dist is not None and pending:
            # the layers' all-reduces were started during the backward; they must cover the flat buffer exactly once
            covered = sorted((lo, hi) for lo, hi, _ in pending)
            assert covered[0][0] == 0 and covered[-1][1] == self.layout.total and all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
            for _, _, h in pending:
                if h is not None:
                    h.wait()
            g, S, stride = self.grad_flat, 1, self.layout.total
        elif self.dist is not None:
            _tr.reduce_slabs(self.slabs, 1.0 / self.world, out=self.grad_flat)
            _par.allreduce_sum_(self.grad_flat)              # RCCL sum over ranks of the pre-scaled flat gradient
            g, S, stride = self.grad_flat, 1, self.layout.total
        else:
            g, S, stride = self.slabs, _tr.GRAD_SPLITS, self.layout.total
        check(lib().clava_adamw_ema_f32(ptr(self.flat), ptr(self.ema_flat), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(g), c_int(S),
                                        c_ll(stride), c_float(1.0), c_ll(self.layout.total), ctypes.c_double(self.lr),
                                        ctypes.c_double(self.betas[0]), ctypes.c_double(self.betas[1]), ctypes.c_double(self.eps),
                                        ctypes.c_double(self.weight_decay), c_ll(self.opt_step), ctypes.c_double(self.ema_rate),
                                        stream_ptr()), "clava_adamw_ema_f32")
        from .. import split_nets
        split_nets.invalidate()      # the kernel above rewrote the parameters through raw pointers: packed copies are stale
        return self._loss_pair(loss_multi, loss_gauss)

    def _loss_pair(self, loss_multi, loss_gauss):
        zero = getattr(self, "_zero", None)
        if zero is None:
            zero = self._zero = torch.zeros((), device=self.flat.device)
        return (loss_multi if loss_multi is not None else zero), (loss_gauss if loss_gauss is not None else zero)

    # ---- split-f16 overflow: snapshot at every calibration, replay of the window on the float32 kernels ---------------
    def _snapshot(self, step, sums, curr_count):
        dev = self.flat.device
        return {"step": step, "flat": self.flat.clone(), "ema": self.ema_flat.clone(), "m": self.exp_avg.clone(),
                "v": self.exp_avg_sq.clone(), "opt_step": self.opt_step, "lr": self.lr, "sums": sums.clone(), "count": curr_count,
                "rows": len(self.loss_history), "rng_cpu": torch.get_rng_state(), "rng_dev": torch.cuda.get_rng_state(dev),
                "word": self._step_word.clone(), "host_word": self._host_word,
                "batches": []}

    def _restore(self, snap, sums):
        self.flat.copy_(snap["flat"]); self.ema_flat.copy_(snap["ema"])
        self.exp_avg.copy_(snap["m"]); self.exp_avg_sq.copy_(snap["v"])
        self.opt_step, self.lr = snap["opt_step"], snap["lr"]
        sums.copy_(snap["sums"])
        self.loss_history = self.loss_history.iloc[:snap["rows"]].copy()
        # Only the torch-generator draw mode (CLAVA_TRAIN_DRAWS=torch) consumes torch's device generator in a step; the default
        # draws are keyed by the step word restored below.  The CPU generator is never rewound: the data iterator keeps its
        # position (the window's batches are replayed from the snapshot) and draws its per-epoch shuffle from that generator,
        # so a rewind would repeat a shuffle already drawn inside the window.
        if not _tr.DEVICE_DRAWS:
            torch.cuda.set_rng_state(snap["rng_dev"], self.flat.device)
        self._step_word.copy_(snap["word"])
        self._host_word = snap["host_word"]
        from .. import split_nets
        split_nets.invalidate()
        return snap["step"], snap["count"]

    def run_loop(self):
        step = 0
        dev = self.flat.device
        sums = torch.zeros(2, device=dev, dtype=torch.float64)     # running loss_multi * n, loss_gauss * n (on the device)
        curr_count = 0
        self.overflow_recoveries = 0
        snap = None                                                # start of the calibration window in progress
        replay = []                                                # batches still to be re-run after an overflow
        while step < self.steps:
            if replay:
                x, out_dict = replay.pop(0)
            else:
                x, out_dict = next(self.train_iter)
                out_dict = {'y': out_dict}
            st = self._split_state()                               # None on the float32 kernels (also while re-running a window)
            pending = self._snapshot(step, sums, curr_count) if st is not None and st.due() else None   # this step calibrates
            if snap is not None:
                snap["batches"].append((x, out_dict))
            try:
                batch_loss_multi, batch_loss_gauss = self._run_step(x, out_dict)
            except _lib.SplitOverflowError:
                # the window that just ended ran with a tensor beyond the float16 range: go back to its start, run it on the
                # float32 kernels, then run the step that detected it (which calibrates afresh)
                if snap is None or len(snap["batches"]) < 2:
                    raise
                self.overflow_recoveries += 1
                replay = list(snap["batches"])
                step, curr_count = self._restore(snap, sums)
                self._split.flag.zero_()
                self._force_f32 = len(replay) - 1
                self._graph = None
                for sg_ in getattr(self, "_stages", {}).values():
                    sg_["graph"] = None
                snap = None
                continue
            if pending is not None:
                pending["batches"].append((x, out_dict))
                snap = pending
            if getattr(self, "_force_f32", 0) > 0:
                self._force_f32 -= 1
                if self._force_f32 == 0 and getattr(self, "_split", None) is not None:
                    self._split.steps_since = self._split.period   # the window was re-run in float32: calibrate afresh next
            self._anneal_lr(step)
            curr_count += len(x)
            sums[0] += batch_loss_multi.double() * len(x)
            sums[1] += batch_loss_gauss.double() * len(x)
            if (step + 1) % self.log_every == 0:
                m, g = (sums / curr_count).tolist()                  # the only host sync, once per log_every steps
                mloss = np.around(m, 4)
                gloss = np.around(g, 4)
                if (step + 1) % self.print_every == 0:
                    print(f'Step {(step + 1)}/{self.steps} MLoss: {mloss} GLoss: {gloss} Sum: {mloss + gloss}')
                self.loss_history.loc[len(self.loss_history)] = [step + 1, mloss, gloss, mloss + gloss]
                curr_count = 0
                sums.zero_()
            step += 1
        self.finalize()
