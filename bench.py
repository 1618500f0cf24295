#!/usr/bin/env python3
"""Headline benchmark: synthetic rows/sec of 1000-step cluster-guided reverse diffusion on MI355X.

    python bench.py [--gpus N --steps K --warmup W]                     (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY section 8d "M2"): the movie_lens (movie -> rating) relation, the
largest child table of that config: D_in = n_num = 2 (rating + the sibling relation's cluster column), C = 50
cluster classes, denoiser MLP [512,1024,1024,1024,1024,512], classifier [128,256,512,1024,512,256,128],
sampling batch 20000 rows, classifier_scale 1.0, T = 1000 reverse steps (configs/movie_lens.json; BASELINE.md
section 2).  Weights are PyTorch default init under seed 0, cluster labels uniform - throughput does not depend on
trained values.  One "step" = one full T-step chain over one 20000-row batch on every rank (rows are sharded per
GPU, no collective on the data path: weak scaling).

The JSON line also carries
  roofline     - the dominant kernel, gemm_f16x3_kernel<256,320,2,2> (split-f16: three v_mfma_f32_16x16x32_f16 per
                 float32 product; the 1024-wide layers of the denoiser, the classifier and its input gradient):
                 algorithmic FLOPs (2*M*N*K of the float32 product) per launch over its mean launch duration, both from
                 HIP events recorded on the launch stream (clava_prof_*) during a serial replay of one chain, against
                 the 2516.8 TFLOP/s dense f16 matrix peak of MI355X.  `frac` is that algorithmic ratio;
                 `frac_of_split_ceiling` divides by peak/3 (the most a 3-MFMA-per-product kernel can reach) and
                 `mfma_issue_frac` is the share of matrix-core issue slots the kernel fills;
  cpu_baseline - the CPU oracle (a torch-CPU restatement of the reference path, pinned to the reference's golden
                 vectors) timed on this box's host cores on a bounded sample of the same workload, once on the
                 process's CPU share and once on one thread; `train.cpu_baseline` does the same for the training step.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2516.8   # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense" = 16 x the f32 matrix rate
DENOISER_LAYERS = [512, 1024, 1024, 1024, 1024, 512]
CLASSIFIER_LAYERS = [128, 256, 512, 1024, 512, 256, 128]


def build_workload(device, n_num, n_clusters, T, fit_steps):
    """Networks of the workload.  The denoiser is fitted for `fit_steps` steps of the library's own Trainer to a synthetic
    N(0,1) table (the marginals of a quantile-normalised ClavaDDPM table, lib/data.py:217-223): an untrained eps-network
    makes the reverse chain diverge geometrically (|x| ~ 1e3 after 50 steps), which no real run does and which would
    push the split-f16 path onto its float32 fallback.  Throughput does not depend on the values otherwise."""
    import numpy as np
    from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
    from clavaddpm_amd.scripts.train import Trainer
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
    torch.manual_seed(0)
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(DENOISER_LAYERS), "dropout": 0.0}).to(device)
    clf = Classifier(d_in=n_num, d_out=n_clusters, dim_t=128, hidden_sizes=list(CLASSIFIER_LAYERS)).to(device).eval()
    diff = GaussianMultinomialDiffusion(np.array([0]), n_num, net, num_timesteps=T, scheduler="cosine",
                                        device=torch.device(device)).to(device)
    if fit_steps > 0:
        g = torch.Generator().manual_seed(7)
        table = torch.randn(64 * 4096, n_num, generator=g)
        y = torch.zeros(4096, dtype=torch.long)

        def rows():
            i = 0
            while True:
                yield table[i * 4096:(i + 1) * 4096], y
                i = (i + 1) % 64

        net.train(); diff.train()
        tr = Trainer(diff, rows(), lr=1e-3, weight_decay=0.0, steps=fit_steps, device=torch.device(device))
        tr.run_loop()
    net.eval(); diff.eval()
    return diff, net, clf, make_cond_fn(clf, 1.0)


def run_chain(diff, cond_fn, ys, seed, row_offset):
    """One full guided chain over len(ys) rows, state left on the device (the caller synchronises)."""
    x, idx = diff._initial_state(len(ys), None, seed, row_offset)
    keep = diff._run_chain(x, idx, cond_fn, ys, None, seed, row_offset)
    return x, keep


def cpu_share():
    """Threads this process may use: its scheduler affinity, capped by OMP_NUM_THREADS when the box sets one (the GPU box
    exports 16 = one GPU's share of a machine whose os.cpu_count() is many times that)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(n, omp) if omp > 0 else n)


def cpu_baseline(n_num, n_clusters, T, batch, n_steps, threads):
    """Time `n_steps` guided reverse steps of the CPU oracle at the bench shapes; returns rows/s for a T-step chain."""
    import numpy as np
    from oracle import chain, guidance, network, schedule
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from weights import classifier_shapes, mlp_shapes, seeded_state
    torch.set_num_threads(threads)
    tb = schedule.tables("cosine", T)
    p = seeded_state(mlp_shapes(n_num, DENOISER_LAYERS), 1)
    cp = seeded_state(classifier_shapes(n_num, n_clusters, CLASSIFIER_LAYERS), 2)
    g = torch.Generator().manual_seed(0)
    ys = torch.randint(0, n_clusters, (batch, 1), generator=g)
    nz = {"z0": torch.randn(batch, n_num, generator=g), "noise": torch.randn(n_steps + 1, batch, n_num, generator=g).double()}
    grad_fn = lambda x, t: guidance.cond_grad(cp, x, t, ys, 1.0)
    with torch.no_grad():
        chain.conditional_sample(tb, T, p, n_num, grad_fn, nz, steps=1)            # warm-up step
        t0 = time.perf_counter()
        chain.conditional_sample(tb, T, p, n_num, grad_fn, nz, steps=n_steps)
        dt = time.perf_counter() - t0
    return batch / (dt / n_steps * T), dt


def train_cpu_baseline(n_num, T, batch, n_steps, threads):
    """`n_steps` training steps of the CPU oracle (oracle.trainer.run_loop: mixed_loss, autograd backward, AdamW, EMA - the
    reference's loop, scripts/train.py:30-71) at the bench shape; returns steps/s."""
    from oracle import schedule, trainer
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from weights import mlp_shapes, seeded_state
    torch.set_num_threads(threads)
    tb = schedule.tables("cosine", T)
    p = seeded_state(mlp_shapes(n_num, DENOISER_LAYERS), 1)
    g = torch.Generator().manual_seed(3)
    mk = lambda: {"t": torch.randint(0, T, (batch,), generator=g), "noise": torch.randn(batch, n_num, generator=g), "u": None}
    xs = [torch.randn(batch, n_num, generator=g) for _ in range(n_steps + 1)]
    draws = [mk() for _ in range(n_steps + 1)]
    trainer.run_loop(tb, T, p, xs[:1], draws[:1], n_num, [], 6e-4, 1e-5, 10 ** 6)
    p = {k: v.detach() for k, v in p.items()}
    t0 = time.perf_counter()
    trainer.run_loop(tb, T, p, xs[1:], draws[1:], n_num, [], 6e-4, 1e-5, 10 ** 6)
    dt = time.perf_counter() - t0
    return n_steps / dt, dt


def train_bench(device, n_num, T, batch, steps, warmup, barrier):
    """Secondary metric of BASELINE.json: denoiser training steps/s (mixed_loss + backward + AdamW + EMA, batch 4096 per
    GPU, data-parallel all-reduce of the flat gradient when several ranks run)."""
    import numpy as np
    from clavaddpm_amd.scripts.train import Trainer
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
    torch.manual_seed(0)
    net = MLPDiffusion(n_num, 0, "none", {"d_layers": list(DENOISER_LAYERS), "dropout": 0.0}).to(device).train()
    diff = GaussianMultinomialDiffusion(np.array([0]), n_num, net, num_timesteps=T, scheduler="cosine",
                                        device=torch.device(device)).to(device).train()
    data = torch.randn(8 * batch, n_num)
    y = torch.zeros(batch, dtype=torch.long)

    def it():
        i = 0
        while True:
            yield data[i * batch:(i + 1) * batch], y
            i = (i + 1) % 8

    tr = Trainer(diff, it(), lr=6e-4, weight_decay=1e-5, steps=10 ** 9, device=torch.device(device))
    src = it()
    for _ in range(warmup):
        x, yy = next(src)
        tr._run_step(x, {"y": yy})
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        x, yy = next(src)
        tr._run_step(x, {"y": yy})
    barrier()
    dt = time.perf_counter() - t0
    flops = batch * 6.0 * (640 * n_num + 4292608)
    tf = flops * steps / dt / 1e12
    # whole-step roofline: algorithmic FLOPs of the step (forward + both backward products of every nn.Linear) per second
    # against the dense f16 matrix peak; the split products issue three MFMAs per term, hence the /3 ceiling
    return {"steps_per_s": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3), "batch_per_gpu": batch,
            "steps_timed": steps, "tflops_per_gpu": round(tf, 1),
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_F16_MFMA_TFLOPS, 4), "frac_of_split_ceiling": round(3.0 * tf / PEAK_F16_MFMA_TFLOPS, 4),
                         "flops_per_step": flops, "scope": "whole training step (not one kernel): B*6*(640*D_in + 4292608) FLOP"}}


def classifier_train_bench(device, n_num, n_clusters, T, batch, steps, warmup, cpu_steps):
    """training steps/s of the guidance classifier (q-sample, train-mode forward with BatchNorm batch
    statistics and Dropout, cross-entropy, backward, AdamW; batch 4096, configs/*.json classifier shape), with the CPU
    oracle (torch CPU ops like the reference) timed on a few steps of the same shape."""
    import numpy as np
    from clavaddpm_amd.classifier_training import ClassifierTrainer
    from clavaddpm_amd.pipeline_utils import Classifier
    from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion
    torch.manual_seed(0)
    clf = Classifier(d_in=n_num, d_out=n_clusters, dim_t=128, hidden_sizes=list(CLASSIFIER_LAYERS)).to(device).train()
    diff = GaussianMultinomialDiffusion(np.array([0]), n_num, None, num_timesteps=T, scheduler="cosine",
                                        device=torch.device(device)).to(device)
    tr = ClassifierTrainer(clf, diff, lr=1e-4, data_parallel=False)      # this leg runs on rank 0 only
    x = torch.randn(8 * batch, n_num)
    y = torch.randint(0, n_clusters, (8 * batch,))
    for i in range(warmup):
        tr.step(x[(i % 8) * batch:(i % 8 + 1) * batch], y[(i % 8) * batch:(i % 8 + 1) * batch])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(x[(i % 8) * batch:(i % 8 + 1) * batch], y[(i % 8) * batch:(i % 8 + 1) * batch])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dims = [128] + list(CLASSIFIER_LAYERS)
    macs = 128 * n_num + 2 * 128 * 128 + sum(a * b for a, b in zip(dims[:-1], dims[1:])) + dims[-1] * n_clusters
    out = {"steps_per_s": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3), "batch_per_gpu": batch, "steps_timed": steps,
           "tflops_per_gpu": round(6.0 * batch * macs * steps / dt / 1e12, 1)}
    if cpu_steps > 0:
        from oracle import classifier_train as oc, schedule
        tb = schedule.tables("cosine", T)
        p = {k: v.detach().cpu().clone() for k, v in clf.state_dict().items()}
        g = torch.Generator().manual_seed(1)
        mk = lambda: (torch.randint(0, T, (batch,), generator=g), torch.randn(batch, n_num, generator=g),
                      (torch.rand(batch, sum(CLASSIFIER_LAYERS), generator=g) < 0.5).to(torch.uint8))
        draws = [mk() for _ in range(cpu_steps + 1)]
        run = lambda d: oc.run_loop(tb, p, [x[:batch]] * len(d), [y[:batch]] * len(d), [a[0] for a in d], [a[1] for a in d],
                                    [a[2] for a in d], 1e-4)
        run(draws[:1])
        t0 = time.perf_counter()
        run(draws[1:])
        dc = time.perf_counter() - t0
        out["cpu_baseline"] = {"steps_per_s": round(cpu_steps / dc, 3), "kind": "port", "cores": torch.get_num_threads(),
                               "sample": f"{cpu_steps} steps, {dc:.1f} s"}
    return out


def gmm_bench(device, n_rows, d, k, iters, cpu_rows):
    """Secondary measurement: the diagonal-covariance GMM E-step of parent-child clustering on a
    movie_lens-rating-sized child table (1 M rows), float64, X resident in HBM across EM iterations.  At k = 50 the kernel
    is bound by the f64 matrix pipe (4*N*d*k algorithmic FLOP; v_mfma_f64_16x16x4_f64 sustains 45 TFLOP/s on this chip,
    tools/micro/mfma_f64_rate.hip, against 78.6 in AMD's data sheet); algorithmic bytes are 8*N*(d + k + 1)."""
    import numpy as np
    from clavaddpm_amd.clustering import gmm_estep, gmm_mstep
    from oracle import gmm as ogmm
    rng = np.random.RandomState(0)
    means = rng.standard_normal((k, d))
    pc = rng.uniform(0.3, 3.0, (k, d))
    w = rng.dirichlet(np.ones(k))
    X = torch.randn(n_rows, d, device=device, dtype=torch.float64) * 2.0
    f = lambda a: torch.from_numpy(a).to(device)
    wd, md, pd_ = f(w), f(means), f(pc)
    for _ in range(2):
        gmm_estep(X, wd, md, pd_)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        lpn, resp, lab = gmm_estep(X, wd, md, pd_)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    # M-step and one whole EM iteration (E-step, M-step, lower bound; N-sized data never leaves the device)
    for _ in range(2):
        gmm_mstep(X, resp, lpn)
    c, e, f2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    c.record()
    for _ in range(iters):
        gmm_mstep(X, resp, lpn)
    e.record()
    for _ in range(iters):
        lpn, resp, lab = gmm_estep(X, wd, md, pd_)
        gmm_mstep(X, resp, lpn)
    f2.record()
    torch.cuda.synchronize()
    ms_m, ms_em = c.elapsed_time(e) / iters, e.elapsed_time(f2) / iters
    bytes_alg = 8.0 * n_rows * (d + k + 1)
    Xc = X[:cpu_rows].cpu().numpy()
    t0 = time.perf_counter()
    lpn_c, resp_c, lab_c = ogmm.estep(Xc, w, means, pc)
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(lab[:cpu_rows].cpu().numpy().astype(np.int64), lab_c))
    t0 = time.perf_counter()
    ogmm.mstep(Xc, resp_c)
    dtm = time.perf_counter() - t0
    # whole fit (device k-means++ seeding + EM, 10 iterations) through the scikit-learn-compatible class, host matrix in;
    # scikit-learn itself on a bounded sample of the same table as the CPU reference point
    import warnings
    from clavaddpm_amd.clustering import HipGaussianMixture
    Xh = X.cpu().numpy()
    kw = dict(n_components=k, covariance_type="diag", init_params="k-means++", tol=0.0, max_iter=10, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        HipGaussianMixture(**kw).fit(Xh[:50000])
        t0 = time.perf_counter()
        HipGaussianMixture(**kw).fit(Xh)
        dt_fit = time.perf_counter() - t0
        fit = {"rows": n_rows, "em_iterations": 10, "seconds": round(dt_fit, 3), "rows_per_s": round(n_rows / dt_fit, 0),
               "note": "HipGaussianMixture.fit(host matrix): upload + device k-means++ (k seeds) + 10 EM iterations"}
        try:
            from sklearn.mixture import GaussianMixture
            t0 = time.perf_counter()
            GaussianMixture(**kw).fit(Xh[:cpu_rows])
            dts = time.perf_counter() - t0
            fit["cpu_baseline"] = {"rows_per_s": round(cpu_rows / dts, 0), "kind": "reference (scikit-learn GaussianMixture.fit)",
                                   "sample": f"{cpu_rows} rows, {dts:.1f} s"}
        except ImportError:
            pass
    return {"rows": n_rows, "d": d, "k": k, "dtype": "f64", "ms_per_estep": round(ms, 3), "rows_per_s": round(n_rows / ms * 1e3, 0),
            "mstep": {"ms": round(ms_m, 3), "tflops_f64": round(4.0 * n_rows * d * k / ms_m / 1e9, 2),
                      "hbm_gbps": round(8.0 * n_rows * (d + k * ((d + 31) // 32) + 1) / ms_m / 1e6, 1),
                      "cpu_rows_per_s": round(cpu_rows / dtm, 0)},
            "fit": fit,
            "em_iteration": {"ms": round(ms_em, 3), "rows_per_s": round(n_rows / ms_em * 1e3, 0),
                             "cpu_rows_per_s": round(cpu_rows / (dt + dtm), 0)},
            "roofline": {"bound": "mfma", "achieved": round(4.0 * n_rows * d * k / ms / 1e9, 2), "peak": 78.6, "unit": "TFLOP/s",
                         "frac": round(4.0 * n_rows * d * k / ms / 1e9 / 78.6, 4), "measured_f64_mfma_peak": 45.4,
                         "frac_of_measured_peak": round(4.0 * n_rows * d * k / ms / 1e9 / 45.4, 4),
                         "hbm_gbps": round(bytes_alg / ms / 1e6, 1)},
            "cpu_baseline": {"rows_per_s": round(cpu_rows / dt, 0), "kind": "port (numpy restatement of scikit-learn's E-step)",
                             "sample": f"{cpu_rows} rows, {dt:.2f} s", "labels_match_gpu": same}}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N rank processes ourselves, as fresh children of a
    parent that has made no GPU call (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), pass rank 0's JSON
    line through and return the launcher's exit status (non-zero when any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this pool (RCCL needs it)
    env["CLAVA_BENCH_SPAWNED"] = "1"
    return subprocess.call(cmd, env=env)


def trace(msg):
    """Progress markers on stderr (CLAVA_BENCH_TRACE=1): where a multi-rank run stands."""
    if os.environ.get("CLAVA_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=20000, help="rows per chain per GPU (sampling.batch_size)")
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--n-num", type=int, default=2)
    ap.add_argument("--clusters", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--train-steps", type=int, default=400, help="timed training steps for the secondary metric (0 = skip)")
    ap.add_argument("--no-gmm", action="store_true", help="skip the GMM E-step leg")
    ap.add_argument("--clf-train-steps", type=int, default=300, help="timed classifier-training steps (0 = skip)")
    ap.add_argument("--fit-steps", type=int, default=600, help="Trainer steps that fit the denoiser to the synthetic table before sampling")
    ap.add_argument("--gemm", choices=["f16x3", "f32"], default="f16x3",
                    help="layer kernels of the sampling loop: split-f16 on the f16 matrix cores (float32-level accuracy) or float32-input MFMA")
    ap.add_argument("--no-scale-sweep", action="store_true", help="skip the step-kernel / scale-sweep (100 num + 50 cat columns) legs")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra chain timed on the float32-input MFMA kernels")
    ap.add_argument("--no-t2000", action="store_true", help="skip the extra chain at T = 2000 (the configs' num_timesteps)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes the launcher (it has not touched the GPU) and exits with the ranks' status
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CLAVA_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
    # devices, collectives over gloo); measurements always use RCCL ("nccl"), one rank per GPU.
    backend = os.environ.get("CLAVA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist = None
    joined = 1
    if dist is not None:
        # ranks that actually joined the group: what `n_gpus` reports
        one = torch.ones(1, device=f"cuda:{local_rank}") if backend == "nccl" else torch.ones(1)
        dist.all_reduce(one)
        joined = int(one.item())
        if joined != args.gpus:
            raise SystemExit(f"bench.py: {joined} ranks joined the process group, --gpus {args.gpus} asked for")
    if os.environ.get("CLAVA_BENCH_SPAWN_CHECK"):
        # launcher rehearsal (tests/test_host.py, CPU, gloo): the group is up; report it and leave before any GPU work
        if rank == 0:
            print(json.dumps({"spawn_check": True, "n_gpus": joined, "backend": backend,
                              "spawned_by_bench": bool(os.environ.get("CLAVA_BENCH_SPAWNED"))}))
        if os.environ.get("CLAVA_BENCH_SPAWN_CHECK") == "fail":
            raise SystemExit(3 if rank == world - 1 else 0)    # a failing rank must fail the whole run
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    from clavaddpm_amd import _lib, split_nets
    L = _lib.lib()
    T, B = args.timesteps, args.batch
    split_nets.set_gemm_mode(args.gemm)
    if os.environ.get("CLAVA_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("CLAVA_BENCH_TRACE_AFTER", "60")), exit=True)   # where a hung rank stands
    trace("process group up; building the workload")
    diff, net, clf, cond_fn = build_workload(device, args.n_num, args.clusters, T, args.fit_steps)
    trace("workload built")
    g = torch.Generator().manual_seed(1234 + rank)
    ys = torch.randint(0, args.clusters, (B, 1), generator=g).to(device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    seed = 20240
    for w in range(args.warmup):
        x, keep = run_chain(diff, cond_fn, ys, seed, rank * B)
        torch.cuda.synchronize()

    n_fam = L.clava_prof_family_count()
    L.clava_prof_family_name.restype = ctypes.c_char_p

    trace("warm-up done")
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        x, keep = run_chain(diff, cond_fn, ys, seed + 1 + k, (world * k + rank) * B)
    barrier()
    elapsed = time.perf_counter() - t0
    trace("timed region done")
    if dist is not None:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(x).all()), "non-finite samples"
    # algorithmic FLOPs per synthetic row: T * (denoiser forward + classifier forward + input gradient)
    dn, cn = [args.n_num, 128] + DENOISER_LAYERS + [args.n_num], [args.n_num, 128] + CLASSIFIER_LAYERS + [args.clusters]
    mac_den = sum(a * b for a, b in zip(dn[:-1], dn[1:]))
    mac_cls = sum(a * b for a, b in zip(cn[:-1], cn[1:]))
    flop_per_row = T * (2.0 * mac_den + 4.0 * mac_cls)
    value_tflops = world * B * args.steps * flop_per_row / elapsed / 1e12
    x_absmax = float(x.abs().max())

    # roofline sample: one more chain of the same workload, replayed on ONE stream, with the library bracketing each of
    # its first launches by two HIP events on the launch stream (clava_prof_*).  In the timed region above the denoiser
    # and the guidance kernels of a step overlap on two streams, so per-launch intervals there are not exclusive.
    roofline = None
    families = {}
    prof_records = 2048 if rank == 0 else 0
    if prof_records:
        os.environ["CLAVA_CHAIN_STREAMS"] = "1"
        _lib.check(L.clava_prof_enable(ctypes.c_int(prof_records)), "clava_prof_enable")
        xr, keepr = run_chain(diff, cond_fn, ys, seed + 1, rank * B)
        torch.cuda.synchronize()
        del os.environ["CLAVA_CHAIN_STREAMS"]
        L.clava_prof_disable()
        launches = (ctypes.c_int * n_fam)()
        ms = (ctypes.c_double * n_fam)()
        fl = (ctypes.c_double * n_fam)()
        by = (ctypes.c_double * n_fam)()
        L.clava_prof_collect(launches, ms, fl, by)
        for f in range(n_fam):
            if launches[f]:
                families[L.clava_prof_family_name(f).decode()] = {
                    "launches": launches[f], "avg_us": 1e3 * ms[f] / launches[f], "ms": ms[f],
                    "tflops": fl[f] / ms[f] / 1e9 if ms[f] > 0 else 0.0, "gbps": by[f] / ms[f] / 1e6 if ms[f] > 0 else 0.0}
        names = [L.clava_prof_family_name(f).decode() for f in range(n_fam)]
        dom = "gemm_f16x3_256x320" if args.gemm == "f16x3" else "gemm_f32_nt_128x128"
        if dom in families:
            traffic = None
            for rnd in ("r03", "r02", "r01"):                       # newest committed PMC measurement of this kernel
                tpath = os.path.join(ROOT, "profiles", f"traffic_{rnd}_f16x3.json" if args.gemm == "f16x3" else f"traffic_{rnd}.json")
                if os.path.exists(tpath):
                    traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                    break
            f = families[dom]
            fi = names.index(dom)
            if args.gemm == "f16x3":
                # algorithmic FLOPs = 2*M*N*K of the float32 product; the kernel issues 3 f16 MFMAs per product term, so
                # the matrix cores are busy for 3x that: both fractions are reported, `frac` is the algorithmic one
                allf = [v for kk, v in families.items() if kk.startswith("gemm_f16x3")]
                all_tf = sum(v["tflops"] * v["ms"] for v in allf) / max(sum(v["ms"] for v in allf), 1e-9)
                roofline = {"bound": "mfma", "measured_on": "serial replay of one chain after the timed region (one stream)",
                            "chain_tflops_timed_region": round(value_tflops, 2),
                            "kernel": "gemm_f16x3_kernel<256,320,2> (split-f16: 3 x v_mfma_f32_16x16x32_f16 per float32 product; "
                                                       "the 1024-wide layers of the denoiser, the classifier and its input gradient)",
                            "all_f16x3_layers_tflops": round(all_tf, 2),
                            "achieved": round(f["tflops"], 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(f["tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                            "frac_of_split_ceiling": round(3.0 * f["tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                            "mfma_issue_frac": round(3.0 * f["tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                            "vs_f32_matrix_peak": round(f["tflops"] / PEAK_F32_MFMA_TFLOPS, 3), "traffic": traffic,
                            "avg_launch_us": round(f["avg_us"], 2), "launches_timed": f["launches"],
                            "flops_per_launch": round(fl[fi] / launches[fi])}
            else:
                roofline = {"bound": "mfma", "kernel": "gemm_f32_kernel<true,true,128,128,true,false> (fp32 v_mfma_f32_32x32x2_f32; the denoiser's layers)",
                            "achieved": round(f["tflops"], 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(f["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                            "avg_launch_us": round(f["avg_us"], 2), "launches_timed": f["launches"],
                            "flops_per_launch": round(fl[fi] / launches[fi])}

    trace("roofline replay done")
    train = None
    if args.train_steps > 0:
        train = train_bench(device, args.n_num, T, 4096, args.train_steps, 10, barrier)
        if dist is not None:
            tm = torch.tensor([train["ms_per_step"]], device=device, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            train["ms_per_step"] = round(float(tm.item()), 3)
            train["steps_per_s"] = round(1e3 / float(tm.item()), 2)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            share = cpu_share()
            v, dtc = train_cpu_baseline(args.n_num, T, 4096, 10, share)
            v1, dt1 = train_cpu_baseline(args.n_num, T, 4096, 3, 1)
            train["cpu_baseline"] = {"value": round(v, 3), "unit": "steps/s", "cores": share, "kind": "port",
                                     "sample": f"10 oracle training steps at B 4096 ({dtc:.1f} s)",
                                     "one_thread": {"value": round(v1, 3), "cores": 1, "sample": f"3 steps ({dt1:.1f} s)"},
                                     "cores_visible": os.cpu_count()}

    trace("training leg done")
    # the same chain on the float32-input MFMA kernels (one chain, after the timed region): the exact-f32 reference point
    f32_leg = None
    if args.gemm == "f16x3" and not args.no_f32_leg:
        split_nets.set_gemm_mode("f32")
        x32, keep32 = run_chain(diff, cond_fn, ys, seed + 1 + (args.steps - 1), (world * (args.steps - 1) + rank) * B)
        barrier()
        t1 = time.perf_counter()
        x32b, keep32 = run_chain(diff, cond_fn, ys, seed + 1 + (args.steps - 1), (world * (args.steps - 1) + rank) * B)
        barrier()
        dt32 = time.perf_counter() - t1
        split_nets.set_gemm_mode(args.gemm)
        dev_rel = float((x - x32).abs().max() / x32.abs().max())
        f32_leg = {"rows_per_s_per_gpu": round(B / dt32, 1), "ms_per_chain": round(1e3 * dt32, 1),
                   "max_rel_dev_of_f16x3_samples": dev_rel}

    # the same chain through the public drop-in API (`conditional_sample`: label upload, chain, float64 result copied to
    # the host as the reference returns it) - the PCIe-inclusive rate; never `value`
    api = None
    if rank == 0:
        ys_cpu = ys.cpu()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        xs, _ = diff.conditional_sample(ys=ys_cpu, model_kwargs={"y": ys_cpu}, cond_fn=cond_fn, seed=seed + 99, row_offset=0)
        dt_api = time.perf_counter() - t1
        api = {"rows_per_s_per_gpu": round(B / dt_api, 1), "ms_per_chain": round(1e3 * dt_api, 1),
               "note": "diffusion.conditional_sample(...) incl. host<->device copies of labels and samples"}

    # the same workload at the configs' own chain length (configs/movie_lens.json:14-45 num_timesteps = 2000):
    # a second denoiser fitted under the T = 2000 schedule, one warm-up quarter chain, one timed chain
    t2000 = None
    if rank == 0 and world == 1 and not args.no_t2000 and T != 2000:
        diff2, net2, clf2, cond2 = build_workload(device, args.n_num, args.clusters, 2000, args.fit_steps)
        x2, idx2 = diff2._initial_state(B, None, seed, 0)
        diff2._run_chain(x2, idx2, cond2, ys, None, seed, 0, n_steps=500)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        x2, keep2 = run_chain(diff2, cond2, ys, seed + 7, 0)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        assert bool(torch.isfinite(x2).all()), "non-finite samples at T = 2000"
        t2000 = {"timesteps": 2000, "rows_per_s_per_gpu": round(B / dt2, 1), "ms_per_chain": round(1e3 * dt2, 1), "chains_timed": 1,
                 "sample_absmax": round(float(x2.abs().max()), 3),
                 "note": "what configs/*.json run (num_timesteps 2000); time is linear in T"}
        del diff2, net2, clf2, cond2, x2, keep2

    gmm = None
    if rank == 0 and world == 1 and not args.no_gmm:              # single-GPU side leg: reported at N = 1 only
        gmm = gmm_bench(device, 1_000_000, 32, 50, 20, 200_000 if not args.no_cpu_baseline else 20_000)

    clf_train = None
    if rank == 0 and world == 1 and args.clf_train_steps > 0:     # single-GPU side leg: reported at N = 1 only
        clf_train = classifier_train_bench(device, args.n_num, args.clusters, T, 4096, args.clf_train_steps, 10,
                                           0 if args.no_cpu_baseline else 3)

    # single-GPU side legs at the scale-sweep shape (BASELINE.json configs[4]): the fused step kernels against the HBM
    # roofline (north_star: "achieved HBM GB/s for the step kernel"), and the training step / reverse step of that table
    step_kernels = sweep = None
    if rank == 0 and world == 1 and not args.no_scale_sweep:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_scale_sweep
        import bench_step
        step_kernels = bench_step.measure(B=20000, K=bench_scale_sweep.classes())
        step_kernels["peak_gbps"] = 8000.0
        step_kernels["note"] = ("clava_gauss_step_f32 / clava_multinomial_step_f32 (reverse mode, in-kernel Philox noise, next denoiser "
                                "input written) at 100 numeric + 50 categorical columns; bytes = DESIGN.md section 4; the categorical "
                                "kernel is bound by its 5 transcendentals per class, not by HBM")
        split_nets.set_gemm_mode("f16x3")
        sweep = bench_scale_sweep.measure(batch=4096, train_steps=60, breakdown=True)
        # how often the float32-level pipelines resolve a Gumbel arg-max differently from exact arithmetic (tools/flip_rate.py:
        # full batch, 4 teacher-forced steps, injected uniforms, float64 evaluation of the whole step in torch on the GPU)
        import flip_rate
        sweep["category_flips_vs_float64"] = flip_rate.measure(batch=4096, steps=4, f64_device=device)
        split_nets.set_gemm_mode(args.gemm)

    rows = world * B * args.steps
    value = rows / elapsed
    out = {
        "metric": f"synth rows/sec ({T}-step reverse diffusion)", "value": round(value, 1), "unit": "rows/s",
        "n_gpus": joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (split-f16x3 MFMA operands, f32 accumulate, 22-bit significand)" if args.gemm == "f16x3" else "f32",
        "data": f"synthetic (N(0,1) table; denoiser fitted {args.fit_steps} Trainer steps to it, classifier random-init)",
        "config": {"workload": "movie_lens (movie->rating) cluster-guided sampling, configs/movie_lens.json shapes",
                   "rows_per_gpu_per_step": B, "timesteps": T, "n_num": args.n_num, "clusters": args.clusters,
                   "denoiser": DENOISER_LAYERS, "classifier": CLASSIFIER_LAYERS, "guidance_scale": 1.0,
                   "parallelism": f"rows sharded over {world} GPU(s), no data-path collective"},
        "roofline": roofline,
        "train": train,
        "f32_mfma_path": f32_leg,
        "api_path_incl_pcie": api,
        "t2000": t2000,
        "gmm_estep": gmm,
        "classifier_train": clf_train,
        "step_kernels": step_kernels,
        "scale_sweep": sweep,
        "sample_absmax": round(x_absmax, 3),
        "f16x3_fallbacks": split_nets.fallback_count,
    }
    if families:
        out["kernel_families"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                                      "tflops": round(v["tflops"], 2), "gbps": round(v["gbps"], 1)} for k, v in families.items()}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = cpu_share()
        cb = min(B, 20000)
        v, dt = cpu_baseline(args.n_num, args.clusters, T, cb, args.cpu_steps, threads)
        n1 = max(2, args.cpu_steps // 10)
        v1, dt1 = cpu_baseline(args.n_num, args.clusters, T, cb, n1, 1)
        torch.set_num_threads(threads)
        out["cpu_baseline"] = {"value": round(v, 2), "unit": "rows/s", "cores": threads, "kind": "port",
                               "sample": f"{args.cpu_steps} guided reverse steps (t={T - 1}..{T - args.cpu_steps}) on {cb} rows, "
                                         f"same network shapes (seeded weights); {dt:.1f} s of CPU work, extrapolated to {T} steps",
                               "one_thread": {"value": round(v1, 2), "cores": 1, "sample": f"{n1} steps, {dt1:.1f} s"},
                               "cores_visible": os.cpu_count(),
                               "note": "cores = this process's CPU share (sched affinity, capped by the box's OMP_NUM_THREADS); "
                                       "os.cpu_count() on the GPU box counts the whole multi-GPU machine"}
        out["speedup_vs_cpu"] = round(value / v, 1)
    if rank == 0:
        print(json.dumps(out))
    trace("done")
    if os.environ.get("CLAVA_BENCH_TRACE"):
        faulthandler.cancel_dump_traceback_later()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
