"""CPU-only checks: C-ABI surface, host-side logic of the drop-in classes, sharding helpers (gloo, world_size 2)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import ROOT, golden

from clavaddpm_amd import _lib, parallel
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
from clavaddpm_amd.tab_ddpm.schedule import build_tables


def test_cabi_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert L.clava_version() == 110
    assert L.clava_device_count() >= 0            # host-only probe; no compute call without a GPU


def test_header_cites_reference_for_every_entry_point():
    text = open(os.path.join(ROOT, "include", "clava_hip.h")).read()
    assert "tab_ddpm/gaussian_multinomial_diffsuion.py" in text and "pipeline_utils.py" in text and "scripts/train.py" in text
    assert 'extern "C"' in text and "at::Tensor" not in text and "torch::" not in text and "#include <torch" not in text


def test_product_schedule_tables_match_reference_golden():
    g = golden("g1_schedule.npz")
    for sched in ("cosine", "linear"):
        for T in (50, 1000, 2000):
            f32, f64 = build_tables(sched, T)
            for k, v in f32.items():
                ref = g[f"{sched}_{T}_{k}"].numpy()
                assert v.dtype == np.float32
                np.testing.assert_allclose(v, ref, rtol=1.2e-7, atol=1e-38, err_msg=f"{sched} {T} {k}")
            np.testing.assert_allclose(f64["posterior_variance"], g[f"{sched}_{T}_posterior_variance_f64"].numpy(), rtol=1e-14)
            np.testing.assert_allclose(f64["model_variance"], g[f"{sched}_{T}_model_variance_f64"].numpy(), rtol=1e-14)
            np.testing.assert_allclose(np.log(f64["model_sigma"]) * 2, g[f"{sched}_{T}_model_log_variance_f64"].numpy(), rtol=1e-12)


def _small(K=(3, 5, 2), n_num=4):
    d_in = n_num + (sum(K) if K[0] else 0)
    net = MLPDiffusion(d_in, 0, "none", {"d_layers": [64, 64], "dropout": 0.0})
    return net, GaussianMultinomialDiffusion(np.array(K), n_num, net, num_timesteps=50)


def test_constructor_attributes_and_state_dict_layout():
    from weights import mlp_shapes
    net, d = _small()
    assert list(net.state_dict().keys()) == list(mlp_shapes(14, [64, 64]).keys())
    for k, shape in mlp_shapes(14, [64, 64]).items():
        assert tuple(net.state_dict()[k].shape) == tuple(shape)
    ref_buffers = ["alphas", "log_alpha", "log_1_min_alpha", "log_1_min_cumprod_alpha", "log_cumprod_alpha", "alphas_cumprod",
                   "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                   "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "Lt_history", "Lt_count"]
    sd = [k for k in d.state_dict().keys() if not k.startswith("_denoise_fn.")]
    assert sd == ref_buffers                                  # the HIP-only tables are non-persistent
    assert d.num_timesteps == 50 and d._denoise_fn is net
    assert d.offsets.tolist() == [0, 3, 8, 10] and d.num_classes_expanded.tolist() == [3] * 3 + [5] * 5 + [2] * 2
    assert d.posterior_variance.dtype == torch.float64 and not d.posterior_variance.is_cuda
    rt = {"d_layers": [8], "dropout": 0.0}
    MLPDiffusion(3, 0, "none", rt)
    assert rt["d_in"] == 128 and rt["d_out"] == 3            # the reference mutates the caller's dict (modules.py:434-435)
    _, d0 = _small(K=(0,))
    assert not d0.has_cat and d0._m == 0 and d0._sum_k == 0


def test_diffusion_pickles_without_library_handles():
    net, d = _small()
    d2 = pickle.loads(pickle.dumps(d))
    assert torch.equal(d2.log_alpha, d.log_alpha) and torch.equal(d2._model_sigma, d._model_sigma)
    for a, b in zip(d.parameters(), d2.parameters()):
        assert torch.equal(a, b)


def test_no_cpu_fallback_the_product_path_fails_loudly():
    net, d = _small(K=(0,), n_num=4)
    with pytest.raises(_lib.ClavaHipError):
        net(torch.zeros(2, 4), torch.zeros(2, dtype=torch.long))
    with pytest.raises(_lib.ClavaHipError):
        d.sample(2, torch.ones(1))
    with pytest.raises(_lib.ClavaHipError):
        d.mixed_loss(torch.zeros(2, 4), {"y": torch.zeros(2, dtype=torch.long)})
    with pytest.raises(_lib.ClavaHipError):
        d.gaussian_q_sample(torch.zeros(2, 4), torch.zeros(2, dtype=torch.long))


def test_product_never_imports_the_oracle():
    import re
    pkg = os.path.join(ROOT, "clavaddpm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)


def test_row_slices_and_batches_partition_the_rows():
    for n in (0, 1, 7, 20000, 10_000_019):
        for world in (1, 2, 3, 8):
            cuts = [parallel.row_slice(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    assert parallel.plan_batches(5, 5, 10) == []
    assert parallel.plan_batches(3, 26, 10) == [(3, 13), (13, 23), (23, 26)]       # short last batch, like the reference loop


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 11
        a, b = parallel.row_slice(n, rank, world)
        local = torch.arange(a, b, dtype=torch.float64).reshape(-1, 1) * torch.ones(1, 3, dtype=torch.float64)
        full = parallel.gather_rows(local)
        empty = parallel.gather_rows(torch.zeros((0 if rank else 2, 3), dtype=torch.float64))
        g = torch.full((5,), float(rank + 1))
        parallel.allreduce_mean_(g)
        q.put((rank, full[:, 0].tolist(), tuple(empty.shape), g.tolist()))
    finally:
        dist.destroy_process_group()


def test_gather_and_gradient_average_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rows, eshape, g in out:
        assert rows == [float(i) for i in range(11)]           # rank order, ragged slices (6 + 5)
        assert eshape == (2, 3)                                # a rank with zero rows still takes part
        assert g == [1.5] * 5                                  # mean of (1, 2)


BERKA = [(None, "district"), ("district", "account"), ("district", "client"), ("account", "loan"), ("account", "order"),
         ("account", "trans"), ("account", "disp"), ("disp", "card")]
BERKA_ROWS = [77, 4500, 5369, 682, 6471, 1056320, 5369, 892]


def _gloo_worker_mstep_relations(rank, world, port, q, tmpdir):
    """Rank body: (1) the row-sharded M-step - every rank reduces its own rows with the oracle, the library-layout totals
    and row counts are summed by parallel.allreduce_mstep_totals_, parameters are finalised from the sums; (2) the
    relation-parallel training driver with stub trainers, with and without a checkpoint directory."""
    import numpy as np
    import torch.distributed as dist
    from clavaddpm_amd import clustering
    from oracle import gmm as ogmm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(5)
        N, d, k = 1001, 5, 7
        X = rng.standard_normal((N, d)) * 2.0
        lr_ = rng.standard_normal((N, k))
        lr_ -= np.log(np.exp(lr_).sum(1, keepdims=True))
        lpn = rng.standard_normal(N)
        a, b = parallel.row_slice(N, rank, world)
        nk, sx, sxx, lp = ogmm.mstep_totals(X[a:b], lr_[a:b], lpn[a:b])
        tot = clustering.pack_mstep_totals(sx, sxx, nk, lp)
        tot, n_all = parallel.allreduce_mstep_totals_(tot, b - a)
        sx2, sxx2, nk2, lp2 = clustering.unpack_mstep_totals(tot, k, d)
        got = ogmm.mstep_from_totals(nk2.numpy(), sx2.numpy(), sxx2.numpy())
        want = ogmm.mstep(X, lr_)
        err = max(float(np.abs(g - w_).max()) for g, w_ in zip(got, want))
        mstep_ok = (n_all == N, err < 1e-12, abs(float(lp2) / n_all - lpn.mean()) < 1e-12)

        cost = dict(zip(BERKA, BERKA_ROWS))
        seen = []

        def stub_train(parent, child):
            seen.append((parent, child))
            assert not parallel.data_parallel_active()      # trainers built here must not all-reduce
            return {"relation": (parent, child), "trained_by": rank}

        assert parallel.data_parallel_active()
        models = parallel.train_relations(BERKA, stub_train, save_dir=None, cost=cost)
        in_order = list(models.keys()) == BERKA
        by = {rel: m["trained_by"] for rel, m in models.items()}
        plan = parallel.assign_relations(BERKA, world, cost)
        planned = all(by[rel] == r for r, rels in plan.items() for rel in rels) and seen == plan[rank]
        # with a checkpoint directory: results travel as files; a second pass finds every checkpoint and trains nothing
        models_f = parallel.train_relations(BERKA, stub_train, save_dir=tmpdir, cost=cost)
        n_seen = len(seen)
        models_f2 = parallel.train_relations(BERKA, stub_train, save_dir=tmpdir, cost=cost)
        files_ok = (models_f == models and models_f2 == models and len(seen) == n_seen
                    and sorted(os.listdir(os.path.join(tmpdir, "models"))) == sorted(f"{p}_{c}_ckpt.pkl" for p, c in BERKA))
        q.put((rank, mstep_ok, in_order, planned, files_ok))
    finally:
        dist.destroy_process_group()


def test_mstep_allreduce_and_relation_training_world_size_2_gloo(tmp_path):
    """GMM M-step sufficient statistics all-reduce, and berka's per-relation training (complex_pipeline.py:59-84)
    spread over ranks - the reduction / assignment / exchange logic on two gloo ranks (the device halves are covered by
    the -m gpu tests)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker_mstep_relations, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mstep_ok, in_order, planned, files_ok in out:
        assert all(mstep_ok), (rank, mstep_ok)
        assert in_order and planned and files_ok, (rank, in_order, planned, files_ok)


def test_relations_are_assigned_to_ranks_deterministically():
    """configs/berka.json: 8 tables, per-relation denoisers trained on separate GPUs (no collective)."""
    rel = [(None, "district"), ("district", "account"), ("district", "client"), ("account", "loan"), ("account", "order"),
           ("account", "trans"), ("account", "disp"), ("disp", "card")]
    cost = {r: c for r, c in zip(rel, [77, 4500, 5369, 682, 6471, 1056320, 5369, 892])}
    plan = parallel.assign_relations(rel, 8, cost)
    key = lambda r: (str(r[0]), r[1])
    assert sorted(sum(plan.values(), []), key=key) == sorted(rel, key=key) and all(len(v) == 1 for v in plan.values())
    plan3 = parallel.assign_relations(rel, 3, cost)
    assert sorted(sum(plan3.values(), []), key=key) == sorted(rel, key=key)
    assert plan3[0] == [("account", "trans")]                      # the 1 M-row table gets a GPU to itself
    assert plan3 == parallel.assign_relations(list(rel), 3, dict(cost))
    one = parallel.assign_relations(rel, 1)
    assert one == {0: rel}                                         # original order kept inside a rank


def test_classifier_training_host_logic():
    """Host pieces of the classifier-training path that need no GPU: the uniform timestep sampler reproduces the
    reference's NumPy draw, the flat layout leaves the never-called transformer out, and the new product modules fail
    loudly without a GPU instead of computing on the CPU."""
    import numpy as np
    from clavaddpm_amd import _lib, training
    from clavaddpm_amd.classifier_training import ClassifierTrainer
    from clavaddpm_amd.pipeline_utils import Classifier
    from clavaddpm_amd.tab_ddpm.resample import create_named_schedule_sampler

    class D:
        num_timesteps = 1000
    s = create_named_schedule_sampler("uniform", D())
    np.random.seed(5)
    t, w = s.sample(16, "cpu")
    np.random.seed(5)
    ref = np.random.choice(1000, size=(16,), p=np.ones(1000) / 1000)
    assert t.dtype == torch.int64 and np.array_equal(t.numpy(), ref) and bool((w == 1).all())
    with pytest.raises(NotImplementedError):
        create_named_schedule_sampler("loss-second-moment", D())

    clf = Classifier(d_in=3, d_out=4, dim_t=128, hidden_sizes=[16, 8], keep_unused_transformer=True)
    lay = training.FlatLayout(clf, skip_prefixes=("transformer_layer",))
    assert not any(n.startswith("transformer_layer") for n in lay.names)
    assert set(lay.names) == {n for n, _ in clf.named_parameters() if not n.startswith("transformer_layer")}
    assert all(o % 4 == 0 for o in lay.offsets)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.ClavaHipError):
            ClassifierTrainer(clf, D())
        with pytest.raises(_lib.ClavaHipError):          # CPU tensors are refused before anything else
            clf.train()(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long))


def test_clustering_and_postprocess_host_logic():
    import numpy as np
    from clavaddpm_amd import _lib, clustering, postprocess
    k, d = 3, 2
    buf = np.arange(2 + k + 3 * k * d, dtype=np.float64)
    st, w, m, cv, pc = clustering.unpack_mstep(buf, k, d)
    assert st.tolist() == [0, 1] and w.tolist() == [2, 3, 4] and m.shape == (k, d) and m[0, 0] == 5 and pc[-1, -1] == buf[-1]
    real = np.array([[1.0, 0.5, 3.0], [2.0, 0.25, 3.0], [2.0, 7.5, 40.0]])
    assert postprocess.discrete_columns(real) == [0, 2]
    if not torch.cuda.is_available():
        with pytest.raises(_lib.ClavaHipError):
            clustering.kmeans_plusplus(torch.zeros((4, 2), dtype=torch.float64), 2, np.random.RandomState(0))
        with pytest.raises(_lib.ClavaHipError):
            postprocess.quantile_inverse(torch.zeros((4, 2), dtype=torch.float64), torch.zeros((10, 2), dtype=torch.float64),
                                         torch.zeros(10, dtype=torch.float64))


def test_bench_gpus_n_starts_n_ranks_itself():
    """VERDICT r2 missing 1: `python bench.py --gpus 2` with no launcher around it must start 2 ranks (fresh children of a
    parent that made no GPU call), report the ranks that joined the group as n_gpus, fail when a rank fails, and refuse a
    launcher whose WORLD_SIZE disagrees with --gpus.  Rehearsed on the CPU with gloo; the group is up, then the ranks leave
    before any GPU work (CLAVA_BENCH_SPAWN_CHECK)."""
    import json, subprocess, sys
    env = dict(os.environ, CLAVA_BENCH_BACKEND="gloo", CLAVA_BENCH_SPAWN_CHECK="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout                      # rank 0 alone prints
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["spawned_by_bench"] is True
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, CLAVA_BENCH_SPAWN_CHECK="fail"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0                             # one failing rank fails the run
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
