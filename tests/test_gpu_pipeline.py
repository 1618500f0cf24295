"""One pipeline-shaped run: the stages of `complex_pipeline.py:59-214` for one (parent, child) relation,
in the reference's order, on synthetic tables - clustering (GMM fit/predict + majority vote per parent, `pipeline_modules.py:
205-344`), denoiser training (`child_training` -> `train_model` -> `Trainer.run_loop`), classifier training
(`train_classifier`), the result dict pickled and re-loaded like `models/{parent}_{child}_ckpt.pkl`
(`complex_pipeline.py:62-78`), cluster-guided sampling by group size (`pipeline_utils.py:866-898`) and the decoding of
the samples (`:912-934`).  Each stage is parity-tested on its own elsewhere; this test checks the PLUMBING between them:
shapes, finiteness, label bookkeeping, and that the unpickled objects sample bit-identically to the live ones."""
import io
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clavaddpm_amd import parallel, postprocess
from clavaddpm_amd.classifier_training import ClassifierTrainer
from clavaddpm_amd.clustering import HipGaussianMixture, group_vote
from clavaddpm_amd.pipeline_utils import Classifier, make_cond_fn
from clavaddpm_amd.scripts.train import Trainer
from clavaddpm_amd.scripts.utils_train import get_model
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion

DEV = "cuda"


def test_one_relation_clustering_training_checkpoint_sampling_decoding():
    from sklearn.preprocessing import QuantileTransformer
    rng = np.random.RandomState(0)
    torch.manual_seed(0)
    # ---- synthetic parent / child tables: 400 parents with 1..6 children each; child rows have 3 numeric columns, one of
    # them integer-valued (rounded after sampling), drawn around a per-parent centre so that clusters exist
    n_parent, k_clusters, T = 400, 8, 50
    lengths = rng.randint(1, 7, n_parent)
    centre = rng.standard_normal((n_parent, 3)) * 2.0
    child = np.repeat(centre, lengths, axis=0) + 0.3 * rng.standard_normal((int(lengths.sum()), 3))
    child[:, 2] = np.clip(np.round(child[:, 2] * 2), -6, 6)                   # 13 integer levels
    parent = centre + 0.1 * rng.standard_normal((n_parent, 3))
    N = child.shape[0]

    # ---- clava_clustering: features = child columns | parent columns (scaled), GMM, majority vote per parent group
    feats = np.concatenate([child, 2.0 * np.repeat(parent, lengths, axis=0)], axis=1)
    feats = (feats - feats.min(0)) / (feats.max(0) - feats.min(0))
    gmm = HipGaussianMixture(n_components=k_clusters, covariance_type="diag", init_params="k-means++", tol=1e-4, random_state=0)
    row_labels = gmm.fit(feats).predict(feats)
    assert row_labels.shape == (N,) and 0 <= row_labels.min() and row_labels.max() < k_clusters
    g_lab, agree, assign, len_prob = group_vote(row_labels, lengths)
    assert g_lab.shape == (n_parent,) and assign.shape == (N, 1)
    assert np.array_equal(assign.reshape(-1), np.repeat(g_lab, lengths))       # every child row carries its parent's cluster
    assert np.all((agree > 0) & (agree <= 1))
    for lab, dist in len_prob.items():
        assert abs(sum(dist.values()) - 1.0) < 1e-12 and set(dist) <= set(lengths[g_lab == lab].tolist())
    C = int(g_lab.max()) + 1

    # ---- child_training: quantile-normalised numeric table -> denoiser (train_model) and classifier (train_classifier)
    qt = QuantileTransformer(output_distribution="normal", n_quantiles=min(N // 30, 1000), subsample=int(1e9), random_state=0)
    x_norm = torch.from_numpy(qt.fit_transform(child)).float()
    y_rows = torch.from_numpy(assign.reshape(-1)).long()
    B = 256

    def loader(seed):
        g = torch.Generator().manual_seed(seed)
        while True:                                                           # FastTensorDataLoader: reshuffle, short last batch
            perm = torch.randperm(N, generator=g)
            for i in range(0, N, B):
                j = perm[i:i + B]
                yield x_norm[j], y_rows[j]

    model_params = {"d_in": 3, "num_classes": 0, "is_y_cond": "none", "rtdl_params": {"d_layers": [128, 256, 128], "dropout": 0.0}}
    net = get_model("mlp", model_params).to(DEV)
    diffusion = GaussianMultinomialDiffusion(np.array([0]), 3, net, num_timesteps=T, scheduler="cosine", device=torch.device(DEV)).to(DEV)
    diffusion.train()
    tr = Trainer(diffusion, loader(1), lr=1e-3, weight_decay=0.0, steps=50, device=torch.device(DEV))
    tr.log_every = 10
    tr.run_loop()
    hist = tr.loss_history
    assert len(hist) >= 4 and np.isfinite(hist["gloss"]).all() and hist["gloss"].iloc[-1] < hist["gloss"].iloc[0]

    clf = Classifier(d_in=3, d_out=C, dim_t=128, hidden_sizes=[128, 256, 128]).to(DEV)
    sched_only = GaussianMultinomialDiffusion(np.array([0]), 3, None, num_timesteps=T, scheduler="cosine", device=torch.device(DEV)).to(DEV)
    ct = ClassifierTrainer(clf, sched_only, lr=1e-3)
    ct.run_loop(loader(2), 50, val_iter=loader(3))
    ct.finalize()
    clf.eval()
    assert len(ct.history) == 5 and all(np.isfinite(r[1]) for r in ct.history)
    assert ct.history[-1][1] < ct.history[0][1]                               # cross-entropy went down

    # ---- the relation's checkpoint, as complex_pipeline.py:62-78 writes and reads it
    diffusion.eval(); net.eval()
    result = {"diffusion": diffusion, "classifier": clf, "num_transform": qt, "group_lengths_prob_dicts": len_prob,
              "label_of_parent": g_lab, "K": np.array([0])}
    buf = io.BytesIO()
    pickle.dump(result, buf)
    loaded = pickle.loads(buf.getvalue())

    # ---- conditional_sampling_by_group_size: a group size per synthetic parent from its cluster's empirical distribution,
    # ys = the cluster label repeated over the group (:866-874), then batches of sample_batch_size through the guided chain
    synth_parent_labels = g_lab[rng.permutation(n_parent)[:150]]
    sizes = [int(rng.choice(list(len_prob[l].keys()), p=list(len_prob[l].values()))) for l in synth_parent_labels.tolist()]
    ys = np.repeat(synth_parent_labels, sizes)
    assert len(ys) == sum(sizes)

    def sample_with(res):
        cf = make_cond_fn(res["classifier"], 1.0)
        return parallel.conditional_sample_rows(res["diffusion"], ys, cf, sample_batch_size=128, seed=77, rank=0, world=1, gather=False)

    x_live, lab_live = sample_with(result)
    x_load, lab_load = sample_with(loaded)
    assert x_live.shape == (len(ys), 3) and x_live.dtype == torch.float64 and bool(torch.isfinite(x_live).all())
    assert torch.equal(lab_live.reshape(-1), torch.from_numpy(ys)) and torch.equal(lab_load, lab_live)
    assert torch.equal(x_live, x_load)                                        # the unpickled objects sample bit-identically

    # ---- decoding (:912-934): inverse quantile transform on the device, integer-valued columns snapped to real values
    decoded = postprocess.inverse_transform(loaded["num_transform"], x_load, device=DEV)
    ref = qt.inverse_transform(x_load.numpy())
    assert np.allclose(decoded.cpu().numpy(), ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    cols = postprocess.discrete_columns(child)
    assert cols == [2]
    postprocess.round_columns(child, decoded, cols)
    out = decoded.cpu().numpy()
    assert np.isfinite(out).all() and set(np.unique(out[:, 2])) <= set(np.unique(child[:, 2]))
    # guidance did something: the sampled rows of a cluster sit closer to that cluster's real rows than to the table at large
    real_means = np.stack([child[assign.reshape(-1) == l].mean(0) if (assign.reshape(-1) == l).any() else np.zeros(3) for l in range(C)])
    d_own = np.linalg.norm(out[:, :2] - real_means[ys][:, :2], axis=1).mean()
    d_any = np.linalg.norm(out[:, None, :2] - real_means[None, :, :2], axis=2).mean()
    print(f"pipeline test: {N} child rows, {C} clusters, {len(ys)} sampled rows; mean distance to own cluster {d_own:.3f}, to any {d_any:.3f}")
