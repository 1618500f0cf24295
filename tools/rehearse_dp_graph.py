#!/usr/bin/env python3
"""Rehearsal of the GRAPH-CAPTURED data-parallel training step on a box with one GPU.

    python tools/rehearse_dp_graph.py [--steps 40] [--out profiles/r03_dp_graph_rehearsal.json]

Two ranks under torch.distributed.run (gloo, both on GPU 0, every rank its own batches, device draws), once with the
captured step (default) and once eager (CLAVA_TRAIN_GRAPH_DP=0).  Checks: the ranks' weights are BIT-EQUAL after every run
(same all-reduced gradient, same fused AdamW); the captured run's weights follow the eager run's (same draws: both key the
library's generator by (seed, step word)); reports host enqueue time and wall time per step of both.  The launcher itself
never touches the GPU."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, faulthandler
sys.path.insert(0, os.environ["CLAVA_ROOT"])
faulthandler.dump_traceback_later(200, exit=True)
import numpy as np, torch
import torch.distributed as dist
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
from clavaddpm_amd import split_nets
from clavaddpm_amd.scripts.train import Trainer
from clavaddpm_amd.tab_ddpm import GaussianMultinomialDiffusion, MLPDiffusion
split_nets.set_gemm_mode("f16x3")
steps, B, n_num, T = int(os.environ["CLAVA_REHEARSAL_STEPS"]), 4096, 2, 1000
torch.manual_seed(0)
net = MLPDiffusion(n_num, 0, "none", {"d_layers": [512, 1024, 1024, 1024, 1024, 512], "dropout": 0.0}).to("cuda").train()
diff = GaussianMultinomialDiffusion(np.array([0]), n_num, net, num_timesteps=T, device=torch.device("cuda")).to("cuda").train()
tr = Trainer(diff, None, lr=1e-3, weight_decay=1e-5, steps=10 ** 6, device=torch.device("cuda"))
tr.eps = 1e-4
g = torch.Generator().manual_seed(100 + rank)
data = [torch.randn(B, n_num, generator=g) for _ in range(8)]
y = {"y": torch.zeros(B, dtype=torch.long)}
# gloo reduces through the host, synchronously (17 MB down, reduce, 17 MB up): that time is the rehearsal's, not the
# step's - RCCL enqueues the collective and returns - so it is measured apart from the rest of the host work
from clavaddpm_amd import parallel as _par
_real_ar, ar_time = _par.allreduce_sum_, [0.0]
def _timed_ar(t):
    t0 = time.perf_counter(); _real_ar(t); ar_time[0] += time.perf_counter() - t0
_par.allreduce_sum_ = _timed_ar
enq = []
t_wall = None
for i in range(steps):
    if i == steps // 2:
        torch.cuda.synchronize(); dist.barrier(); t_wall = time.perf_counter()
    t0 = time.perf_counter(); ar_time[0] = 0.0
    tr._run_step(data[i % 8], y)
    enq.append(time.perf_counter() - t0 - ar_time[0])
torch.cuda.synchronize(); dist.barrier()
wall = (time.perf_counter() - t_wall) / (steps - steps // 2)
# ---- the guidance classifier's trainer, same check
from clavaddpm_amd.classifier_training import ClassifierTrainer
from clavaddpm_amd.pipeline_utils import Classifier
torch.manual_seed(1)
clf = Classifier(d_in=n_num, d_out=50, dim_t=128, hidden_sizes=[128, 256, 512, 1024, 512, 256, 128]).to("cuda").train()
sched = GaussianMultinomialDiffusion(np.array([0]), n_num, None, num_timesteps=T, device=torch.device("cuda")).to("cuda")
ct = ClassifierTrainer(clf, sched, lr=1e-4)
labels = [torch.randint(0, 50, (B,), generator=g) for _ in range(8)]
c_enq = []
for i in range(steps):
    t0 = time.perf_counter(); ar_time[0] = 0.0
    ct.step(data[i % 8], labels[i % 8])
    c_enq.append(time.perf_counter() - t0 - ar_time[0])
torch.cuda.synchronize(); dist.barrier()
np.savez(os.path.join(os.environ["CLAVA_REHEARSAL_DIR"], f"{os.environ['CLAVA_REHEARSAL_TAG']}_r{rank}.npz"), flat=tr.flat.cpu().numpy(),
         clf_flat=ct.flat.cpu().numpy(), clf_enq=np.array(c_enq), clf_captured=np.array([int(bool(getattr(ct, "_graphs", {})))]),
         ema=tr.ema_flat.cpu().numpy(), enq=np.array(enq), wall=np.array([wall]),
         captured=np.array([int(any(sg["graph"] is not None for sg in getattr(tr, "_stages", {}).values()))]))
dist.barrier()
dist.destroy_process_group()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_dp_graph_rehearsal.json"))
    args = ap.parse_args()
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        wpath = os.path.join(tmp, "worker.py")
        with open(wpath, "w") as f:
            f.write(WORKER)
        res = {}
        for tag, flag in (("graph", "1"), ("eager", "0")):
            env = dict(os.environ, CLAVA_ROOT=ROOT, CLAVA_REHEARSAL_DIR=tmp, CLAVA_REHEARSAL_STEPS=str(args.steps), CLAVA_REHEARSAL_TAG=tag,
                       CLAVA_TRAIN_GRAPH_DP=flag, MASTER_ADDR="127.0.0.1")
            subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(29400 + (os.getpid() + len(res)) % 500), wpath], env=env, check=True, timeout=500)
            res[tag] = [np.load(os.path.join(tmp, f"{tag}_r{r}.npz")) for r in range(2)]
        out = {"what": "2-rank data-parallel Trainer (gloo, both ranks on one MI355X), captured step vs eager step, device draws",
               "steps": args.steps, "batch_per_rank": 4096}
        for tag, (r0, r1) in res.items():
            half = args.steps // 2
            out[tag] = {"ranks_bit_equal": bool(np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["ema"], r1["ema"])),
                        "captured": bool(r0["captured"][0]),
                        "host_enqueue_ms_per_step_without_the_gloo_allreduce": round(1e3 * float(np.median(r0["enq"][half:])), 3),
                        "wall_ms_per_step_two_ranks_one_gpu": round(1e3 * float(r0["wall"][0]), 3)}
            out[tag]["classifier"] = {"ranks_bit_equal": bool(np.array_equal(r0["clf_flat"], r1["clf_flat"])), "captured": bool(r0["clf_captured"][0]),
                                      "host_enqueue_ms_per_step_without_the_gloo_allreduce": round(1e3 * float(np.median(r0["clf_enq"][half:])), 3)}
        out["classifier_graph_vs_eager_max_abs_weight_dev"] = float(np.abs(res["graph"][0]["clf_flat"] - res["eager"][0]["clf_flat"]).max())
        a, b = res["graph"][0]["flat"], res["eager"][0]["flat"]
        out["graph_vs_eager_max_abs_weight_dev"] = float(np.abs(a - b).max())
        out["max_abs_weight"] = float(np.abs(b).max())
    print(json.dumps(out))
    assert out["graph"]["ranks_bit_equal"] and out["eager"]["ranks_bit_equal"], "ranks diverged"
    assert out["graph"]["captured"] and not out["eager"]["captured"]
    assert out["graph_vs_eager_max_abs_weight_dev"] <= 1e-4, out["graph_vs_eager_max_abs_weight_dev"]
    assert out["graph"]["classifier"]["ranks_bit_equal"] and out["eager"]["classifier"]["ranks_bit_equal"], "classifier ranks diverged"
    assert out["graph"]["classifier"]["captured"] and not out["eager"]["classifier"]["captured"]
    assert out["classifier_graph_vs_eager_max_abs_weight_dev"] <= 1e-4, out["classifier_graph_vs_eager_max_abs_weight_dev"]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
