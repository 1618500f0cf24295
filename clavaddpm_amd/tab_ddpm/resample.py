"""Timestep sampler for classifier training: the `'uniform'` sampler ClavaDDPM builds at `pipeline_utils.py:779-781`
(reference `tab_ddpm/resample.py`, whose loss-aware samplers and their dead `dist.all_gather` calls are not provided).

Contract kept: `create_named_schedule_sampler('uniform', diffusion).sample(batch_size, device)` returns
`(timesteps int64 [B], importance weights float32 [B])` on `device`, and the timesteps come from ONE
`np.random.choice(T, size=(B,), p=uniform)` call on NumPy's global generator, so `np.random.seed(s)` reproduces the
reference's draws exactly.  `ClassifierTrainer` draws on the device instead unless a sampler is handed to it.
"""
import numpy as np
import torch


class UniformSampler:
    """Every timestep of the diffusion is equally likely; the importance weight of each draw is therefore 1."""

    def __init__(self, diffusion):
        self.diffusion = diffusion
        self.num_timesteps = int(diffusion.num_timesteps)

    def weights(self):
        return np.full(self.num_timesteps, 1.0)

    def sample(self, batch_size, device):
        T = self.num_timesteps
        prob = self.weights() / float(T)                       # == w / w.sum(): the sum of T ones is exactly T
        drawn = np.random.choice(T, size=(batch_size,), p=prob)
        t = torch.as_tensor(drawn, dtype=torch.long).to(device)
        # 1 / (T * p[t]) of the general importance sampler; with p = 1/T it is exactly one
        importance = torch.as_tensor(1.0 / (T * prob[drawn]), dtype=torch.float32).to(device)
        return t, importance


def create_named_schedule_sampler(name, diffusion):
    if name != "uniform":
        raise NotImplementedError(f"unknown schedule sampler: {name} (ClavaDDPM only uses 'uniform')")
    return UniformSampler(diffusion)
