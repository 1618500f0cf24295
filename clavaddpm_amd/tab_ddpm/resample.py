"""Timestep samplers for classifier training (`tab_ddpm/resample.py:10-69` of the reference).

Only the sampler ClavaDDPM uses is provided: `create_named_schedule_sampler('uniform', diffusion)`
(`pipeline_utils.py:779-781`).  `sample` keeps the reference's host-side draw (`np.random.choice`, so a seeded NumPy
generator reproduces the reference's timesteps) and returns `(timesteps long[B], weights float[B])` on `device`;
`ClassifierTrainer` draws on the device instead unless a sampler is handed to it.
"""
import numpy as np
import torch


class UniformSampler:
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        indices_np = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = torch.from_numpy(indices_np).long().to(device)
        weights = torch.from_numpy(1 / (len(p) * p[indices_np])).float().to(device)
        return indices, weights


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name} (ClavaDDPM only uses 'uniform')")
