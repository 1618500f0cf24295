"""Deterministic parameter fill shared by the golden-vector generator and the tests.

Large networks are not stored in fixtures; both sides rebuild identical weights
from a seed with this function instead.
"""
import numpy as np
import torch


def seeded_state(shapes, seed, scale=1.0):
    """shapes: ordered dict name -> shape.  Returns name -> float32 tensor.

    Linear weights ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) * scale, biases likewise,
    BatchNorm statistics kept away from degenerate values.
    """
    rng = np.random.RandomState(seed)
    out = {}
    fan = 1
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(7, dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif name.endswith("running_mean"):
            out[name] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
        elif len(shape) == 2:
            fan = shape[1]
            b = scale / np.sqrt(fan)
            out[name] = torch.from_numpy(rng.uniform(-b, b, shape).astype(np.float32))
        elif len(shape) == 1 and ".weight" in name:          # BatchNorm gamma
            out[name] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        else:                                                # biases / BatchNorm beta
            b = scale / np.sqrt(max(fan, 1))
            out[name] = torch.from_numpy(rng.uniform(-b, b, shape).astype(np.float32))
    return out


def mlp_shapes(d_in, d_layers, dim_t=128):
    """state_dict layout of the reference `MLPDiffusion` (`tab_ddpm/modules.py:425-449`)."""
    s = {}
    dims = [dim_t] + list(d_layers)
    for i in range(len(d_layers)):
        s[f"mlp.blocks.{i}.linear.weight"] = (dims[i + 1], dims[i])
        s[f"mlp.blocks.{i}.linear.bias"] = (dims[i + 1],)
    s["mlp.head.weight"] = (d_in, dims[-1])
    s["mlp.head.bias"] = (d_in,)
    s["proj.weight"] = (dim_t, d_in)
    s["proj.bias"] = (dim_t,)
    for j in (0, 2):
        s[f"time_embed.{j}.weight"] = (dim_t, dim_t)
        s[f"time_embed.{j}.bias"] = (dim_t,)
    return s


def classifier_shapes(d_in, d_out, hidden, dim_t=128):
    """state_dict layout of the live part of the reference `Classifier`
    (`pipeline_utils.py:584-623`; the unused `transformer_layer` is left out)."""
    s = {"proj.weight": (dim_t, d_in), "proj.bias": (dim_t,)}
    for j in (0, 2):
        s[f"time_embed.{j}.weight"] = (dim_t, dim_t)
        s[f"time_embed.{j}.bias"] = (dim_t,)
    dims = [dim_t] + list(hidden)
    for i in range(len(hidden)):
        j = 4 * i
        s[f"model.{j}.weight"] = (dims[i + 1], dims[i])
        s[f"model.{j}.bias"] = (dims[i + 1],)
        s[f"model.{j + 2}.weight"] = (dims[i + 1],)
        s[f"model.{j + 2}.bias"] = (dims[i + 1],)
        s[f"model.{j + 2}.running_mean"] = (dims[i + 1],)
        s[f"model.{j + 2}.running_var"] = (dims[i + 1],)
        s[f"model.{j + 2}.num_batches_tracked"] = ()
    j = 4 * len(hidden)
    s[f"model.{j}.weight"] = (d_out, dims[-1])
    s[f"model.{j}.bias"] = (d_out,)
    return s
