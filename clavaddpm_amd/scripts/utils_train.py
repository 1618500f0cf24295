"""`get_model` / `update_ema` under the reference's names (`scripts/utils_train.py:7-29`).

`make_dataset_from_df` (host-side pandas / scikit-learn preparation, runs once per table) is out of scope of the HIP
path and stays with the reference pipeline.
"""
from ..tab_ddpm.modules import MLPDiffusion


def get_model(model_name, model_params):
    """Model factory (`:7-18`): only 'mlp' is live in ClavaDDPM (every config sets "model_type": "mlp")."""
    print(model_name)
    if model_name == "mlp":
        return MLPDiffusion(**model_params)
    raise ValueError(f"Unknown model: {model_name!r} (the reference's 'resnet' denoiser is broken upstream and unused)")


def update_ema(target_params, source_params, rate=0.999):
    """In-place EMA of parameters (`:20-29`).  The fused Trainer does this inside clava_adamw_ema_f32; this
    per-tensor form is kept for callers that use it directly (a few tiny in-place torch ops per parameter)."""
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src.detach(), alpha=1 - rate)
