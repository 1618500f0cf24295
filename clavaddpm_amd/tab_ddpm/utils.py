"""Small host-side helpers kept importable under the reference's names (`tab_ddpm/utils.py`)."""
import torch


class FoundNANsError(BaseException):
    """Raised by `sample_all` when a sampled row contains NaN.  Derives from BaseException like the
    reference's (`tab_ddpm/utils.py:171-174`), so a bare `except Exception` does not swallow it."""

    def __init__(self, message="Found NANs during sampling."):
        super().__init__(message)


def ohe_to_categories(ohe, K):
    """Per-feature argmax over side-by-side one-hot blocks (`tab_ddpm/utils.py:92-98`); host utility."""
    edges = [0]
    for k in K:
        edges.append(edges[-1] + int(k))
    return torch.stack([ohe[:, a:b].argmax(dim=1) for a, b in zip(edges[:-1], edges[1:])], dim=1)
