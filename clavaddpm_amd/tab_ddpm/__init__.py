from . import logger  # noqa: F401
from .gaussian_multinomial_diffsuion import GaussianMultinomialDiffusion, get_named_beta_schedule, betas_for_alpha_bar  # noqa: F401
from .modules import MLP, MLPDiffusion, timestep_embedding  # noqa: F401
from .resample import create_named_schedule_sampler  # noqa: F401
from .utils import FoundNANsError  # noqa: F401
