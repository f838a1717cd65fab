from trebi_amd.nets import TemporalUnet, ValueFunction  # noqa: F401
from trebi_amd.diffusion import GaussianDiffusion, ValueDiffusion, Sample  # noqa: F401
from . import helpers, temporal, diffusion  # noqa: F401
