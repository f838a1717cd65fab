from trebi_amd.diffusion import (  # noqa: F401
    ValueGuide, n_step_guided_p_sample, n_step_guided_and_constrained_p_sample)
from . import guides, functions  # noqa: F401
from trebi_amd.policy import GuidedPolicy, Trajectories  # noqa: F401,E402
