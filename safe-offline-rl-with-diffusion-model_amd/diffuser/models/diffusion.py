from trebi_amd.diffusion import (  # noqa: F401
    GaussianDiffusion, ValueDiffusion, Sample, default_sample_fn, sort_by_values, max_by_values_in_batch,
    make_timesteps)
