from trebi_amd.diffusion import n_step_guided_p_sample, n_step_guided_and_constrained_p_sample  # noqa: F401
