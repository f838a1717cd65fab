from trebi_amd.diffusion import extract, cosine_beta_schedule, apply_conditioning  # noqa: F401
