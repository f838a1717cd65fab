from trebi_amd.diffusion import ValueGuide  # noqa: F401
