from trebi_amd.policy import GuidedPolicy, Trajectories  # noqa: F401
