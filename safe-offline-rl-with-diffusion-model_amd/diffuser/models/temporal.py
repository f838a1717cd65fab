from trebi_amd.nets import TemporalUnet, ValueFunction  # noqa: F401
