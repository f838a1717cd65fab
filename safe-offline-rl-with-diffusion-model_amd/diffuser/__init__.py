"""Drop-in alias package: the reference resolves its hot-path classes from dotted strings such as
'models.TemporalUnet' or 'sampling.ValueGuide' under the `diffuser` package
(/root/reference/config/locomotion.py:23-24,68-69,154-155; diffuser/utils/config.py:6-19).
Putting safe-offline-rl-with-diffusion-model_amd/ on sys.path makes those names resolve to the
MI355X implementations."""
from . import models, sampling, utils  # noqa: F401
