"""Reading the reference's checkpoints into the HIP modules.

The reference's Trainer writes `state_{epoch}.pt` = {'step', 'model', 'ema'} of plain state_dicts
(diffuser/utils/training.py:140-166) and loads `data['ema']` into the EMA copy of the GaussianDiffusion /
ValueDiffusion that planning uses.  The modules here keep the reference's parameter and buffer names, so the file
loads with `strict=True`.  The file is opened with `weights_only=True`: nothing in it is executed.  The pickled
`*_config.pkl` files next to it (diffuser/utils/serialization.py:36-60) hold class objects and are NOT read: build the
modules with the constructor arguments of the run (config/locomotion.py) instead."""
import torch


def load_state(path, key='ema'):
    """The `key` state_dict ('ema' is what planning uses, 'model' the raw weights) and the trainer step."""
    data = torch.load(path, map_location='cpu', weights_only=True)
    if not isinstance(data, dict) or key not in data:
        raise RuntimeError(f'{path}: not a reference trainer checkpoint (expected a dict with {key!r})')
    return data[key], int(data.get('step', 0))


def load_into(module, path, key='ema', strict=True):
    """module: a GaussianDiffusion / ValueDiffusion from this package, built like the checkpoint's run.
    Returns the trainer step.  Mismatching shapes or names raise (strict) -- never silently skipped."""
    state, step = load_state(path, key)
    module.load_state_dict(state, strict=strict)
    return step
