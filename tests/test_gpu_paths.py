"""Every launch strategy of the library produces the reference's chains.

The planner picks between single-layer launches (split-K conv kernel), resident chains (direct and
split-K stages, one or two candidates per workgroup) and single-layer direct launches by batch size;
diagnostic environment switches force each of them.  The switches are read once per process, so each
configuration runs the golden-loop check (`test_fused_loop_golden`, tolerance 5e-4 as stated there) in ONE
child process, sequentially."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CONFIGS = {
    'no_chains': {'TREBI_CHAIN': '0'},
    'split_k_chains_only': {'TREBI_CHAIN_DIRECT': '0'},
    'wide_chains_two_per_workgroup': {'TREBI_CHAIN_MAXN': '256', 'TREBI_CHAIN_S': '2'},
    'direct_single_layers': {'TREBI_DIRECT_SINGLE': '1', 'TREBI_CHAIN': '0'},
    # debug aid of SURVEY.md section 5: workspaces NaN-filled before every pass -- nothing may read what it did not write
    'value_head_as_its_own_kernel': {'TREBI_CHAIN_HEAD': '0'},
    # narrow.hip: the persistent 32 / 64-channel kernel takes its layers from 512 candidates up; here at every batch size
    'persistent_narrow_layers_at_every_batch': {'TREBI_NARROW_MINB': '1', 'TREBI_CHAIN': '0'},
    'poisoned_workspaces': {'TREBI_POISON': '1'},
    'poisoned_workspaces_no_chains': {'TREBI_POISON': '1', 'TREBI_CHAIN': '0'},
}


@pytest.mark.parametrize('name', list(CONFIGS))
def test_golden_loops_under_path(name):
    env = dict(os.environ, **CONFIGS[name])
    cmd = [sys.executable, '-m', 'pytest', os.path.join(HERE, 'test_gpu_sampling.py'), '-q', '-x', '-m', 'gpu',
           '-k', 'fused_loop_golden', '-p', 'no:cacheprovider', '--timeout', '120', '--timeout-method', 'thread']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout


def test_full_size_copies_with_the_tap_major_loop():
    """From 4096 candidates up the position-major tiles of the 5-tap layers run the channel-block-major K loop (conv_body W5,
    DESIGN section 4 (v)); `test_gpu_fullsize.py` covers it at 8192 candidates under the default policy.  Here the same
    full-size check runs with that loop switched off, so the tap-major loop it replaced stays pinned to the golden chains too."""
    env = dict(os.environ, TREBI_WIN5='0')
    cmd = [sys.executable, '-m', 'pytest', os.path.join(HERE, 'test_gpu_fullsize.py'), '-q', '-x', '-m', 'gpu',
           '-k', 'walker148', '-p', 'no:cacheprovider', '--timeout', '200', '--timeout-method', 'thread']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout


def _child(files, env, k=None, timeout=600):
    cmd = [sys.executable, '-m', 'pytest'] + [os.path.join(HERE, f) for f in files] + \
          ['-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', '--timeout', '200', '--timeout-method', 'thread']
    if k:
        cmd += ['-k', k]
    r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout


def test_units_nets_and_training_with_the_persistent_narrow_kernel():
    """narrow.hip at small batches: every 32 / 64-channel Conv1dBlock and input-gradient layer (forward with saved operands, backward
    behind the GroupNorm + Mish prologue, the training step's plain input gradients) against its oracle unit (test_gpu_ops.py), the
    golden nets and guide gradients (test_gpu_nets.py) and the training goldens (test_gpu_train.py)."""
    _child(['test_gpu_ops.py', 'test_gpu_nets.py', 'test_gpu_train.py'], {'TREBI_NARROW_MINB': '1'})


def test_full_size_copies_without_the_persistent_narrow_kernel():
    """test_gpu_fullsize.py runs the narrow layers of 8192 / 4100 candidates through narrow.hip (default from 512 candidates up);
    with it switched off the tiles it replaced stay pinned to the golden chains too."""
    _child(['test_gpu_fullsize.py'], {'TREBI_NARROW': '0'}, k='walker148 or 4100 or hopper')
