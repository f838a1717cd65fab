"""CPU suite: the C-ABI library loads without a GPU and exports every symbol include/trebi_hip.h
declares; the ctypes binding covers the same set; and the pure-host entry points (parameter table,
argument validation) behave.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from trebi_amd import _lib, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'trebi_hip.h')


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(trebi_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in trebi_hip.h but not exported'


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == header_functions()
    assert _lib.lib().trebi_version() >= 1


def _desc(kind, H, D, dim, mults, T=20):
    d = _lib.NetDesc()
    d.kind, d.horizon, d.transition_dim, d.dim, d.n_mults, d.max_timesteps = kind, H, D, dim, len(mults), T
    for i, m in enumerate(mults):
        d.dim_mults[i] = m
    return d


@pytest.mark.parametrize('kind,H,D,mults', [(0, 32, 23, (1, 2, 4, 8)), (0, 32, 23, (1, 4, 8)), (0, 4, 23, (1, 4, 8)),
                                            (0, 128, 4, (1, 2, 4, 8)), (1, 32, 23, (1, 2, 4, 8)),
                                            (1, 32, 14, (1, 2, 4, 8)), (1, 4, 23, (1, 4, 8))])
def test_parameter_table_matches_python_spec(kind, H, D, mults):
    """The library's idea of the reference state_dict (names, order, sizes) == trebi_amd.spec, which the
    golden generator proved equal to the reference modules' own state_dict()."""
    lib = _lib.lib()
    d = _desc(kind, H, D, 32, mults)
    shapes = (spec.unet_param_shapes if kind == 0 else spec.value_param_shapes)(H, D, 32, mults)
    n = lib.trebi_net_param_count(C.byref(d))
    assert n == len(shapes)
    buf = C.create_string_buffer(256)
    numel = C.c_int64()
    for i, (name, shape) in enumerate(shapes.items()):
        assert lib.trebi_net_param_info(C.byref(d), i, buf, 256, C.byref(numel)) == 0
        assert buf.value.decode() == name
        want = 1
        for s in shape:
            want *= s
        assert numel.value == want, name


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.lib()
    d = _desc(2, 32, 23, 32, (1, 2, 4, 8))
    assert lib.trebi_net_param_count(C.byref(d)) == -1           # TREBI_BAD_ARG: unknown kind
    assert b'kind' in lib.trebi_last_error()
    d = _desc(0, 32, 23, 12, (1, 2))
    assert lib.trebi_net_param_count(C.byref(d)) == -3           # TREBI_UNSUPPORTED_CONFIG: GroupNorm(8)
    with pytest.raises(RuntimeError):
        _lib.check(lib.trebi_sampler_set_profiling(None, 1))
    h = C.c_void_p()
    assert lib.trebi_ctx_create(0, None) == -1
    # no GPU in the CPU container: context creation must fail loudly, never fall back
    rc = lib.trebi_ctx_create(0, C.byref(h))
    if rc == 0:
        lib.trebi_ctx_destroy(h)
    else:
        assert rc in (-1, -3, -5) and lib.trebi_last_error()
