"""ctypes binding of libtrebi_hip.so (the C ABI in include/trebi_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is
raised.  PyTorch only supplies device memory and the current stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TREBI_LIB') or os.path.join(_HERE, 'libtrebi_hip.so')   # TREBI_LIB: A/B builds on one box (diagnostics)

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)


class NetDesc(C.Structure):
    _fields_ = [('kind', C.c_int), ('horizon', C.c_int), ('transition_dim', C.c_int), ('dim', C.c_int),
                ('n_mults', C.c_int), ('dim_mults', C.c_int * 8), ('max_timesteps', C.c_int)]


class StepDesc(C.Structure):
    _fields_ = [('horizon', C.c_int), ('transition_dim', C.c_int), ('action_dim', C.c_int),
                ('observation_dim', C.c_int), ('n_timesteps', C.c_int), ('scale', C.c_float),
                ('t_stopgrad', C.c_int), ('scale_grad_by_std', C.c_int), ('state_grad_mask', C.c_int),
                ('cost_grad_weight', C.c_float), ('guide_type', C.c_int), ('n_cond', C.c_int),
                ('cond_index', C.c_int * 4)]


class SamplerDesc(C.Structure):
    _fields_ = [('step', StepDesc), ('kind', C.c_int), ('n_guide_steps', C.c_int), ('use_graph', C.c_int),
                ('skip_dead_evals', C.c_int)]


class TrainDesc(C.Structure):
    _fields_ = [('horizon', C.c_int), ('observation_dim', C.c_int), ('action_dim', C.c_int), ('n_timesteps', C.c_int),
                ('dim', C.c_int), ('n_mults', C.c_int), ('dim_mults', C.c_int * 8), ('predict_epsilon', C.c_int),
                ('loss_l1', C.c_int), ('n_cond', C.c_int), ('cond_index', C.c_int * 4)]


class TensorRef(C.Structure):
    _fields_ = [('where', C.c_int), ('L', C.c_int), ('C', C.c_int), ('ld', C.c_int), ('offset_floats', C.c_size_t)]


class LayerInfo(C.Structure):
    _fields_ = [('name', C.c_char * 120), ('kind', C.c_int), ('Lin', C.c_int), ('Lout', C.c_int), ('Cin', C.c_int),
                ('Cin2', C.c_int), ('N', C.c_int), ('stride', C.c_int), ('pad', C.c_int), ('phase_split', C.c_int),
                ('group_norm', C.c_int), ('time_bias', C.c_int), ('prologue', C.c_int),
                ('src', TensorRef), ('src2', TensorRef), ('res', TensorRef), ('dst', TensorRef),
                ('saved_xhat', TensorRef), ('saved_gain', TensorRef), ('saved_stats', TensorRef),
                ('pro_xhat', TensorRef), ('pro_gain', TensorRef), ('pro_stats', TensorRef)]


# name -> (restype, argtypes); every symbol the header declares
_VP = C.c_void_p
SIGNATURES = {
    'trebi_last_error': (C.c_char_p, []),
    'trebi_version': (C.c_int, []),
    'trebi_ctx_create': (C.c_int, [C.c_int, C.POINTER(_VP)]),
    'trebi_ctx_destroy': (None, [_VP]),
    'trebi_net_create': (C.c_int, [_VP, C.POINTER(NetDesc), C.POINTER(C.c_char_p), C.POINTER(_VP),
                                   c_i64_p, C.c_int, C.POINTER(_VP)]),
    'trebi_net_destroy': (None, [_VP]),
    'trebi_net_set_precision': (C.c_int, [_VP, C.c_int, C.c_int]),
    'trebi_trainer_set_precision': (C.c_int, [_VP, C.c_int, C.c_int]),
    'trebi_net_param_count': (C.c_int, [C.POINTER(NetDesc)]),
    'trebi_net_param_info': (C.c_int, [C.POINTER(NetDesc), C.c_int, C.c_char_p, C.c_size_t, c_i64_p]),
    'trebi_net_flops': (C.c_double, [_VP, C.c_int]),
    'trebi_unet_workspace_bytes': (C.c_size_t, [_VP, C.c_int]),
    'trebi_unet_forward': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    'trebi_unet_tap': (C.c_int, [_VP, C.c_char_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]),
    'trebi_net_layer_count': (C.c_int, [_VP, C.c_int]),
    'trebi_net_layer_info': (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.POINTER(LayerInfo)]),
    'trebi_unet_forward_static': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, C.c_int, C.c_int, _VP]),
    'trebi_value_workspace_bytes': (C.c_size_t, [_VP, C.c_int]),
    'trebi_value_gradients': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t,
                                        _VP]),
    'trebi_guide_update': (C.c_int, [_VP, C.POINTER(StepDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                     C.POINTER(_VP), C.c_int, _VP]),
    'trebi_ddpm_update': (C.c_int, [_VP, C.POINTER(StepDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                    C.POINTER(_VP), C.c_int, _VP]),
    'trebi_feasibility_rule': (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, _VP, _VP]),
    'trebi_argsort_desc': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    'trebi_group_argmax': (C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, _VP]),
    'trebi_policy_prepare': (C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, _VP, C.c_int, _VP, _VP, C.c_int, _VP, _VP]),
    'trebi_policy_first_action': (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_int, _VP, _VP, _VP,
                                            _VP]),
    'trebi_sampler_create': (C.c_int, [_VP, _VP, _VP, _VP, C.POINTER(SamplerDesc), c_float_p, c_float_p,
                                       c_float_p, C.c_int, C.POINTER(_VP)]),
    'trebi_sampler_destroy': (None, [_VP]),
    'trebi_sampler_bind': (C.c_int, [_VP, _VP, C.POINTER(_VP), _VP, _VP, _VP, _VP, _VP]),
    'trebi_sampler_run': (C.c_int, [_VP, _VP, C.POINTER(_VP), _VP, _VP, _VP, _VP, _VP, _VP]),
    'trebi_sampler_set_profiling': (C.c_int, [_VP, C.c_int]),
    'trebi_sampler_get_profile': (C.c_int, [_VP, C.POINTER(C.c_double), c_i64_p, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double)]),
    'trebi_trainer_create': (C.c_int, [_VP, C.POINTER(TrainDesc), C.c_int, C.POINTER(_VP)]),
    'trebi_trainer_destroy': (None, [_VP]),
    'trebi_trainer_param_count': (C.c_int64, [_VP]),
    'trebi_trainer_flops': (C.c_double, [_VP, C.c_int]),
    'trebi_trainer_bind': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, c_float_p, c_float_p, c_float_p]),
    'trebi_trainer_repack': (C.c_int, [_VP, _VP]),
    'trebi_trainer_bucket_split': (C.c_int64, [_VP]),
    'trebi_trainer_set_bucket_event': (C.c_int, [_VP, _VP]),
    'trebi_trainer_accumulate': (C.c_int, [_VP, _VP, C.POINTER(_VP), _VP, _VP, C.c_int, C.c_float, _VP, _VP]),
    'trebi_trainer_step': (C.c_int, [_VP, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float,
                                     _VP]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Fails loudly: there is no other code path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with __graft_entry__.build() or '
                f'`make -C safe-offline-rl-with-diffusion-model_amd/csrc`. There is no fallback path.')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().trebi_last_error()
        raise RuntimeError(f'trebi_hip error {rc}: {msg.decode() if msg else "?"}')


_ctx = {}


def ctx(device_index):
    """One trebi_ctx per device, kept for the life of the process."""
    if device_index not in _ctx:
        h = _VP()
        check(lib().trebi_ctx_create(int(device_index), C.byref(h)))
        _ctx[device_index] = h
    return _ctx[device_index]


def ptr(t):
    """Device (or host) address of a torch tensor, or None."""
    return None if t is None else _VP(t.data_ptr())


def ptr_array(tensors):
    arr = (_VP * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
