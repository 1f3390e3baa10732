"""ctypes binding of libdreamer_hip.so (the C-ABI declared in include/dreamer_hip.h).

No torch types cross the boundary: tensors are passed as raw device pointers (`tensor.data_ptr()`), the
stream as `torch.cuda.current_stream().cuda_stream`.  Every call checks the return code and raises
`DreamerHipError` with the library's thread-local message.  There is no CPU fallback: if the shared
library is missing, or a tensor is not a contiguous CUDA(HIP) fp32 tensor, the call fails loudly.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DM_LIB_PATH') or os.path.join(_HERE, 'libdreamer_hip.so')      # (DM_LIB_PATH: A/B of two builds on one box)

DM_MAX_MLP_LAYERS = 8
DM_GEMM_ACCUM = 1
DM_GEMM_ELU = 2
DM_C2I_ELU = 1
DM_FLAG_IMAGE_U8 = 16
DM_FLAG_BF16 = 128            # dm_shape.flags: this call runs its contractions on bf16 operands (conf.amp)
DM_GEMM_BF16 = 256            # dm_gemm_f32 flags: the same for a single product
DM_SPLITK_FLOATS = 16 * 1024 * 1024    # split-K partial region carved at the front of every operator workspace

RSSM_PARAM_ORDER = [
    'z_mlp.weight', 'z_mlp.bias', 'a_mlp.weight', 'in_norm.weight', 'in_norm.bias',
    'gru.layers.0.weight_ih', 'gru.layers.0.weight_hh', 'gru.layers.0.bias_ih', 'gru.layers.0.bias_hh',
    'prior_mlp_h.weight', 'prior_mlp_h.bias', 'prior_norm.weight', 'prior_norm.bias', 'prior_mlp.weight', 'prior_mlp.bias',
    'post_mlp_h.weight', 'post_mlp_h.bias', 'post_mlp_e.weight', 'post_norm.weight', 'post_norm.bias',
    'post_mlp.weight', 'post_mlp.bias',
]
# the LayerNorm GRU cells (rnn.py:95-138): no gate biases, LayerNorm parameters in the 6 extra slots (include/dreamer_hip.h)
GRU_KINDS = {'gru': 0, 'gru_layernorm': 1, 'gru_layernorm_dv2': 2}
DM_FLAG_GRU_SHIFT = 5
_LN_SLOTS = {
    'gru': [None] * 6,
    'gru_layernorm': ['ln_reset.weight', 'ln_reset.bias', 'ln_update.weight', 'ln_update.bias', 'ln_newval.weight', 'ln_newval.bias'],
    'gru_layernorm_dv2': ['lnorm.weight', 'lnorm.bias', None, None, None, None],
}


DM_FLAG_GRU_LAYERS_SHIFT = 8
DM_MAX_GRU_LAYERS = 4


def rssm_param_names(gru_type='gru', gru_layers=1):
    """Parameter name (relative to wm.core.cell) of every dm_rssm_params slot, None for slots the cell does not have."""
    names = list(RSSM_PARAM_ORDER)
    if gru_type != 'gru':
        ren = {'gru.layers.0.weight_ih': 'gru.layers.0.weight_ih.weight', 'gru.layers.0.weight_hh': 'gru.layers.0.weight_hh.weight',
               'gru.layers.0.bias_ih': None, 'gru.layers.0.bias_hh': None}
        names = [ren.get(n, n) for n in names]
    ln = lambda i: [n and f'gru.layers.{i}.{n}' for n in _LN_SLOTS[gru_type]]
    names = names + ln(0)
    cell = ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh') if gru_type == 'gru' else ('weight_ih.weight', 'weight_hh.weight', None, None)
    for i in range(1, DM_MAX_GRU_LAYERS):       # GRUCellStack layers 1..3 (rnn.py:56): DM_RSSM_GRU_L{i}_{WIH,WHH,BIH,BHH}
        names += [f'gru.layers.{i}.{n}' if (n and i < gru_layers) else None for n in cell]
    for i in range(1, DM_MAX_GRU_LAYERS):       # ... and DM_RSSM_GRU_L{i}_LN_* of a stack of LayerNorm cells
        names += ln(i) if i < gru_layers else [None] * 6
    return names


DM_RSSM_NPARAMS = len(RSSM_PARAM_ORDER) + 6 + (4 + 6) * (DM_MAX_GRU_LAYERS - 1)


class DreamerHipError(RuntimeError):
    pass


class dm_shape(Structure):
    _fields_ = [(n, c_int32) for n in ('T', 'B', 'I', 'H', 'D', 'Hd', 'S', 'C', 'E', 'A', 'mlp_hidden', 'mlp_layers',
                                       'cnn_depth', 'img', 'img_ch', 'flags')]


class dm_mlp_params(Structure):
    _fields_ = [('w', c_void_p * (DM_MAX_MLP_LAYERS + 1)), ('b', c_void_p * (DM_MAX_MLP_LAYERS + 1)),
                ('ln_g', c_void_p * DM_MAX_MLP_LAYERS), ('ln_b', c_void_p * DM_MAX_MLP_LAYERS),
                ('precision', c_int32), ('reserved_', c_int32)]      # precision: 0 fp32, 1 bf16 operands (conf.amp)


class dm_mlp_grads(Structure):
    _fields_ = [('w', c_void_p * (DM_MAX_MLP_LAYERS + 1)), ('b', c_void_p * (DM_MAX_MLP_LAYERS + 1)),
                ('ln_g', c_void_p * DM_MAX_MLP_LAYERS), ('ln_b', c_void_p * DM_MAX_MLP_LAYERS)]


class dm_conv_params(Structure):
    _fields_ = [('w', c_void_p * 5), ('b', c_void_p * 5)]


dm_conv_grads = dm_conv_params


class dm_rssm_params(Structure):
    _fields_ = [('p', c_void_p * DM_RSSM_NPARAMS)]


dm_rssm_grads = dm_rssm_params


class dm_reduce_item(Structure):
    _fields_ = [('x', c_void_p), ('n', c_int64), ('scale', c_float), ('mode', c_int32), ('center', c_void_p)]


_P = c_void_p
_SIGNATURES = {
    'dm_version': (c_int, []),
    'dm_last_error': (c_char_p, []),
    'dm_device_check': (c_int, []),
    'dm_workspace_bytes': (c_size_t, [POINTER(dm_shape)]),
    'dm_stream_create_cu_mask': (c_int, [_P, c_int, POINTER(c_void_p)]),
    'dm_rccl_available': (c_int, []),
    'dm_rccl_version': (c_int, []),
    'dm_rccl_unique_id': (c_int, [_P]),
    'dm_rccl_comm_init': (c_int, [POINTER(c_void_p), c_int, _P, c_int]),
    'dm_rccl_comm_destroy': (c_int, [_P]),
    'dm_allreduce_grads': (c_int, [_P, c_size_t, _P, _P]),
    'dm_stream_destroy': (c_int, [_P]),
    'dm_gemm_f32': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P, c_int, c_int,
                            _P, c_size_t, _P]),
    'dm_gemm_bf16h': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    'dm_ln_elu_fwd': (c_int, [c_int, c_int, _P, c_int, _P, _P, c_float, _P, c_int, _P, _P]),
    'dm_ln_elu_bwd': (c_int, [c_int, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, c_size_t, _P]),
    'dm_colsum': (c_int, [c_int, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    'dm_gru_gates_fwd': (c_int, [c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P]),
    'dm_gru_gates_bwd': (c_int, [c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P]),
    'dm_sample_onehot': (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P]),
    'dm_kl_balance_fwd': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'dm_kl_balance_bwd': (c_int, [c_int, c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P]),
    'dm_st_softmax_bwd': (c_int, [c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, _P]),
    'dm_mask_rows': (c_int, [c_int, c_int, _P, c_int, _P, _P, c_int, _P]),
    'dm_kl_sampled_fwd': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'dm_kl_sampled_bwd': (c_int, [c_int, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P]),
    'dm_kl_sampled_gauss_fwd': (c_int, [c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    'dm_kl_sampled_gauss_bwd': (c_int, [c_int, c_int, _P, _P, _P, c_int, c_float, _P, _P, _P, _P, c_int, _P]),
    'dm_reduce_i': (c_int, [c_int, c_int, c_int, _P, c_int, _P, _P, _P]),
    'dm_combine_rows': (c_int, [c_int, c_int64, POINTER(c_void_p), POINTER(c_float), _P, _P]),
    'dm_scale_rows': (c_int, [c_int64, c_int, _P, c_int, _P, c_float, _P]),
    'dm_im2col_s2': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
    'dm_col2im_s2': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P]),
    'dm_mlp_acts_floats': (c_size_t, [c_int, c_int, c_int]),
    'dm_mlp_ws_floats': (c_size_t, [c_int, c_int, c_int]),
    'dm_mlp_head_fwd': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, POINTER(dm_mlp_params), _P, _P, _P,
                                c_size_t, _P]),
    'dm_mlp_head_fwd_sparse': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, POINTER(dm_mlp_params), _P, _P, _P, c_size_t, _P]),
    'dm_mlp_head_bwd': (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, POINTER(dm_mlp_params), _P, _P,
                                POINTER(dm_mlp_grads), _P, c_int, c_int, _P, c_size_t, _P]),
    'dm_head_loss': (c_int, [c_int, c_int, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    'dm_preprocess_image_u8': (c_int, [c_int64, c_int, c_int, _P, _P, _P]),
    'dm_conv_encoder_acts_floats': (c_size_t, [POINTER(dm_shape)]),
    'dm_conv_encoder_fwd': (c_int, [POINTER(dm_shape), _P, POINTER(dm_conv_params), _P, _P, _P, c_size_t, _P]),
    'dm_conv_encoder_fwd_rows': (c_int, [POINTER(dm_shape), c_int, c_int, c_int, _P, POINTER(dm_conv_params), _P, _P, _P,
                                         c_size_t, _P]),
    'dm_conv_encoder_bwd': (c_int, [POINTER(dm_shape), _P, POINTER(dm_conv_params), _P, _P, POINTER(dm_conv_grads), _P,
                                    c_size_t, _P]),
    'dm_conv_decoder_acts_floats': (c_size_t, [POINTER(dm_shape)]),
    'dm_conv_decoder_pred_offset': (c_size_t, [POINTER(dm_shape)]),
    'dm_conv_decoder_mse_fwd': (c_int, [POINTER(dm_shape), _P, c_int, _P, POINTER(dm_conv_params), _P, _P, _P, _P,
                                        c_size_t, _P]),
    'dm_conv_decoder_mse_fwd_rows': (c_int, [POINTER(dm_shape), c_int, c_int, c_int, _P, c_int, _P, POINTER(dm_conv_params),
                                             _P, _P, _P, _P, c_size_t, _P]),
    'dm_conv_decoder_mse_bwd': (c_int, [POINTER(dm_shape), _P, c_int, _P, POINTER(dm_conv_params), _P, c_float,
                                        POINTER(dm_conv_grads), _P, c_int, _P, c_size_t, _P]),
    'dm_conv_decoder_mse_bwd_rows': (c_int, [POINTER(dm_shape), _P, c_int, _P, POINTER(dm_conv_params), _P, c_float, _P,
                                             POINTER(dm_conv_grads), _P, c_int, _P, c_size_t, _P]),
    'dm_rssm_acts_floats': (c_size_t, [POINTER(dm_shape)]),
    'dm_rssm_sequence_fwd': (c_int, [POINTER(dm_shape), _P, _P, _P, _P, _P, _P, _P, POINTER(dm_rssm_params), _P, _P, _P,
                                     _P, _P, _P, c_size_t, _P]),
    'dm_rssm_sequence_fwd_steps': (c_int, [POINTER(dm_shape), c_int, c_int, _P, _P, _P, _P, _P, _P, _P,
                                           POINTER(dm_rssm_params), _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'dm_rssm_sequence_bwd': (c_int, [POINTER(dm_shape), _P, _P, _P, POINTER(dm_rssm_params), _P, _P, _P, _P, _P, _P,
                                     POINTER(dm_rssm_grads), _P, _P, c_size_t, _P]),
    'dm_dream_rollout': (c_int, [POINTER(dm_shape), c_int, _P, POINTER(dm_rssm_params), POINTER(dm_mlp_params), _P, _P,
                                 _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'dm_gae_losses': (c_int, [c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    'dm_actor_loss': (c_int, [c_int, c_int, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    'dm_critic_loss': (c_int, [c_int, _P, _P, _P, c_float, _P, _P, _P]),
    'dm_sample_continuous': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P]),
    'dm_actor_loss_continuous': (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    'dm_multi_sum': (c_int, [c_int, POINTER(dm_reduce_item), _P, _P]),
    'dm_combine': (c_int, [c_int, _P, POINTER(c_float), _P, _P]),
    'dm_multi_tensor_norm_clip': (c_int, [_P, c_int64, c_float, _P, _P, c_size_t, _P]),
    'dm_scale_inplace': (c_int, [_P, c_int64, _P, _P]),
    'dm_adamw_step': (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, _P, _P]),
    'dm_copy_params': (c_int, [_P, _P, c_int64, _P]),
    'dm_axpby': (c_int, [c_int64, c_float, _P, c_float, _P, _P]),
    'dm_prof_begin': (c_int, [c_int]),
    'dm_prof_end': (c_int, [POINTER(ctypes.c_double), c_int]),
    'dm_prof_rows': (c_int, [POINTER(ctypes.c_double), c_int]),
    'dm_mlp_chain_min_rows': (c_int, [c_int]),
    'dm_rollout_fuse_act_enable': (c_int, [c_int]),
    'dm_bf16_twins_enable': (c_int, [c_int]),
    'dm_gemm_dma_enable': (c_int, [c_int]),
    'dm_dec_l4_bwd_direct_enable': (c_int, [c_int]),
    'dm_rssm_lds_enable': (c_int, [c_int]),
    'dm_bptt_fold_enable': (c_int, [c_int]),
    'dm_rssm_lds_status': (c_int, []),
    'dm_rssm_lds_status_ack': (c_int, []),
    'dm_rssm_lds_gave_up': (c_int, []),
    'dm_rssm_lds_prof': (c_int, [_P, c_int]),
    'dm_wgrad_side_arm': (c_int, [c_int]),
    'dm_wgrad_side_touch': (c_int, []),
    'dm_wgrad_side_join': (c_int, [_P]),
    'dm_dream_rollout_marks': (c_int, [c_int, POINTER(c_int), POINTER(c_void_p)]),
    'dm_mlp_head_fwd_rows': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, POINTER(dm_mlp_params), _P, _P, _P,
                                     c_size_t, _P]),
}

_lib = None
DM_ABI_VERSION = 13     # include/dreamer_hip.h dm_version(): the struct layouts above (dm_rssm_params: 58 slots) belong to this one


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DreamerHipError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C pydreamer_amd/csrc`). There is no CPU fallback for the training path.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.dm_version() != DM_ABI_VERSION:
            raise DreamerHipError(f'{LIB_PATH} has ABI version {handle.dm_version()}, this binding is written for '
                                  f'{DM_ABI_VERSION}: rebuild it (`make -C pydreamer_amd/csrc`)')
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def call(name, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        msg = lib().dm_last_error().decode('utf-8', 'replace')
        raise DreamerHipError(f'{name} failed with code {rc}: {msg}')
    return rc


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (or NULL for None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DreamerHipError(f'expected a CUDA/HIP tensor, got device {t.device} (no CPU path)')
    if not t.is_contiguous():
        raise DreamerHipError(f'expected a contiguous tensor, got strides {t.stride()} for shape {tuple(t.shape)}')
    return c_void_p(t.data_ptr())


def fptr(t):
    if t is not None and t.dtype != torch.float32:
        raise DreamerHipError(f'expected float32, got {t.dtype}')
    return ptr(t)


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def make_shape(**kw):
    s = dm_shape()
    for k, v in kw.items():
        setattr(s, k, int(v))
    return s


def workspace_bytes(shape):
    return int(lib().dm_workspace_bytes(ctypes.byref(shape)))


def mlp_struct(tensors_w, tensors_b, tensors_g, tensors_be, cls=dm_mlp_params, precision=0):
    """Pack per-layer tensors into a dm_mlp_params / dm_mlp_grads struct (keeps no references)."""
    s = cls()
    if cls is dm_mlp_params:
        s.precision = int(precision)
    for i, t in enumerate(tensors_w):
        s.w[i] = t.data_ptr()
    for i, t in enumerate(tensors_b):
        s.b[i] = t.data_ptr()
    for i, t in enumerate(tensors_g):
        s.ln_g[i] = t.data_ptr()
    for i, t in enumerate(tensors_be):
        s.ln_b[i] = t.data_ptr()
    return s


def conv_struct(ws, bs, cls=dm_conv_params):
    s = cls()
    for i, t in enumerate(ws):
        s.w[i] = t.data_ptr()
    for i, t in enumerate(bs):
        s.b[i] = t.data_ptr()
    return s


def cu_masked_stream(words, device):
    """A torch stream confined to the CUs set in `words` (list of uint32, 32 CUs each); lives for the process."""
    import torch
    arr = (ctypes.c_uint32 * len(words))(*words)
    out = c_void_p()
    with torch.cuda.device(device):
        call('dm_stream_create_cu_mask', arr, len(words), ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value, device=device)


def rssm_struct(tensors, cls=dm_rssm_params):
    """tensors: one per slot (rssm_param_names order); trailing slots (the 6 LayerNorm-GRU ones, the 12 + 18 of the stack's
    layers 1..3) may be omitted, absent slots are None."""
    tensors = list(tensors) + [None] * (DM_RSSM_NPARAMS - len(tensors))
    assert len(tensors) == DM_RSSM_NPARAMS
    s = cls()
    for i, t in enumerate(tensors):
        s.p[i] = None if t is None else t.data_ptr()
    return s
