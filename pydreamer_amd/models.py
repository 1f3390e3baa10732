"""Dreamer / WorldModel / ActorCritic with pydreamer's module API, executing on hand-written HIP kernels.

Mirrors the reference module tree so `state_dict()` keys, shapes and optimizer grouping are identical
(reference: pydreamer/models/{dreamer,rssm,rnn,a2c,encoders,decoders,common,probes}.py), but no module here has a
torch `forward`: the nn.Modules only own `nn.Parameter`s.  `Dreamer.training_step()` packs raw device pointers and
calls the C-ABI of libdreamer_hip.so (pydreamer_amd/hip.py) through three `torch.autograd.Function`s — world
model, actor, critic — so each of the 4 returned losses supports an independent `.backward()` exactly like the
reference (train.py:184-187).  There is no CPU path: tensors must live on a gfx950 device.

Supported configuration (everything else raises NotImplementedError): iwae_samples>=1 (also with the logging flags), gru_type in {gru, gru_layernorm,
gru_layernorm_dv2}, gru_layers 1..4, stoch_discrete>0 or 0 (Gaussian latents, also with iwae_samples>1), layer_norm True or False,
aux_critic, image_encoder/decoder='cnn' at 64x64, actor_dist in {onehot, tanh_normal, normal_tanh}, actor_grad='reinforce',
probe_model='none', no vecobs / reward_input.
"""
import contextlib
import ctypes
import os
import math

import torch
import torch.nn as nn

from . import hip as H
from .optim import FusedAdamW

MLP_HIDDEN = 400          # a2c.py:16, decoders.py:259,289
ACTOR_KINDS = {'onehot': 0, 'tanh_normal': 1, 'normal_tanh': 2}   # dm_shape.flags bits 0-1
REWARD_STD = 0.3989422804  # decoders.py:289


# ---------------------------------------------------------------------------------------------------------------
# parameter holders (no forward): names/shapes follow torch.nn so state_dict keys equal the reference's
# ---------------------------------------------------------------------------------------------------------------
class _Params(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError('pydreamer_amd modules hold parameters only; compute runs in libdreamer_hip.so')


class LinearP(_Params):
    def __init__(self, in_dim, out_dim, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_dim, in_dim))
        self.bias = nn.Parameter(torch.empty(out_dim)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))           # torch.nn.Linear default (used by `ac`)
        if bias:
            bound = 1 / math.sqrt(in_dim)
            nn.init.uniform_(self.bias, -bound, bound)


class LayerNormP(_Params):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class ConvP(_Params):
    def __init__(self, shape, bias_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.zeros(bias_dim))
        nn.init.xavier_uniform_(self.weight)


class GRUCellP(_Params):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.weight_ih = nn.Parameter(torch.empty(3 * hidden_size, input_size))
        self.weight_hh = nn.Parameter(torch.empty(3 * hidden_size, hidden_size))
        self.bias_ih = nn.Parameter(torch.zeros(3 * hidden_size))
        self.bias_hh = nn.Parameter(torch.zeros(3 * hidden_size))


class _Slot(_Params):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the reference (LN/ELU/Flatten slots)."""


def init_weights_tf2(m):
    """functions.py:81-94, applied to the world model only (dreamer.py:283-284)."""
    if isinstance(m, (LinearP, ConvP)):
        nn.init.xavier_uniform_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)
    if isinstance(m, GRUCellP):
        nn.init.xavier_uniform_(m.weight_ih.data)
        nn.init.orthogonal_(m.weight_hh.data)
        nn.init.zeros_(m.bias_ih.data)
        nn.init.zeros_(m.bias_hh.data)


class MLP(_Params):
    """common.py:37-65: model = Sequential[Linear, LayerNorm | NoNorm, ELU]*L + [Linear] (+Flatten if out_dim==1).
    layer_norm=False: the norm slots are parameter-free (common.py:68-74) and the library gets null LayerNorm pointers."""

    def __init__(self, in_dim, out_dim, hidden_dim, hidden_layers, layer_norm=True):
        super().__init__()
        self.in_dim, self.out_dim, self.hidden_dim, self.hidden_layers = in_dim, out_dim, hidden_dim, hidden_layers
        self.layer_norm = bool(layer_norm)
        layers, dim = [], in_dim
        for _ in range(hidden_layers):
            layers += [LinearP(dim, hidden_dim), LayerNormP(hidden_dim) if layer_norm else _Slot(), _Slot()]
            dim = hidden_dim
        layers += [LinearP(dim, out_dim)]
        if out_dim == 1:
            layers += [_Slot()]
        self.model = nn.Sequential(*layers)

    # --- C-ABI marshalling
    def tensors(self):
        L = self.hidden_layers
        w = [self.model[3 * i].weight for i in range(L)] + [self.model[3 * L].weight]
        b = [self.model[3 * i].bias for i in range(L)] + [self.model[3 * L].bias]
        g = [self.model[3 * i + 1].weight for i in range(L)] if self.layer_norm else []
        be = [self.model[3 * i + 1].bias for i in range(L)] if self.layer_norm else []
        return w, b, g, be

    def param_list(self):
        """Parameters in nn.Module.parameters() order (the order of the flat gradient buffer)."""
        return list(self.parameters())

    def struct(self):
        # precision: 1 = this head's contractions take bf16 operands (set by Dreamer from conf.amp; per call, no global)
        return H.mlp_struct(*self.tensors(), precision=getattr(self, 'precision', 0))

    def grad_struct(self, grads_by_param):
        w, b, g, be = self.tensors()
        pick = lambda ts: [grads_by_param[id(t)] for t in ts]
        return H.mlp_struct(pick(w), pick(b), pick(g), pick(be), cls=H.dm_mlp_grads)

    def acts_floats(self, rows):
        return int(H.lib().dm_mlp_acts_floats(rows, self.hidden_dim, self.hidden_layers))

    def fwd(self, x2d, ldx, rows, ws, acts=None, save_acts=True, sparse_cols=0, window=None, out=None):
        """x2d: device tensor whose rows (leading dim ldx floats) hold in_dim features. Returns (out, acts).
        save_acts=False (heads nobody differentiates: critic_target, the dream's reward / terminal heads, inference):
        no activation buffer is allocated or written; the library ping-pongs through the workspace.
        sparse_cols: the last sparse_cols input columns are mostly zero (the one-hot latent part of a feature row); same
        result for any input, cheaper first layer on large batches (dm_mlp_head_fwd_sparse).
        window=(rows_total, row0): x2d, `out` and `acts` are the FULL arrays of rows_total rows (given by the caller) and
        only rows [row0, row0 + rows) are computed (dm_mlp_head_fwd_rows); a row's result does not depend on the window."""
        need = 4 * int(H.lib().dm_mlp_ws_floats(rows, self.hidden_dim, self.hidden_layers))
        if ws.numel() < need:
            raise H.DreamerHipError(f'MLP.fwd: workspace of {ws.numel()} bytes, need {need} for {rows} rows')
        if window is not None:
            rows_total, row0 = window
            assert out is not None and out.shape[0] == rows_total and (acts is not None or not save_acts)
            st = self.struct()
            H.call('dm_mlp_head_fwd_rows', rows_total, row0, rows, self.in_dim, sparse_cols if 0 < sparse_cols < self.in_dim else 0,
                   self.hidden_dim, self.hidden_layers, self.out_dim, H.fptr(x2d), ldx, ctypes.byref(st),
                   H.fptr(acts) if save_acts else None, H.fptr(out), H.ptr(ws), ws.numel(), H.stream())
            return out, acts
        if acts is None and save_acts:
            acts = torch.empty(self.acts_floats(rows), device=x2d.device)
        out = torch.empty(rows, self.out_dim, device=x2d.device)
        st = self.struct()
        if 0 < sparse_cols < self.in_dim:
            H.call('dm_mlp_head_fwd_sparse', rows, self.in_dim, sparse_cols, self.hidden_dim, self.hidden_layers, self.out_dim,
                   H.fptr(x2d), ldx, ctypes.byref(st), H.fptr(acts), H.fptr(out), H.ptr(ws), ws.numel(), H.stream())
        else:
            H.call('dm_mlp_head_fwd', rows, self.in_dim, self.hidden_dim, self.hidden_layers, self.out_dim, H.fptr(x2d), ldx,
                   ctypes.byref(st), H.fptr(acts), H.fptr(out), H.ptr(ws), ws.numel(), H.stream())
        return out, acts

    def bwd(self, x2d, ldx, rows, acts, dout, ws, dx=None, lddx=0, dx_accum=False, scratch=False):
        """Returns (gradients in parameters() order, the flat buffer they are views of, direct): `direct` means the views
        ARE the optimizer's `.grad` slots (freshly zeroed), so autograd has nothing left to accumulate."""
        plist = self.param_list()
        flat, out, direct = _flat_views(plist, x2d.device, getattr(self, '_fused', None), scratch)
        grads = {id(p): v for p, v in zip(plist, out)}
        st, gs = self.struct(), self.grad_struct(grads)
        H.call('dm_mlp_head_bwd', rows, self.in_dim, self.hidden_dim, self.hidden_layers, self.out_dim, H.fptr(x2d), ldx,
               ctypes.byref(st), H.fptr(acts), H.fptr(dout), ctypes.byref(gs), H.fptr(dx) if dx is not None else None, lddx,
               1 if dx_accum else 0, H.ptr(ws), ws.numel(), H.stream())
        return out, flat, direct


# ---------------------------------------------------------------------------------------------------------------
# world-model sub-modules (parameter trees only)
# ---------------------------------------------------------------------------------------------------------------
class ConvEncoder(_Params):
    """encoders.py:72-96."""

    def __init__(self, in_channels=3, cnn_depth=32):
        super().__init__()
        self.out_dim = cnn_depth * 32
        d = cnn_depth
        chans = [(in_channels, d), (d, 2 * d), (2 * d, 4 * d), (4 * d, 8 * d)]
        layers = []
        for ci, co in chans:
            layers += [ConvP((co, ci, 4, 4), co), _Slot()]
        layers += [_Slot()]
        self.model = nn.Sequential(*layers)

    def convs(self):
        return [self.model[i] for i in (0, 2, 4, 6)]


class MultiEncoder(_Params):
    """encoders.py:10-69 (image encoder only)."""

    def __init__(self, conf):
        super().__init__()
        if conf.reward_input or conf.vecobs_size or conf.image_encoder != 'cnn':
            raise NotImplementedError('only image_encoder=cnn without reward_input/vecobs is built in the HIP path')
        self.encoder_image = ConvEncoder(in_channels=conf.image_channels, cnn_depth=conf.cnn_depth)
        self.out_dim = self.encoder_image.out_dim


class ConvDecoder(_Params):
    """decoders.py:111-180 (mlp_layers=0)."""

    def __init__(self, in_dim, out_channels=3, cnn_depth=32):
        super().__init__()
        self.in_dim = in_dim
        d = cnn_depth
        self.model = nn.Sequential(
            LinearP(in_dim, d * 32), _Slot(),
            ConvP((d * 32, d * 4, 5, 5), d * 4), _Slot(),
            ConvP((d * 4, d * 2, 5, 5), d * 2), _Slot(),
            ConvP((d * 2, d, 6, 6), d), _Slot(),
            ConvP((d, out_channels, 6, 6), out_channels))

    def layers(self):
        return [self.model[i] for i in (0, 2, 4, 6, 8)]


class DenseNormalDecoder(_Params):
    """decoders.py:287-319."""

    def __init__(self, in_dim, hidden_layers, layer_norm=True):
        super().__init__()
        self.model = MLP(in_dim, 1, MLP_HIDDEN, hidden_layers, layer_norm)
        self.std = REWARD_STD


class DenseBernoulliDecoder(_Params):
    """decoders.py:257-284."""

    def __init__(self, in_dim, hidden_layers, layer_norm=True):
        super().__init__()
        self.model = MLP(in_dim, 1, MLP_HIDDEN, hidden_layers, layer_norm)


class MultiDecoder(_Params):
    """decoders.py:10-108."""

    def __init__(self, features_dim, conf):
        super().__init__()
        if conf.image_decoder != 'cnn' or conf.reward_decoder_categorical or conf.vecobs_size:
            raise NotImplementedError('only image_decoder=cnn with Normal reward decoder is built in the HIP path')
        self.image_weight, self.reward_weight, self.terminal_weight = conf.image_weight, conf.reward_weight, conf.terminal_weight
        self.image = ConvDecoder(in_dim=features_dim, out_channels=conf.image_channels, cnn_depth=conf.cnn_depth)
        self.reward = DenseNormalDecoder(features_dim, conf.reward_decoder_layers, conf.layer_norm)
        self.terminal = DenseBernoulliDecoder(features_dim, conf.terminal_decoder_layers, conf.layer_norm)


class NormGRUCellP(_Params):
    """rnn.py:95-104 (gru_layernorm): bias-free gate products, one LayerNorm(eps 1e-3) per gate."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.weight_ih = LinearP(input_size, 3 * hidden_size, bias=False)
        self.weight_hh = LinearP(hidden_size, 3 * hidden_size, bias=False)
        self.ln_reset = LayerNormP(hidden_size)
        self.ln_update = LayerNormP(hidden_size)
        self.ln_newval = LayerNormP(hidden_size)


class NormGRUCellLateResetP(_Params):
    """rnn.py:117-125 (gru_layernorm_dv2): one LayerNorm over all three gates, update bias -1, late reset."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.weight_ih = LinearP(input_size, 3 * hidden_size, bias=False)
        self.weight_hh = LinearP(hidden_size, 3 * hidden_size, bias=False)
        self.lnorm = LayerNormP(3 * hidden_size)


class GRUCellStack(_Params):
    """rnn.py:40-67; cell_type in {gru, gru_layernorm, gru_layernorm_dv2}, up to 4 layers of any of them: layer i owns
    columns [i*layer_size, (i+1)*layer_size) of the state."""

    def __init__(self, input_size, hidden_size, num_layers, cell_type):
        super().__init__()
        cells = dict(gru=GRUCellP, gru_layernorm=NormGRUCellP, gru_layernorm_dv2=NormGRUCellLateResetP)
        if cell_type not in cells or not 1 <= num_layers <= H.DM_MAX_GRU_LAYERS:
            raise NotImplementedError(f'gru_type={cell_type!r}, gru_layers={num_layers} not built in the HIP path')
        layer_size = hidden_size // num_layers
        assert layer_size * num_layers == hidden_size, 'Must be divisible'
        if layer_size % 4:
            raise NotImplementedError(f'deter_dim / gru_layers = {layer_size} must be a multiple of 4 in the HIP path')
        self.cell_type, self.num_layers = cell_type, num_layers
        self.layers = nn.ModuleList([cells[cell_type](input_size, layer_size)] +
                                    [cells[cell_type](layer_size, layer_size) for _ in range(num_layers - 1)])


class RSSMCell(_Params):
    """rssm.py:97-203."""

    def __init__(self, embed_dim, action_dim, deter_dim, stoch_dim, stoch_discrete, hidden_dim, gru_layers, gru_type, layer_norm):
        super().__init__()
        norm = LayerNormP if layer_norm else (lambda n: _Slot())       # NoNorm (common.py:68-74): no parameters
        self.stoch_dim, self.stoch_discrete, self.deter_dim = stoch_dim, stoch_discrete, deter_dim
        # discrete latents: stoch_dim one-hot groups of stoch_discrete; Gaussian latents (stoch_discrete = 0): z is stoch_dim
        # wide and the prior / posterior heads emit (mean | raw std)                                 (rssm.py:103,112,117)
        Z, ZP = stoch_dim * (stoch_discrete or 1), stoch_dim * (stoch_discrete or 2)
        self.z_mlp = LinearP(Z, hidden_dim)
        self.a_mlp = LinearP(action_dim, hidden_dim, bias=False)
        self.in_norm = norm(hidden_dim)
        self.gru = GRUCellStack(hidden_dim, deter_dim, gru_layers, gru_type)
        self.prior_mlp_h = LinearP(deter_dim, hidden_dim)
        self.prior_norm = norm(hidden_dim)
        self.prior_mlp = LinearP(hidden_dim, ZP)
        self.post_mlp_h = LinearP(deter_dim, hidden_dim)
        self.post_mlp_e = LinearP(embed_dim, hidden_dim, bias=False)
        self.post_norm = norm(hidden_dim)
        self.post_mlp = LinearP(hidden_dim, ZP)

    def ordered(self):
        """Tensors in the DM_RSSM_* order of include/dreamer_hip.h (None for slots this cell type does not have)."""
        named = dict(self.named_parameters())
        return [named.get(n) for n in H.rssm_param_names(self.gru.cell_type, self.gru.num_layers)]      # None: slot not present

    def init_state(self, batch_size):
        dev = self.z_mlp.weight.device
        return (torch.zeros((batch_size, self.deter_dim), device=dev),
                torch.zeros((batch_size, self.stoch_dim * (self.stoch_discrete or 1)), device=dev))


class RSSMCore(_Params):
    """rssm.py:15-93."""

    def __init__(self, embed_dim, action_dim, deter_dim, stoch_dim, stoch_discrete, hidden_dim, gru_layers, gru_type, layer_norm):
        super().__init__()
        self.cell = RSSMCell(embed_dim, action_dim, deter_dim, stoch_dim, stoch_discrete, hidden_dim, gru_layers, gru_type, layer_norm)

    def init_state(self, batch_size):
        return self.cell.init_state(batch_size)


class NoProbeHead(nn.Module):
    """probes.py:140-150: keeps the 4-loss / 4-optimizer structure."""

    def __init__(self):
        super().__init__()
        self.dummy = nn.Parameter(torch.zeros(1), requires_grad=True)

    def training_step(self, features, obs):
        return torch.square(self.dummy), {}, {}


def _torch_actor_distribution(actor_dist, y):
    """The distribution object Dreamer.inference hands to the acting process (a2c.py:43-55, functions.py:59-78); the
    parameters come from the HIP actor, the torch.distributions wrapper is only the return type of the reference API."""
    import torch.distributions as D
    import torch.nn.functional as F
    if actor_dist == 'onehot':
        return D.OneHotCategorical(logits=y)
    mean_, std_ = y.chunk(2, -1)
    if actor_dist == 'normal_tanh':
        return D.Independent(D.Normal(torch.tanh(mean_), torch.sigmoid(std_) + 0.01), 1)
    normal = D.Independent(D.Normal(5 * torch.tanh(mean_ / 5), F.softplus(std_) + 0.1), 1)
    dist = D.TransformedDistribution(normal, [D.TanhTransform()])
    dist.entropy = normal.entropy
    return dist


class _Mean:
    """Stand-in for the torch.distributions objects Dreamer.dream returns: only `.mean` is consumed (dreamer.py:155-156)."""

    def __init__(self, mean):
        self.mean = mean


# ---------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------
def _flat_views(plist, device, fused=None, scratch=False):
    """Gradient buffers for one backward pass: (flat, per-parameter views, direct).
    With a FusedAdamW attached whose gradient buffer was just zeroed (the trainer's zero_grad -> backward order,
    train.py:186-192) the views are the optimizer's own `.grad` slots: the kernels write the gradients in place and the
    ~120 per-parameter AccumulateGrad additions of autograd disappear.  Otherwise (no fused optimizer, or a second
    backward without zero_grad = gradient accumulation) a scratch buffer is returned and autograd accumulates as usual.
    scratch=True (backward passes pre-launched inside training_step, i.e. BEFORE the trainer's zero_grad): the optimizer's
    persistent scratch buffer in the same padded layout; _finish_backward() moves it into `.grad` with one kernel."""
    if fused is not None and scratch:      # scratch: True, or the generation training_step() claimed on the caller's thread
        views = fused.scratch_views(plist, bump=scratch is True)
        if views is not None:
            return fused.scratch, views, ('scratch', fused.scratch_gen if scratch is True else int(scratch))
    if fused is not None and not scratch:
        views = fused.claim_fresh_grads(plist)
        if views is not None:
            return fused.flat_grad, views, True
    flat = torch.empty(sum(p.numel() for p in plist), device=device)
    views, off = [], 0
    for p in plist:
        views.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return flat, views, False


def _multi_sum(items, device, out=None):
    """items: list of (tensor, scale) or (tensor, scale, center_tensor[, mode]) -> 1-D tensor of results (written into `out`,
    a slice of the step's metric buffer, when given)."""
    arr = (H.dm_reduce_item * len(items))()
    for i, it in enumerate(items):
        arr[i].x = it[0].data_ptr()
        arr[i].n = it[0].numel()
        arr[i].scale = it[1]
        if len(it) > 2:
            arr[i].mode = it[3] if len(it) > 3 else 1
            arr[i].center = it[2].data_ptr()
        else:
            arr[i].mode = 0
            arr[i].center = None
    if out is None:
        out = torch.empty(len(items), device=device)
    assert out.numel() == len(items) and out.is_contiguous()
    H.call('dm_multi_sum', len(items), arr, H.fptr(out), H.stream())
    return out


# One device buffer per step for every loss / metric scalar: slot layout
METRIC_SLOTS = dict(loss_kl=0, loss_image=1, loss_reward=2, loss_terminal=3, entropy_prior=4, entropy_post=5, loss_model=6,
                    loss_critic=8, loss_actor=9, policy_entropy=10, policy_value=11, policy_value_im=12, policy_reward=13,
                    policy_reward_std=14, grad_norm=16, grad_norm_probe=18, grad_norm_actor=20, grad_norm_critic=22,
                    loss_critic_aux=24, policy_value_aux=25)      # slot 7: loss_model + aux_critic_weight * loss_critic_aux
METRIC_BUF_FLOATS = 28


def _finish_backward(owner, grads, flat, direct, grad_loss):
    """Chain rule with the incoming scalar gradient (1.0 unless a GradScaler is active) without a host sync, and hand-over
    of the gradients: returns the tuple for autograd (None when they already sit in the optimizer's `.grad` slots)."""
    gl = grad_loss.detach().float().reshape(1).contiguous()
    fused = getattr(owner, '_fused', None)
    if fused is not None and flat is fused.scratch:
        if not isinstance(direct, tuple) or direct[1] != fused.scratch_gen:
            raise RuntimeError('the gradients pre-computed by this training_step() were overwritten by a later training_step() '
                               'before backward() was called; call backward() after each training_step(), or set '
                               'model.overlap_backward = False to compute gradients inside backward()')
        fused.order_after_current()          # (pipelined mode: `gl` was made on this stream, the hand-over runs on the group's)
        if fused.home is not None:
            gl.record_stream(fused.home)
        fused.adopt_scratch(gl)              # buffer swap ('=' semantics) or accumulation; see FusedAdamW.adopt_scratch
        return tuple(None for _ in grads)
    H.call('dm_scale_inplace', H.fptr(flat), flat.numel(), H.fptr(gl), H.stream())
    return tuple(None for _ in grads) if direct else tuple(grads)


def _prelaunched(owner, fn):
    """Runs a pre-launched backward pass (on its side stream, from the launcher thread) and, under data parallelism,
    starts the all-reduce of its gradient buffer right behind it on the same stream."""
    out = fn()
    fused = getattr(owner, '_fused', None)
    if fused is not None and fused.dp is not None and out[1] is fused.scratch:
        from . import dist as D
        D.allreduce_scratch_async(fused)
    return out


_HEADS_EARLY = os.environ.get('DM_HEADS_EARLY', '1') != '0'    # A/B switch: 0 runs the heads over the imagined states behind the rollout only
_WGRAD_SIDE_DP = os.environ.get('DM_WGRAD_SIDE_DP', '1') != '0'    # A/B switch: 0 = no side stream under data parallelism (rounds 3-5)
_WGRAD_SIDE = os.environ.get('DM_WGRAD_SIDE', '1') != '0'      # A/B switch: 0 keeps every weight gradient on the caller's stream


def _warn_if_communicator_exists():
    """The step's streams are about to be created: has torch's default process group built its RCCL communicator already?"""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != 'nccl':
            return
        be = dist.distributed_c10d._get_default_group()._get_backend(torch.device('cuda'))
        if be._is_initialized():
            import warnings
            warnings.warn('pydreamer_amd: an RCCL communicator exists before the model\'s streams were created - every training step '
                          'will be slower by 10-20 ms (hardware-queue assignment order; profiles/r06_force_dp.txt).  Build the model and '
                          'call init_optimizers() / prepare_streams() before the first collective.', RuntimeWarning, stacklevel=3)
    except Exception:       # a diagnostic only: private torch API
        pass


class _Overlap:
    """Side streams for the backward passes.  Everything the three backward passes consume is fixed once the matching
    forward has run (dreamer.py:149-157 detaches the features the actor-critic trains on), so training_step() launches
      * the world-model backward (decoder / heads, the strictly sequential BPTT chain of T x ~9 small kernels that
        leaves most of the 256 CUs idle, encoder) on `s_wm` right after the world-model forward, concurrently with the
        imagination rollout and the actor-critic forward on the caller's stream, and
      * the actor / critic backward on `s_ac` right after their forward.
    They write into the optimizers' scratch buffers (the trainer calls zero_grad() between training_step() and
    backward(), train.py:186-192); loss.backward() joins the stream and moves the result into `.grad`, so callers see
    ordinary stream semantics.  Each stream has its own workspace."""

    def __init__(self, device):
        # the latency-bound chain gets the high-priority queue: its 40-110-workgroup kernels must be dispatched ahead of
        # the thousands of queued GEMM workgroups of the concurrent work, or the chain just slows down
        self.s_wm = torch.cuda.Stream(device, priority=int(os.environ.get('DM_WM_PRIO', '-1')))
        # DM_AC_RESERVE_CUS=k (experiment): the actor-critic stream is created through
        # hipExtStreamCreateWithCUMask WITHOUT the first k CUs of every 32, so the world-model stream's latency chain always finds
        # k CUs per XCD free of this stream's full-chip products - no additional stream, s_wm stays unmasked
        res = int(os.environ.get('DM_AC_RESERVE_CUS', '0'))
        if res > 0:
            self.s_ac = H.cu_masked_stream([0xFFFFFFFF ^ ((1 << res) - 1)] * 8, device)
        else:
            self.s_ac = torch.cuda.Stream(device, priority=int(os.environ.get('DM_AC_PRIO', '0')))
        self.ev_wm_fwd = torch.cuda.Event()
        self.ev_fwd = torch.cuda.Event()
        self.ev_fork, self.ev_tail = torch.cuda.Event(), torch.cuda.Event()      # the world-model forward's tail on s_wm (WorldModel._forward)
        # the rollout's progress mark (recorded by the library, dm_dream_rollout_marks) and the early head window's completion
        self.ev_mark, self.ev_heads = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.device(device):
            self.ev_mark.record()          # (torch creates the HIP event at the first record; the library needs its handle)
            self.ev_heads.record()
        # Every stream of the step gets one command NOW (round 6): a HIP stream takes its hardware queue when it first gets work, and
        # which streams later SHARE a queue depends on that order - a communicator created before these streams had run anything
        # cost the step +13 ms, an early all-reduce whose communicator appeared mid-step +6 ms (profiles/r06_force_dp.txt).
        if os.environ.get('DM_STREAM_PREBIND', '1') != '0':
            with torch.cuda.device(device):
                for st in (self.s_wm, self.s_ac):
                    with torch.cuda.stream(st):
                        torch.zeros(64, device=device)
                if _WGRAD_SIDE:
                    try:
                        H.call('dm_wgrad_side_touch')
                    except H.DreamerHipError:       # an optimisation only (the library owns ONE side stream, on the device that used it first)
                        pass
                torch.cuda.synchronize(device)
        self.ws_wm = None
        self.ws_ac = None
        # The HIP runtime needs ~6 us of host time per kernel launch and a step is ~1800 launches; at small per-GPU
        # batches (data parallel: B/8 rows) the step is bound by exactly that.  The pre-launched backward passes are
        # therefore ENQUEUED by a second host thread (ctypes drops the GIL inside the C calls), in parallel with the
        # caller's thread enqueuing the imagination rollout.
        self.device = device
        self.pool = None

    def submit(self, stream, wait_event, fn):
        """Run fn() on `stream` (after `wait_event`) from the launcher thread; returns a Future."""
        if torch.cuda.is_current_stream_capturing():      # graph capture is bound to the capturing thread
            stream.wait_event(wait_event)
            with torch.cuda.stream(stream):
                return _Done(fn())
        if self.pool is None:
            import concurrent.futures
            self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix='dm-launch')

        def job():
            with torch.cuda.device(self.device), torch.no_grad():
                stream.wait_event(wait_event)
                with torch.cuda.stream(stream):
                    return fn()
        fut = self.pool.submit(job)
        from . import dist as D
        D.track(fut)              # a main-thread collective must not overtake the collectives this job may issue
        return fut


def pack_metrics(*dicts):
    """the trainer reads ~20 scalar metrics per logged step with one `.item()` (= one device sync) each
    (train.py:204-214).  All metrics of this package are 0-d device tensors produced without a sync; this packs any
    number of metric dicts into ONE 1-D device tensor so a single `.cpu()` / non-blocking copy replaces the syncs:
        names, packed = pack_metrics(loss_metrics, grad_metrics);  values = dict(zip(names, packed.tolist()))"""
    names, vals = [], []
    for d in dicts:
        for k, v in d.items():
            names.append(k)
            vals.append(v.detach().reshape(()).float())
    return names, (torch.stack(vals) if vals else torch.empty(0))


class _Thunk:
    __slots__ = ('fn',)

    def __init__(self, fn):
        self.fn = fn


class LazyTensors(dict):
    """The `tensors` dict of training_step() (train.py:218-221 logs it only when `will_log_batch`): values may be thunks
    that are computed - and cached - the first time somebody reads them.  `image_rec` (123 MB at Atari-literal,
    decoders.py:177) is the one that matters: it is no longer written on the steps that never look at it."""

    def _resolve(self, k, v):
        if isinstance(v, _Thunk):
            v = v.fn()
            dict.__setitem__(self, k, v)
        return v

    def lazy(self, k, fn):
        dict.__setitem__(self, k, _Thunk(fn))

    def __getitem__(self, k):
        return self._resolve(k, dict.__getitem__(self, k))

    def __iter__(self):                     # overriding __iter__ also keeps dict(x) / {**x} off CPython's raw-copy fast path
        return iter(list(dict.keys(self)))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]

    def pop(self, k, *default):
        if k in self:
            v = self[k]
            dict.__delitem__(self, k)
            return v
        if default:
            return default[0]
        raise KeyError(k)

    def copy(self):
        new = LazyTensors()
        for k in dict.keys(self):
            dict.__setitem__(new, k, dict.__getitem__(self, k))      # thunks stay thunks
        return new

    def is_materialised(self, k):
        return not isinstance(dict.__getitem__(self, k), _Thunk)


class _Done:
    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


class _StepArena:
    """Allocator of everything the library's launch chains read or write in one step (dm_rssm_sequence_fwd / _bwd,
    dm_dream_rollout).  Rounds 3-4 kept these buffers persistent per geometry for the hipGraph replay of the chains
    (csrc/chain_graph.hip), which removed host time only and was deleted in round 5 (the chains are GPU-latency-bound); what is
    left is `get` = torch.empty - nothing is shared between steps, a backward() on an older step's losses is always valid -
    behind the same interface, so the call sites did not move."""

    on = False

    def begin_step(self, geometry):
        return None

    def current(self, stamp):
        return True

    def get(self, name, shape, dtype=torch.float32, device=None):
        return torch.empty(tuple(int(x) for x in shape), dtype=dtype, device=device)


def _require_cuda(t, what):
    if not t.is_cuda:
        raise H.DreamerHipError(f'{what} is on {t.device}: pydreamer_amd has no CPU path (the HIP library is the product)')


# ---------------------------------------------------------------------------------------------------------------
# world model
# ---------------------------------------------------------------------------------------------------------------
class _WMStep(torch.autograd.Function):
    """loss_model = WorldModel forward; backward = hand-written BPTT + conv backward through the C-ABI."""

    @staticmethod
    def forward(ctx, wm, pack, *params):
        ctx.wm, ctx.pack = wm, pack
        tail = pack.get('tail')
        if tail is not None:                  # the loss was written on the world-model stream (WorldModel._forward)
            with torch.cuda.stream(tail.s_wm):
                return pack['loss'].clone()
        return pack['loss'].clone()

    @staticmethod
    def backward(ctx, grad_loss):
        wm, pk = ctx.wm, ctx.pack
        if pk.get('consumed'):
            raise RuntimeError('loss_model.backward() called twice (saved activations were released)')
        ov = pk.get('overlap')
        if 'pre' in pk:                                   # launched on s_wm inside training_step()
            grads, flat, direct = pk.pop('pre').result()
            torch.cuda.current_stream().wait_stream(ov.s_wm)
        else:
            grads, flat, direct = wm._backward(pk, pk['ws'])
        for k in ('enc_acts', 'rssm_acts', 'dec_acts', 'r_acts', 't_acts', 'aux'):
            pk.pop(k, None)                               # only now: the side stream may have been reading them
        pk['consumed'] = True
        return (None, None) + _finish_backward(wm, grads, flat, direct, grad_loss)


class WorldModel(_Params):
    """dreamer.py:228-396."""

    def __init__(self, conf):
        super().__init__()
        if conf.aux_critic and conf.iwae_samples != 1:
            raise NotImplementedError('aux_critic assumes iwae_samples = 1 (as the reference does, dreamer.py:349)')
        if conf.image_size != 64:
            raise NotImplementedError('conv geometry is built for 64x64 observations')
        self.conf = conf
        self.deter_dim, self.stoch_dim, self.stoch_discrete = conf.deter_dim, conf.stoch_dim, conf.stoch_discrete
        self.kl_weight = conf.kl_weight
        self.kl_balance = None if conf.kl_balance == 0.5 else conf.kl_balance     # dreamer.py:241
        self.aux_critic_weight = conf.aux_critic_weight
        self.encoder = MultiEncoder(conf)
        features_dim = conf.deter_dim + conf.stoch_dim * (conf.stoch_discrete or 1)
        self.features_dim = features_dim
        self.decoder = MultiDecoder(features_dim, conf)
        self.core = RSSMCore(embed_dim=self.encoder.out_dim, action_dim=conf.action_dim, deter_dim=conf.deter_dim,
                             stoch_dim=conf.stoch_dim, stoch_discrete=conf.stoch_discrete, hidden_dim=conf.hidden_dim,
                             gru_layers=conf.gru_layers, gru_type=conf.gru_type, layer_norm=conf.layer_norm)
        # Auxiliary critic on the real trajectory (dreamer.py:267-279): a full ActorCritic lives inside the world model (its
        # parameters belong to the world-model optimizer and get the tf2 init); only its critic is trained (dreamer.py:347-358)
        self.ac_aux = None
        if conf.aux_critic:
            self.ac_aux = ActorCritic(in_dim=features_dim, out_actions=conf.action_dim, layer_norm=conf.layer_norm,
                                      gamma=conf.gamma_aux, lambda_gae=conf.lambda_gae_aux, entropy_weight=conf.entropy,
                                      target_interval=conf.target_interval_aux, actor_grad=conf.actor_grad,
                                      actor_dist=conf.actor_dist)
        for m in self.modules():
            init_weights_tf2(m)
        self._ws = None
        self._pipe = None
        self._arena = _StepArena()
        self._ws_dec = None                # own workspaces of the decoder / encoder backward while weight gradients are deferred
        self._ws_enc = None
        # Forward time-chunk pipeline over 3 streams (encoder chunk i+1 | posterior steps of chunk i | decoder chunk i-1).
        # OFF by default: measured on MI355X / ROCm 7.2 it LOSES (61.9 vs 47.7 ms per step at B=50, 33.1 vs 16.5 ms at
        # B=7): the loop's 1024-thread workgroups starve behind the conv GEMMs of the other streams and every cross-stream
        # dependency costs a cache writeback/invalidate between queues.  Kept because it is exact (tested) and cheap to flip.
        self.pipeline_chunks = int(os.environ.get('DM_PIPELINE_CHUNKS', '1'))

    def init_state(self, batch_size):
        return self.core.init_state(batch_size)

    # ---- shape / workspace
    def shape(self, T, B, H_):
        c = self.conf
        return H.make_shape(T=T, B=B, I=1, H=H_, D=c.deter_dim, Hd=c.hidden_dim, S=c.stoch_dim, C=c.stoch_discrete,
                            E=self.encoder.out_dim, A=c.action_dim, mlp_hidden=MLP_HIDDEN, mlp_layers=4,
                            cnn_depth=c.cnn_depth, img=c.image_size, img_ch=c.image_channels,
                            flags=ACTOR_KINDS.get(c.actor_dist, 0) | (H.GRU_KINDS[c.gru_type] << H.DM_FLAG_GRU_SHIFT) |
                            ((c.gru_layers - 1) << H.DM_FLAG_GRU_LAYERS_SHIFT) |
                            (H.DM_FLAG_BF16 if getattr(c, 'amp', False) else 0))

    def workspace(self, shp, device):
        need = H.workspace_bytes(shp)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def _pipeline(self, shp, T, B, chunks, device):
        """Streams, events and per-stream workspaces of the forward time-chunk pipeline (cached per geometry)."""
        key = (T, B, chunks, str(device))
        pp = self._pipe
        if pp is None or pp['key'] != key:
            step = -(-T // chunks)
            ranges = [(t, min(t + step, T)) for t in range(0, T, step)]
            sub = H.dm_shape.from_buffer_copy(shp)
            sub.T = step                                   # per-range decoder workspace: patch matrices scale with rows
            # DM_PIPE_RESERVE_CUS=r (0..31): the chain stream owns r CUs of every 32 and the convolution streams the other
            # 32-r (hipExtStreamCreateWithCUMask), so the chain's small kernels never wait for a CU to drain
            r = int(os.environ.get('DM_PIPE_RESERVE_CUS', '0'))
            if 0 < r < 32:
                lo, n_words = (1 << r) - 1, 8
                s_chain = H.cu_masked_stream([lo] * n_words, device)
                s_dec = H.cu_masked_stream([0xFFFFFFFF ^ lo] * n_words, device)
                s_enc = H.cu_masked_stream([0xFFFFFFFF ^ lo] * n_words, device)
            else:
                s_chain, s_dec, s_enc = torch.cuda.Stream(device, priority=-1), torch.cuda.Stream(device), None
            pp = dict(key=key, ranges=ranges, s_chain=s_chain, s_dec=s_dec, s_enc=s_enc,
                      ev_prep=torch.cuda.Event(), ev_enc=[torch.cuda.Event() for _ in ranges],
                      ev_chain=[torch.cuda.Event() for _ in ranges],
                      ws_chain=torch.empty((int(H.DM_SPLITK_FLOATS) + 4096) * 4, dtype=torch.uint8, device=device),
                      ws_dec=torch.empty(H.workspace_bytes(sub), dtype=torch.uint8, device=device))
            self._pipe = pp
        return pp

    def forward(self, obs, in_state, u_post=None):
        """dreamer.py:289-295: features and out_state only (used by Dreamer.inference)."""
        with torch.no_grad():
            pk = self._forward(obs, in_state, u_post, None, forward_only=True)
        T, B = obs['action'].shape[:2]
        return (pk['feat'].clone() if self._arena.on else pk['feat']).view(T, B, 1, -1), pk['out_state']

    # ---- forward through the C-ABI
    def _forward(self, obs, in_state, u_post, forced_idx, forward_only=False, imag_horizon=1, open_loop=False, mbuf=None,
                 iwae=1, tail=None):
        """iwae = I > 1 (rssm.py:35-41): every row-wise stage runs on T*B*I rows, row n = (t*B + b)*I + i; only the conv
        encoder sees the T*B frames once.  Three geometry structs: `shp` (T,B,I: conv decoder + workspace), `shp_e`
        (T,B,1: encoder), `shp_r` (T, B*I, 1: the RSSM calls take the expanded batch as their batch)."""
        c = self.conf
        image, action = obs['image'], obs['action']
        _require_cuda(image, "obs['image']")
        T, B = action.shape[:2]
        I = int(iwae)
        BI, NE = B * I, T * B
        N, dev = T * B * I, image.device
        D_, F_, E = c.deter_dim, self.features_dim, self.encoder.out_dim
        gauss = not c.stoch_discrete         # Gaussian latents: z is stoch_dim wide, its parameters (mean | raw std) twice that
        Z, ZP = c.stoch_dim * (c.stoch_discrete or 1), c.stoch_dim * (c.stoch_discrete or 2)
        if gauss and forced_idx is not None:
            raise NotImplementedError('Gaussian latents (stoch_discrete=0) have no indices to force')
        shp = self.shape(T, B, imag_horizon)
        shp.I = I
        ws = self.workspace(shp, dev)
        u8 = image.dtype == torch.uint8
        if u8:
            # the replay's native frames (T,B,H,W,C) uint8 are consumed AS THEY ARE: x/255-0.5 and HWC->CHW
            # (preprocessing.py:21-29) happen inside the first conv's patch loader and inside the MSE kernel
            # (dm_shape.flags bit DM_FLAG_IMAGE_U8); no float image is ever written
            if image.dim() != 5 or tuple(image.shape[2:]) != (c.image_size, c.image_size, c.image_channels):
                raise ValueError(f'uint8 image must be (T,B,{c.image_size},{c.image_size},{c.image_channels}), got {tuple(image.shape)}')
            image = image.contiguous()
            shp.flags |= H.DM_FLAG_IMAGE_U8
        else:
            image = image.float().contiguous()
        action = action.float().contiguous()
        reset = obs['reset'].to(torch.uint8).contiguous()
        h0, z0 = (x.float().contiguous() for x in in_state)
        # the C side sizes every access from dm_shape alone: check what the caller handed over before packing pointers
        want = dict(image=(T, B, c.image_size, c.image_size, c.image_channels) if u8 else
                    (T, B, c.image_channels, c.image_size, c.image_size), action=(T, B, c.action_dim), reset=(T, B))
        got = dict(image=tuple(image.shape), action=tuple(action.shape), reset=tuple(reset.shape))
        for k in ('reward', 'terminal'):
            if not forward_only or k in obs:
                if k in obs:
                    want[k], got[k] = (T, B), tuple(obs[k].shape)
        want['in_state[0]'], got['in_state[0]'] = (BI, D_), tuple(h0.shape)
        want['in_state[1]'], got['in_state[1]'] = (BI, Z), tuple(z0.shape)
        if u_post is not None:
            want['u_post numel'], got['u_post numel'] = T * BI * c.stoch_dim, u_post.numel()
        if forced_idx is not None:
            want['forced_idx'], got['forced_idx'] = (T, BI, c.stoch_dim), tuple(forced_idx.shape)
        bad = {k: (got[k], want[k]) for k in want if got[k] != want[k]}
        if bad:
            raise ValueError('training_step input shapes (got, expected): ' + ', '.join(f'{k}: {v[0]} != {v[1]}' for k, v in bad.items()))
        # Everything the posterior chain touches lives in the step arena (stable addresses -> the chain's hipGraph is replayed)
        ar = self._arena
        gen = ar.begin_step((T, B, I))
        if ar.on:
            u_buf = None
            if forced_idx is None or u_post is not None:
                # uniforms for the categorical inverse-CDF rule; standard-normal eps of Normal.rsample for Gaussian latents
                u_buf = ar.get('u_post', (T, BI, c.stoch_dim), device=dev)
                if u_post is not None:
                    u_buf.copy_(u_post.reshape(T, BI, c.stoch_dim))
                elif gauss:
                    u_buf.normal_()
                else:
                    u_buf.uniform_()
            u_post = u_buf
            h0 = ar.get('h0', (BI, D_), device=dev).copy_(h0)
            z0 = ar.get('z0', (BI, Z), device=dev).copy_(z0)
        elif u_post is not None:
            u_post = u_post.reshape(T, BI, c.stoch_dim).float().contiguous()
        elif forced_idx is None:
            u_post = (torch.randn if gauss else torch.rand)(T, BI, c.stoch_dim, device=dev)
        lib = H.lib()
        shp_e = shp_r = shp
        if I > 1:
            shp_e = H.dm_shape.from_buffer_copy(shp)
            shp_e.I = 1
            shp_r = H.dm_shape.from_buffer_copy(shp)
            shp_r.B, shp_r.I = BI, 1

        enc = self.encoder.encoder_image
        enc_p = H.conv_struct([m.weight for m in enc.convs()], [m.bias for m in enc.convs()])
        enc_acts = torch.empty(int(lib.dm_conv_encoder_acts_floats(ctypes.byref(shp_e))), device=dev)
        embed = ar.get('embed', (NE, E), device=dev)
        cell = self.core.cell
        rssm_p = H.rssm_struct(cell.ordered())
        if open_loop:
            # rssm.py:50-53: every step is RSSMCell.forward_prior - the same trunk, then the PRIOR head on h' with no
            # embedding term.  That is dm_rssm_sequence_fwd with the posterior head's parameter slots pointing at the
            # prior head and a zero embedding (its product with post_mlp_e is then an exact zero).  Forward only.
            po = list(cell.ordered())
            ix = {n: i for i, n in enumerate(H.RSSM_PARAM_ORDER)}
            for dst, src in (('post_mlp_h.weight', 'prior_mlp_h.weight'), ('post_mlp_h.bias', 'prior_mlp_h.bias'),
                             ('post_norm.weight', 'prior_norm.weight'), ('post_norm.bias', 'prior_norm.bias'),
                             ('post_mlp.weight', 'prior_mlp.weight'), ('post_mlp.bias', 'prior_mlp.bias')):
                po[ix[dst]] = po[ix[src]]
            rssm_p = H.rssm_struct(po)
        rssm_acts = ar.get('rssm_acts', (int(lib.dm_rssm_acts_floats(ctypes.byref(shp_r))),), device=dev)
        feat = ar.get('feat', (N, F_), device=dev)
        post = ar.get('post', (N, ZP), device=dev)
        prior = ar.get('prior', (N, ZP), device=dev)
        idx = ar.get('idx', (N, c.stoch_dim), torch.int32, device=dev)
        fidx = None
        if forced_idx is not None:
            fidx = (ar.get('forced_idx', (T, BI, c.stoch_dim), torch.int32, device=dev).copy_(forced_idx) if ar.on
                    else forced_idx.to(torch.int32).contiguous())
        u_ptr = H.fptr(u_post) if u_post is not None else None
        dec = self.decoder
        dl = dec.image.layers()
        dec_p = H.conv_struct([m.weight for m in dl], [m.bias for m in dl])
        if not forward_only:
            dec_acts = torch.empty(int(lib.dm_conv_decoder_acts_floats(ctypes.byref(shp))), device=dev)
            loss_image = torch.empty(N, device=dev)
            image_rec = None          # materialised lazily from the decoder's saved prediction (see LazyTensors)

        chunks = min(self.pipeline_chunks, T) if (T >= 4 and not forward_only and not open_loop and I == 1) else 1
        # rssm.py:35-41: (T,B,X) -> (T,B*I,X), pure data movement - into the arena (I = 1: a plain copy of the caller's tensors)
        A_ = c.action_dim
        if ar.on or I > 1:
            action_x = ar.get('action_x', (T, BI, A_), device=dev)
            action_x.view(T, B, I, A_).copy_(action.view(T, B, 1, A_).expand(T, B, I, A_))
            reset_x = ar.get('reset_x', (T, BI), torch.uint8, device=dev)
            reset_x.view(T, B, I).copy_(reset.view(T, B, 1).expand(T, B, I))
        else:
            action_x, reset_x = action, reset
        embed_x = embed
        if chunks <= 1:
            H.call('dm_conv_encoder_fwd', ctypes.byref(shp_e), H.ptr(image), ctypes.byref(enc_p), H.fptr(enc_acts),
                   H.fptr(embed), H.ptr(ws), ws.numel(), H.stream())
            if I > 1:
                embed_x = ar.get('embed_x', (N, E), device=dev)
                embed_x.view(T, B, I, E).copy_(embed.view(T, B, 1, E).expand(T, B, I, E))
            embed_rssm = ar.get('embed_zero', tuple(embed_x.shape), device=dev).zero_() if open_loop else embed_x
            H.call('dm_rssm_sequence_fwd', ctypes.byref(shp_r), H.fptr(embed_rssm), H.fptr(action_x), H.ptr(reset_x), H.fptr(h0),
                   H.fptr(z0), u_ptr, H.ptr(fidx), ctypes.byref(rssm_p), H.fptr(rssm_acts), H.fptr(feat), H.fptr(post),
                   H.fptr(prior), H.ptr(idx), H.ptr(ws), ws.numel(), H.stream())
        else:
            # Time-chunk pipeline over three streams: the posterior loop is a latency chain of T x ~10 small kernels
            # (rssm.py:38-58) that leaves most CUs idle, so the encoder of chunk i+1 and the decoder of chunk i-1 run
            # beside the loop steps of chunk i.  Same kernels on the same rows as the single-stream path.
            pp = self._pipeline(shp, T, B, chunks, dev)
            main = torch.cuda.current_stream()
            H.call('dm_conv_encoder_fwd_rows', ctypes.byref(shp), 0, 0, 1, H.ptr(image), ctypes.byref(enc_p),
                   H.fptr(enc_acts), H.fptr(embed), H.ptr(ws), ws.numel(), H.stream())
            H.call('dm_conv_decoder_mse_fwd_rows', ctypes.byref(shp), 0, 0, 1, H.fptr(feat), F_, H.ptr(image),
                   ctypes.byref(dec_p), H.fptr(dec_acts), H.fptr(loss_image), None, H.ptr(ws), ws.numel(),
                   H.stream())
            pp['ev_prep'].record(main)
            pp['s_chain'].wait_event(pp['ev_prep'])
            pp['s_dec'].wait_event(pp['ev_prep'])
            s_enc = pp['s_enc'] or main
            s_enc.wait_event(pp['ev_prep'])
            for i, (t0, t1) in enumerate(pp['ranges']):
                with torch.cuda.stream(s_enc):
                    H.call('dm_conv_encoder_fwd_rows', ctypes.byref(shp), t0 * B, (t1 - t0) * B, 0, H.ptr(image),
                           ctypes.byref(enc_p), H.fptr(enc_acts), H.fptr(embed), H.ptr(ws), ws.numel(), H.stream())
                    pp['ev_enc'][i].record(s_enc)
                pp['s_chain'].wait_event(pp['ev_enc'][i])
                with torch.cuda.stream(pp['s_chain']):
                    H.call('dm_rssm_sequence_fwd_steps', ctypes.byref(shp), t0, t1, H.fptr(embed), H.fptr(action_x),
                           H.ptr(reset_x), H.fptr(h0), H.fptr(z0), u_ptr, H.ptr(fidx), ctypes.byref(rssm_p),
                           H.fptr(rssm_acts), H.fptr(feat), H.fptr(post), H.fptr(prior), H.ptr(idx), H.ptr(pp['ws_chain']),
                           pp['ws_chain'].numel(), H.stream())
                    pp['ev_chain'][i].record(pp['s_chain'])
                pp['s_dec'].wait_event(pp['ev_chain'][i])
                with torch.cuda.stream(pp['s_dec']):
                    H.call('dm_conv_decoder_mse_fwd_rows', ctypes.byref(shp), t0 * B, (t1 - t0) * B, 0, H.fptr(feat), F_,
                           H.ptr(image), ctypes.byref(dec_p), H.fptr(dec_acts), H.fptr(loss_image), None,
                           H.ptr(pp['ws_dec']), pp['ws_dec'].numel(), H.stream())
            main.wait_stream(pp['s_chain'])
            main.wait_stream(pp['s_dec'])
            if pp['s_enc'] is not None:
                main.wait_stream(pp['s_enc'])

        last = feat[(T - 1) * BI:]
        out_state = (last[:, :D_].clone(), last[:, D_:].clone())                  # detached by construction (rssm.py:77)
        pk = dict(shp=shp, shp_e=shp_e, shp_r=shp_r, T=T, B=B, I=I, feat=feat, post=post, prior=prior, idx=idx,
                  out_state=out_state, embed=embed, embed_x=embed_x, action_x=action_x, reset_x=reset_x, gen=gen)
        if forward_only:
            return pk
        # The tail of the world-model forward - decoder + MSE, reward / terminal heads, KL, the loss sums (dreamer.py:311-365) -
        # is consumed by the world-model BACKWARD only: the imagination rollout (dreamer.py:149-157) needs the posterior
        # features and nothing else.  With `tail` (the training step's side streams, Dreamer.training_step) it is enqueued on
        # the world-model stream, in front of the pre-launched backward, with that stream's workspace, and the caller's stream
        # goes from the posterior loop straight to the rollout.  Same kernels on the same operands: bit-identical
        # (test_world_model_tail_on_side_stream_is_bit_identical).
        use_tail = tail is not None and chunks <= 1
        if use_tail:
            need = ws.numel()
            if tail.ws_wm is None or tail.ws_wm.numel() < need:
                tail.ws_wm = torch.empty(need, dtype=torch.uint8, device=dev)
            ws = tail.ws_wm
            tail.ev_fork.record(torch.cuda.current_stream())
            tail.s_wm.wait_event(tail.ev_fork)
            pk['tail'] = tail
        with (torch.cuda.stream(tail.s_wm) if use_tail else contextlib.nullcontext()):
            return self._forward_tail(pk, obs, c, dec, dec_p, dec_acts, loss_image, ws, mbuf, chunks, image, action, reset,
                                      enc_acts, rssm_acts)

    def _forward_tail(self, pk, obs, c, dec, dec_p, dec_acts, loss_image, ws, mbuf, chunks, image, action, reset, enc_acts,
                      rssm_acts):
        shp, T, B, I, feat, post, prior, idx = (pk[k] for k in ('shp', 'T', 'B', 'I', 'feat', 'post', 'prior', 'idx'))
        lib = H.lib()
        N, NE, dev = T * B * I, T * B, feat.device
        D_, F_ = c.deter_dim, self.features_dim
        gauss = not c.stoch_discrete
        Z = c.stoch_dim * (c.stoch_discrete or 1)
        if chunks <= 1:
            H.call('dm_conv_decoder_mse_fwd', ctypes.byref(shp), H.fptr(feat), F_, H.ptr(image), ctypes.byref(dec_p),
                   H.fptr(dec_acts), H.fptr(loss_image), None, H.ptr(ws), ws.numel(), H.stream())

        reward_t = obs['reward'].float().contiguous()
        terminal_t = obs['terminal'].float().contiguous()
        if I > 1:                                  # targets expanded over I (decoders.py:270,305: insert_dim)
            reward_t = reward_t.repeat_interleave(I, dim=1).contiguous()
            terminal_t = terminal_t.repeat_interleave(I, dim=1).contiguous()
        sp = 0 if c.stoch_discrete == 0 else c.stoch_dim * c.stoch_discrete      # the one-hot latent columns of a feature row
        mu, r_acts = dec.reward.model.fwd(feat, F_, N, ws, sparse_cols=sp)
        tl, t_acts = dec.terminal.model.fwd(feat, F_, N, ws, sparse_cols=sp)
        loss_reward, dmu, reward_rec = (torch.empty(N, device=dev) for _ in range(3))
        loss_terminal, dtl, terminal_rec = (torch.empty(N, device=dev) for _ in range(3))
        # -Normal(mu, std).log_prob(y) * std^2 = 0.5 (mu-y)^2 + std^2 (log std + log sqrt(2 pi))   (decoders.py:296-304)
        loss_const = REWARD_STD ** 2 * (math.log(REWARD_STD) + math.log(math.sqrt(2 * math.pi)))
        # gw: the data-parallel shard weight B_r/B (dist.attach), folded into every gradient scale; loss values are unaffected
        gw = float(getattr(self, 'grad_weight', 1.0))
        H.call('dm_head_loss', 0, N, H.fptr(mu), H.fptr(reward_t), gw * dec.reward_weight / NE, loss_const, H.fptr(loss_reward),
               H.fptr(dmu), H.fptr(reward_rec), H.stream())
        H.call('dm_head_loss', 1, N, H.fptr(tl), H.fptr(terminal_t), gw * dec.terminal_weight / NE, 0.0, H.fptr(loss_terminal),
               H.fptr(dtl), H.fptr(terminal_rec), H.stream())

        # KL + entropies (dreamer.py:326-343,369-379)
        kl, ent_post, ent_prior = (torch.empty(N, device=dev) for _ in range(3))
        H.call('dm_kl_balance_fwd', N, c.stoch_dim, c.stoch_discrete, H.fptr(post), H.fptr(prior), H.fptr(kl),
               H.fptr(ent_post), H.fptr(ent_prior), H.stream())

        if mbuf is None:
            mbuf = torch.zeros(METRIC_BUF_FLOATS, device=dev)
        loss = mbuf[6]
        w = (ctypes.c_float * 4)(self.kl_weight, dec.image_weight, dec.reward_weight, dec.terminal_weight)
        iw = None
        aux = None
        if self.ac_aux is not None and T >= 2:
            # Auxiliary critic (dreamer.py:347-358): ActorCritic.training_step on the REAL trajectory - features (T,B,F) as
            # the "dream", the batch's own actions / rewards / terminals; only loss_critic is kept.  Its actor forward and
            # policy loss are computed by the reference and then thrown away, so they are simply not run here.
            ac = self.ac_aux
            # a2c.py:76-79 via dreamer.py:349: ac_aux.training_step runs with log_only=False also under no_grad evaluation,
            # so the target refresh and the counter advance on EVERY call (the reference's behaviour, mirrored)
            if ac.train_steps % ac.target_interval == 0:
                ac.update_critic_target()
            ac.train_steps += 1
            rows = (T - 1) * B
            value_t, _ = ac.critic_target.fwd(feat, F_, N, ws, save_acts=False, sparse_cols=0 if gauss else Z)
            value, aux_acts = ac.critic.fwd(feat, F_, N, ws, sparse_cols=0 if gauss else Z)
            adv, agae, vtgt, wgt = (torch.empty(T - 1, B, device=dev) for _ in range(4))
            H.call('dm_gae_losses', T - 1, B, ac.gamma, ac.lambda_, H.fptr(reward_t), H.fptr(terminal_t), H.fptr(value_t),
                   H.fptr(adv), H.fptr(agae), H.fptr(vtgt), H.fptr(wgt), H.stream())
            lc = torch.empty(rows, device=dev)
            dvalue = torch.zeros(N, device=dev)                      # value[-1] gets no gradient
            H.call('dm_critic_loss', rows, H.fptr(value), H.fptr(vtgt), H.fptr(wgt), gw * self.aux_critic_weight / rows, H.fptr(lc),
                   H.fptr(dvalue), H.stream())
            _multi_sum([(lc, 1.0 / rows), (value.view(T, B)[:-1], 1.0 / rows)], dev, out=mbuf[24:26])
            aux = dict(acts=aux_acts, dvalue=dvalue, value=value, lc=lc, rows=rows)
        if I == 1:
            means = _multi_sum([(kl, 1.0 / N), (loss_image, 1.0 / N), (loss_reward, 1.0 / N), (loss_terminal, 1.0 / N),
                                (ent_prior, 1.0 / N), (ent_post, 1.0 / N)], dev, out=mbuf[0:6])
            H.call('dm_combine', 4, H.fptr(means), w, ctypes.c_void_p(mbuf.data_ptr() + 24), H.stream())   # dreamer.py:362-365
            if aux is not None:       # loss = loss_model.mean() + aux_critic_weight * loss_critic_aux (dreamer.py:365) -> slot 7
                t5 = _multi_sum([(kl, 1.0 / N), (loss_image, 1.0 / N), (loss_reward, 1.0 / N), (loss_terminal, 1.0 / N),
                                 (aux['lc'], 1.0 / aux['rows'])], dev)
                w5 = (ctypes.c_float * 5)(self.kl_weight, dec.image_weight, dec.reward_weight, dec.terminal_weight,
                                          self.aux_critic_weight)
                H.call('dm_combine', 5, H.fptr(t5), w5, ctypes.c_void_p(mbuf.data_ptr() + 28), H.stream())
                loss = mbuf[7]
            tb = lambda x: x.view(T, B)
            t_kl, t_ep, t_eq, t_li, t_lr, t_lt, t_rr, t_tr = (tb(x) for x in (kl, ent_prior, ent_post, loss_image, loss_reward,
                                                                              loss_terminal, reward_rec, terminal_rec))
        else:
            # IWAE: sampled KL (dreamer.py:340-343); loss_model = mean_tb -logavgexp_i(-loss_tbi) (dreamer.py:362-365);
            # the logged tensors are -logavgexp_i(-x) of the per-sample losses (decoders.py:170,277,312), means over I for the
            # entropies and the reconstructions (dreamer.py:371-372, decoders.py:171,278,313)
            kl_s = torch.empty(N, device=dev)
            if gauss:      # Normal log-densities of the reparameterised sample z = feat[:, D:] (rssm.py:202-203)
                H.call('dm_kl_sampled_gauss_fwd', N, c.stoch_dim, H.fptr(post), H.fptr(prior),
                       ctypes.c_void_p(feat.data_ptr() + 4 * D_), F_, H.fptr(kl_s), H.stream())
            else:
                H.call('dm_kl_sampled_fwd', N, c.stoch_dim, c.stoch_discrete, H.fptr(post), H.fptr(prior), H.ptr(idx), H.fptr(kl_s),
                       H.stream())
            l_tbi = torch.empty(N, device=dev)
            ptrs = (ctypes.c_void_p * 4)(kl_s.data_ptr(), loss_image.data_ptr(), loss_reward.data_ptr(), loss_terminal.data_ptr())
            H.call('dm_combine_rows', 4, N, ptrs, w, H.fptr(l_tbi), H.stream())
            iw = torch.empty(N, device=dev)                   # importance weights softmax_i(-loss_tbi) = d loss_tb / d loss_tbi
            red = torch.empty(9, NE, device=dev)
            H.call('dm_reduce_i', NE, I, 1, H.fptr(l_tbi), 1, H.fptr(red[0]), H.fptr(iw), H.stream())
            for j, (x, mode) in enumerate(((kl, 1), (loss_image, 1), (loss_reward, 1), (loss_terminal, 1), (ent_prior, 0),
                                           (ent_post, 0), (reward_rec, 0), (terminal_rec, 0))):
                H.call('dm_reduce_i', NE, I, 1, H.fptr(x), mode, H.fptr(red[j + 1]), None, H.stream())
            _multi_sum([(red[1], 1.0 / NE), (red[2], 1.0 / NE), (red[3], 1.0 / NE), (red[4], 1.0 / NE), (red[5], 1.0 / NE),
                        (red[6], 1.0 / NE), (red[0], 1.0 / NE)], dev, out=mbuf[0:7])
            means = mbuf[0:6]
            tb = lambda x: x.view(T, B)
            t_kl, t_li, t_lr, t_lt, t_ep, t_eq, t_rr, t_tr = (tb(red[j]) for j in range(1, 9))

        pk.update(loss=loss, image=image, action=action, reset=reset, enc_acts=enc_acts, rssm_acts=rssm_acts,
                  dec_acts=dec_acts, r_acts=r_acts, t_acts=t_acts, dmu=dmu, dtl=dtl, ws=ws, mbuf=mbuf, iw=iw, aux=aux)
        pk['tensors'] = LazyTensors(loss_kl=t_kl, entropy_prior=t_ep, entropy_post=t_eq,
                                    loss_image=t_li, image_rec=None,
                                    loss_reward=t_lr, reward_rec=t_rr,
                                    loss_terminal=t_lt, terminal_rec=t_tr)
        Cc, hw = c.image_channels, c.image_size * c.image_size
        pred_off = int(lib.dm_conv_decoder_pred_offset(ctypes.byref(shp)))

        def image_rec_thunk(acts=dec_acts):       # holds the decoder activations alive until the dict is dropped
            with torch.no_grad():
                pred = acts[pred_off:pred_off + N * hw * Cc].view(N, hw, Cc)                       # NHWC
                rec = pred.transpose(1, 2).contiguous().view(T, B, I, Cc, c.image_size, c.image_size)   # -> (T,B,I,C,H,W)
                return rec[:, :, 0] if I == 1 else rec.mean(2)       # decoded.mean(dim=2), decoders.py:171 (logging only)
        pk['tensors'].lazy('image_rec', image_rec_thunk)
        pk['metrics'] = dict(loss_model=mbuf[6], loss_kl=means[0], entropy_prior=means[4], entropy_post=means[5],
                             loss_image=means[1], loss_reward=means[2], loss_terminal=means[3])
        if aux is not None:
            pk['metrics'].update(loss_critic_aux=mbuf[24], policy_value_aux=mbuf[25])
            pk['tensors']['policy_value_aux'] = aux['value'].view(T, B)
        return pk

    def _param_order(self):
        return list(self.parameters())

    def _backward(self, pk, ws, scratch=False, defer_wgrad=False):
        c = self.conf
        shp, T, B, I = pk['shp'], pk['T'], pk['B'], pk.get('I', 1)
        NE, N = T * B, T * B * I
        iw = pk.get('iw')          # IWAE importance weights (N,) or None: every per-sample gradient of loss_model carries them
        feat, dev = pk['feat'], pk['feat'].device
        F_, Z, E = self.features_dim, c.stoch_dim * (c.stoch_discrete or 2), self.encoder.out_dim   # Z: parameter width here
        if not self._arena.current(pk.get('gen')):
            raise RuntimeError('the activations saved by this training_step() were overwritten by a later training_step() '
                               '(they live in per-model buffers with stable addresses); call backward() after each '
                               'training_step()')
        ar = self._arena
        plist = self._param_order()
        flat, views, direct = _flat_views(plist, dev, getattr(self, '_fused', None), scratch)
        gof = {id(p): v for p, v in zip(plist, views)}
        dec = self.decoder

        gw = float(getattr(self, 'grad_weight', 1.0))          # data-parallel shard weight B_r/B (dist.attach)
        dfeat = ar.get('dfeat', (N, F_), device=dev).zero_()
        # dense heads (decoders.py:73-83)
        if iw is not None and not pk.get('iw_applied'):
            for dout in (pk['dmu'], pk['dtl']):
                H.call('dm_scale_rows', N, 1, H.fptr(dout), 1, H.fptr(iw), 1.0, H.stream())
            pk['iw_applied'] = True
        heads = [(dec.reward.model, pk['r_acts'], pk['dmu']), (dec.terminal.model, pk['t_acts'], pk['dtl'])]
        if pk.get('aux') is not None:          # the auxiliary critic trains on the world model's own features (not detached)
            heads.append((self.ac_aux.critic, pk['aux']['acts'], pk['aux']['dvalue']))
        for head, acts, dout in heads:
            st, gs = head.struct(), head.grad_struct(gof)
            H.call('dm_mlp_head_bwd', N, F_, head.hidden_dim, head.hidden_layers, 1, H.fptr(feat), F_, ctypes.byref(st),
                   H.fptr(acts), H.fptr(dout), ctypes.byref(gs), H.fptr(dfeat), F_, 1, H.ptr(ws), ws.numel(), H.stream())
        # image decoder
        dl = dec.image.layers()
        dec_p = H.conv_struct([m.weight for m in dl], [m.bias for m in dl])
        dec_g = H.conv_struct([gof[id(m.weight)] for m in dl], [gof[id(m.bias)] for m in dl], cls=H.dm_conv_grads)
        # Parameter gradients are leaves: nothing waits for them before the gradient clip, while the DATA gradients are the
        # chain decoder -> BPTT -> encoder.  Armed, the library enqueues the decoder's and the RSSM's parameter gradients on its
        # low-priority side stream, where they run beside the BPTT loop (a 50-row latency chain that leaves most CUs idle); they
        # read gradient buffers in the workspace of THEIR call, so each call gets its own (include/dreamer_hip.h
        # dm_wgrad_side_arm).  The encoder backward is not deferred (it is the tail: there is nothing left to hide behind).
        # (measured: no gain on a 7-column shard, -1.1 ms at 25 columns.  Rounds 3-5 switched it off under data parallelism, unmeasured;
        #  round 6 measured it over a one-rank RCCL group: off costs +0.9 ... +1.2 ms at 25 / 50 columns, on costs nothing beside the
        #  late all-reduce or the library's own - profiles/r06_force_dp.txt run K.  DM_WGRAD_SIDE_DP=0 restores the old behaviour.)
        dp_on = getattr(getattr(self, '_fused', None), 'dp', None) is not None and not _WGRAD_SIDE_DP
        side = defer_wgrad and _WGRAD_SIDE and B * I >= 16 and not dp_on and not torch.cuda.is_current_stream_capturing()
        ws_dec = ws_enc = ws
        if side:
            need = H.workspace_bytes(shp)
            if self._ws_dec is None or self._ws_dec.numel() < need or self._ws_dec.device != dev:
                self._ws_dec = torch.empty(need, dtype=torch.uint8, device=dev)
                self._ws_enc = torch.empty(need, dtype=torch.uint8, device=dev)
            ws_dec, ws_enc = self._ws_dec, self._ws_enc       # decoder / BPTT (`ws`) / encoder: one workspace each until the join
            H.call('dm_wgrad_side_arm', 1)
        try:
            H.call('dm_conv_decoder_mse_bwd_rows', ctypes.byref(shp), H.fptr(feat), F_, H.ptr(pk['image']), ctypes.byref(dec_p),
                   H.fptr(pk['dec_acts']), gw * dec.image_weight / NE, H.fptr(iw), ctypes.byref(dec_g), H.fptr(dfeat), F_,
                   H.ptr(ws_dec), ws_dec.numel(), H.stream())
            # KL (dreamer.py:334-343)
            dpost = ar.get('dpost', (N, Z), device=dev)
            dprior = ar.get('dprior', (N, Z), device=dev)
            if iw is not None and not c.stoch_discrete:      # sampled Normal KL: explicit parameter gradients + the path through z
                D_ = c.deter_dim
                H.call('dm_kl_sampled_gauss_bwd', N, c.stoch_dim, H.fptr(pk['post']), H.fptr(pk['prior']),
                       ctypes.c_void_p(feat.data_ptr() + 4 * D_), F_, gw * self.kl_weight / NE, H.fptr(iw), H.fptr(dpost),
                       H.fptr(dprior), ctypes.c_void_p(dfeat.data_ptr() + 4 * D_), F_, H.stream())
            elif iw is not None:         # sampled KL of the IWAE bound
                H.call('dm_kl_sampled_bwd', N, c.stoch_dim, c.stoch_discrete, H.fptr(pk['post']), H.fptr(pk['prior']),
                       H.ptr(pk['idx']), gw * self.kl_weight / NE, H.fptr(iw), H.fptr(dpost), H.fptr(dprior), H.stream())
            else:
                if self.kl_balance is None:
                    sp = sq = gw * self.kl_weight / N
                else:
                    sp, sq = gw * self.kl_weight * (1 - self.kl_balance) / N, gw * self.kl_weight * self.kl_balance / N
                H.call('dm_kl_balance_bwd', N, c.stoch_dim, c.stoch_discrete, H.fptr(pk['post']), H.fptr(pk['prior']), sp, sq,
                       H.fptr(dpost), H.fptr(dprior), H.stream())
            # RSSM BPTT
            cell = self.core.cell
            rssm_p = H.rssm_struct(cell.ordered())
            rssm_g = H.rssm_struct([None if p is None else gof[id(p)] for p in cell.ordered()], cls=H.dm_rssm_grads)
            dembed = ar.get('dembed', (N, E), device=dev)
            H.call('dm_rssm_sequence_bwd', ctypes.byref(pk['shp_r']), H.fptr(pk['embed_x']), H.fptr(pk['action_x']),
                   H.ptr(pk['reset_x']), ctypes.byref(rssm_p), H.fptr(pk['rssm_acts']), H.fptr(feat), H.fptr(pk['post']),
                   H.fptr(dfeat), H.fptr(dpost), H.fptr(dprior), ctypes.byref(rssm_g), H.fptr(dembed), H.ptr(ws), ws.numel(),
                   H.stream())
            if I > 1:                  # the I samples of a (t,b) share one embedding row: their gradients add up
                dsum = torch.empty(NE, E, device=dev)
                H.call('dm_reduce_i', NE, I, E, H.fptr(dembed), 2, H.fptr(dsum), None, H.stream())
                dembed = dsum
            # encoder
            enc = self.encoder.encoder_image
            enc_p = H.conv_struct([m.weight for m in enc.convs()], [m.bias for m in enc.convs()])
            enc_g = H.conv_struct([gof[id(m.weight)] for m in enc.convs()], [gof[id(m.bias)] for m in enc.convs()],
                                  cls=H.dm_conv_grads)
            H.call('dm_conv_encoder_bwd', ctypes.byref(pk['shp_e']), H.ptr(pk['image']), ctypes.byref(enc_p),
                   H.fptr(pk['enc_acts']), H.fptr(dembed), ctypes.byref(enc_g), H.ptr(ws_enc), ws_enc.numel(), H.stream())
        except BaseException:
            if side:
                H.lib().dm_wgrad_side_join(H.stream())      # disarm this thread; the error propagates
            raise
        if side:
            H.call('dm_wgrad_side_join', H.stream())    # every gradient of this pass is complete on this stream from here on
        return views, flat, direct

    def _image_pred(self, pk, obs, u_pred):
        """do_image_pred (dreamer.py:381-394): decode from a PRIOR sample instead of the posterior sample and report the
        reconstruction losses as logprob_* / *_pred (decoders.py:50-108 with extra_metrics).  Logging variant, no grads.
        iwae_samples = I > 1 (the call shape of evaluate(), train.py:353-359,380-385): one prior sample per (t,b,i) row; the
        decoders reduce TBI => TB by -logavgexp(-loss) for the losses and a mean for the predictions
        (decoders.py:170-171,277-278,312-313).  The NaN-masked sign / terminal splits (decoders.py:94-105) are (T,B) torch
        expressions."""
        c, dec = self.conf, self.decoder
        T, B, I, feat, shp = pk['T'], pk['B'], pk.get('I', 1), pk['feat'], pk['shp']
        NE, N, dev = T * B, T * B * I, feat.device
        D_, S, C, F_ = c.deter_dim, c.stoch_dim, c.stoch_discrete, self.features_dim
        ws = self.workspace(shp, dev)
        lib = H.lib()
        with torch.no_grad():
            if u_pred is None:
                u_pred = (torch.rand if C else torch.randn)(N, S, device=dev)
            u_pred = u_pred.reshape(N, S).contiguous()
            fp = feat.clone()                                                 # feature_replace_z: [h | prior sample]
            idx = torch.empty(N, S, dtype=torch.int32, device=dev)
            H.call('dm_sample_onehot', N, S, C, H.fptr(pk['prior']), S * (C or 2), H.fptr(u_pred), None,
                   ctypes.c_void_p(fp.data_ptr() + 4 * D_), F_, H.ptr(idx), H.stream())
            dl = dec.image.layers()
            dec_p = H.conv_struct([m.weight for m in dl], [m.bias for m in dl])
            acts = torch.empty(int(lib.dm_conv_decoder_acts_floats(ctypes.byref(shp))), device=dev)
            li = torch.empty(N, device=dev)
            image_pred = torch.empty(T, B, I, c.image_channels, c.image_size, c.image_size, device=dev)
            H.call('dm_conv_decoder_mse_fwd', ctypes.byref(shp), H.fptr(fp), F_, H.ptr(pk['image']), ctypes.byref(dec_p),
                   H.fptr(acts), H.fptr(li), H.fptr(image_pred), H.ptr(ws), ws.numel(), H.stream())
            mu, _ = dec.reward.model.fwd(fp, F_, N, ws, save_acts=False)
            tl, _ = dec.terminal.model.fwd(fp, F_, N, ws, save_acts=False)
            lr, lt, rp, tp, scratch = (torch.empty(N, device=dev) for _ in range(5))
            loss_const = REWARD_STD ** 2 * (math.log(REWARD_STD) + math.log(math.sqrt(2 * math.pi)))
            reward_t, terminal_t = obs['reward'].float().contiguous(), obs['terminal'].float().contiguous()
            if I > 1:                              # targets expanded over I (decoders.py:270,305: insert_dim)
                reward_t = reward_t.repeat_interleave(I, dim=1).contiguous()
                terminal_t = terminal_t.repeat_interleave(I, dim=1).contiguous()
            H.call('dm_head_loss', 0, N, H.fptr(mu), H.fptr(reward_t), 0.0, loss_const, H.fptr(lr),
                   H.fptr(scratch), H.fptr(rp), H.stream())
            H.call('dm_head_loss', 1, N, H.fptr(tl), H.fptr(terminal_t), 0.0, 0.0, H.fptr(lt),
                   H.fptr(scratch), H.fptr(tp), H.stream())
            if I > 1:                              # TBI => TB
                red = torch.empty(5, NE, device=dev)
                for j, (x, mode) in enumerate(((li, 1), (lr, 1), (lt, 1), (rp, 0), (tp, 0))):
                    H.call('dm_reduce_i', NE, I, 1, H.fptr(x), mode, H.fptr(red[j]), None, H.stream())
                li, lr, lt, rp, tp = (red[j] for j in range(5))
                image_pred = image_pred.mean(2)                               # decoded.mean(dim=2), decoders.py:171
            else:
                image_pred = image_pred[:, :, 0]
            tb = lambda x: x.view(T, B)
            tensors = dict(logprob_image=tb(li), logprob_reward=tb(lr), logprob_terminal=tb(lt),
                           image_pred=image_pred, reward_pred=tb(rp), terminal_pred=tb(tp))
            metrics = dict(logprob_image=li.mean(), logprob_reward=lr.mean(), logprob_terminal=lt.mean())
            nan = torch.full((), float('nan'), device=dev)
            nanmean = lambda x: torch.nansum(x) / (~torch.isnan(x)).sum()     # functions.py:149-150
            for sig in (-1, 1):
                lp = torch.where(torch.sign(obs['reward'].float()) == sig, tb(lr), nan)
                metrics[f'logprob_reward{sig}'], tensors[f'logprob_reward{sig}'] = nanmean(lp), lp
            lp = torch.where(obs['terminal'].float() > 0, tb(lt), nan)
            metrics['logprob_terminal1'], tensors['logprob_terminal1'] = nanmean(lp), lp
        return metrics, tensors, idx

    def training_step(self, obs, in_state, iwae_samples=1, do_open_loop=False, do_image_pred=False, forward_only=False,
                      u_post=None, forced_idx=None, imag_horizon=1, u_pred=None, mbuf=None, _internal=False, _tail=None):
        """dreamer.py:297-396. Returns (loss, features (T,B,1,F), states, out_state, metrics, tensors)."""
        I = int(iwae_samples)
        if do_open_loop and torch.is_grad_enabled():
            raise NotImplementedError('do_open_loop is an evaluation variant: call it under torch.no_grad() like '
                                      'train.py:353-359 does (its backward is not built)')
        T, B = obs['action'].shape[:2]
        if forward_only:
            feats, out_state = self.forward(obs, in_state)
            return torch.tensor(0.0), feats, None, out_state, {}, {}
        pk = self._forward(obs, in_state, u_post, forced_idx, imag_horizon=imag_horizon, open_loop=do_open_loop, mbuf=mbuf,
                           iwae=I, tail=None if (do_open_loop or do_image_pred) else _tail)
        loss = _WMStep.apply(self, pk, *self._param_order())
        if pk.get('tail') is not None:        # everything the forward's tail produced is final once the caller's stream has
            pk['tail'].ev_tail.record(pk['tail'].s_wm)      # waited for this event (Dreamer.training_step does, before returning)
        D_ = self.deter_dim
        # the feature matrix lives in the step arena (overwritten by the next step): callers get their own copy, except
        # Dreamer.training_step, which consumes it before returning (_internal)
        feat = pk['feat'] if (_internal or not self._arena.on) else pk['feat'].clone()
        features = feat.view(T, B, I, -1)
        states = (feat[:, :D_].view(T, B, I, -1), feat[:, D_:].view(T, B, I, -1))
        self._last_pack = pk
        if do_image_pred:
            m, t, pk['pred_idx'] = self._image_pred(pk, obs, u_pred)
            pk['metrics'] = dict(pk['metrics'], **m)
            pk['tensors'].update(t)
        return loss, features, states, pk['out_state'], pk['metrics'], pk['tensors']


# ---------------------------------------------------------------------------------------------------------------
# actor critic
# ---------------------------------------------------------------------------------------------------------------
class _HeadLoss(torch.autograd.Function):
    """A scalar loss whose gradient w.r.t. one MLP's parameters is produced by dm_mlp_head_bwd."""

    @staticmethod
    def forward(ctx, mlp, pack, *params):
        ctx.mlp, ctx.pack = mlp, pack
        return pack['loss'].clone()

    @staticmethod
    def backward(ctx, grad_loss):
        mlp, pk = ctx.mlp, ctx.pack
        ov = pk.get('overlap')
        if 'pre' in pk:                                   # launched on s_ac inside training_step()
            grads, flat, direct = pk.pop('pre').result()
            fused = getattr(mlp, '_fused', None)
            if fused is None or fused.home is not ov.s_ac or flat is not fused.scratch:
                torch.cuda.current_stream().wait_stream(ov.s_ac)
            else:      # what the backward pass reads was allocated on the caller's stream and is released below: not before s_ac is done with it
                for t in (pk.get('x'), pk.get('acts'), pk.get('dout')):
                    if torch.is_tensor(t):
                        t.record_stream(ov.s_ac)
            # else (Dreamer.pipeline_ac_optimizer): the hand-over, the clip and the AdamW step of this group are enqueued on
            # s_ac behind the backward pass; the caller's stream goes on to the next step's forward without waiting for it
        else:
            grads, flat, direct = mlp.bwd(pk['x'], pk['ldx'], pk['rows'], pk['acts'], pk['dout'], pk['ws'])
        pk.pop('acts', None)
        return (None, None) + _finish_backward(mlp, grads, flat, direct, grad_loss)


class ActorCritic(_Params):
    """a2c.py:11-152."""

    def __init__(self, in_dim, out_actions, hidden_dim=400, hidden_layers=4, layer_norm=True, gamma=0.999, lambda_gae=0.95,
                 entropy_weight=1e-3, target_interval=100, actor_grad='reinforce', actor_dist='onehot'):
        super().__init__()
        if actor_dist not in ACTOR_KINDS or actor_grad != 'reinforce':
            raise NotImplementedError(f'actor_dist={actor_dist!r} / actor_grad={actor_grad!r}: the HIP path builds onehot, '
                                      f'tanh_normal and normal_tanh actors with actor_grad=reinforce (actor_grad=dynamics '
                                      f'asserts in the reference itself, SURVEY 0.5)')
        self.dist_kind = ACTOR_KINDS[actor_dist]
        self.in_dim, self.out_actions = in_dim, out_actions
        self.gamma, self.lambda_, self.entropy_weight = gamma, lambda_gae, entropy_weight
        self.target_interval, self.actor_grad, self.actor_dist = target_interval, actor_grad, actor_dist
        actor_out_dim = out_actions if actor_dist == 'onehot' else 2 * out_actions      # a2c.py:35
        self.actor = MLP(in_dim, actor_out_dim, hidden_dim, hidden_layers, layer_norm)
        self.critic = MLP(in_dim, 1, hidden_dim, hidden_layers, layer_norm)
        self.critic_target = MLP(in_dim, 1, hidden_dim, hidden_layers, layer_norm)
        self.critic_target.requires_grad_(False)
        self.train_steps = 0
        self.sparse_cols = 0                  # trailing feature columns known to be one-hot samples (set by the owner)
        self.defer_target_update = False      # a caller that captures the step into a graph does the refresh + counter itself

    def update_critic_target(self):
        """a2c.py:151-152."""
        with torch.no_grad():
            for dst, src in zip(self.critic_target.parameters(), self.critic.parameters()):
                H.call('dm_copy_params', H.fptr(dst), H.fptr(src), dst.numel(), H.stream())

    def begin_step(self, log_only=False):
        """The head of training_step (a2c.py:76-79): refresh the target network every target_interval steps, count the step."""
        if not log_only and not self.defer_target_update:
            if self.train_steps % self.target_interval == 0:
                self.update_critic_target()
            self.train_steps += 1

    @staticmethod
    def split_steps(J):
        """The heads over the (J, M) imagined states run as two row windows, steps [0, split) and [split, J): the first one can
        start while the rollout is still producing the last steps (Dreamer._heads_forward).  Always the same split, so every
        execution order computes the same numbers.  Short horizons: one window."""
        return (J * 5) // 8 if J >= 9 else J

    def value_buffers(self, rows, device):
        return dict(value_t=torch.empty(rows, 1, device=device), value=torch.empty(rows, 1, device=device),
                    c_acts=torch.empty(self.critic.acts_floats(rows), device=device))

    def forward_values(self, feats, F_, rows_total, row0, rows, ws, bufs):
        """critic_target and critic over rows [row0, row0 + rows) of the feature matrix (a2c.py:85,113)."""
        sp = self.sparse_cols        # the one-hot latent columns at the end of a feature row (0: unknown)
        self.critic_target.fwd(feats, F_, rows, ws, save_acts=False, sparse_cols=sp, window=(rows_total, row0), out=bufs['value_t'])
        self.critic.fwd(feats, F_, rows, ws, acts=bufs['c_acts'], sparse_cols=sp, window=(rows_total, row0), out=bufs['value'])

    def training_step(self, features, actions, rewards, terminals, log_only=False, act_idx=None, ws=None,
                      actor_acts=None, actor_logits=None, overlap=None, mbuf=None, values=None, _begun=False):
        """features (J,M,F), actions (H,M,A) one-hot, rewards/terminals (J,M). a2c.py:61-149.
        actor_acts / actor_logits: forward_actor(features[:-1]) as already computed by the dream rollout on the same
        features and weights (bit-identical to recomputing it, which is what the reference does, a2c.py:119)."""
        _require_cuda(features, 'features')
        if not _begun:
            self.begin_step(log_only)
        J, M, F_ = features.shape
        Hh, A, dev = J - 1, self.out_actions, features.device
        feats = features.contiguous().view(J * M, F_)
        rewards, terminals = rewards.contiguous(), terminals.contiguous()
        if self.dist_kind == 0:
            if act_idx is None:
                act_idx = actions.argmax(-1).to(torch.int32)
            act_idx = act_idx.contiguous().view(-1)
        if ws is None:
            raise H.DreamerHipError('ActorCritic.training_step needs the model workspace (called through Dreamer.training_step)')

        sp = self.sparse_cols        # the one-hot latent columns at the end of a feature row (0: unknown)
        if values is None:           # (Dreamer.training_step hands them over: computed next to the reward / terminal heads)
            values = self.value_buffers(J * M, dev)
            r0 = self.split_steps(J) * M
            for a, b in ((0, r0), (r0, J * M)):
                if b > a:
                    self.forward_values(feats, F_, J * M, a, b - a, ws, values)
        value_t, value, c_acts = values['value_t'], values['value'], values['c_acts']
        if actor_acts is not None:
            logits, a_acts = actor_logits, actor_acts
        else:
            logits, a_acts = self.actor.fwd(feats, F_, Hh * M, ws, sparse_cols=sp)   # features[:-1] = first H*M rows
        adv, agae, vtgt, wgt = (torch.empty(Hh, M, device=dev) for _ in range(4))
        H.call('dm_gae_losses', Hh, M, self.gamma, self.lambda_, H.fptr(rewards), H.fptr(terminals), H.fptr(value_t),
               H.fptr(adv), H.fptr(agae), H.fptr(vtgt), H.fptr(wgt), H.stream())
        rows = Hh * M
        lc = torch.empty(rows, device=dev)
        dvalue = torch.zeros(J * M, device=dev)                      # value[-1] gets no gradient
        gw = float(getattr(self, 'grad_weight', 1.0))          # data-parallel shard weight B_r/B (dist.attach), gradients only
        H.call('dm_critic_loss', rows, H.fptr(value), H.fptr(vtgt), H.fptr(wgt), gw / rows, H.fptr(lc), H.fptr(dvalue),
               H.stream())
        la, ent = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        dlogits = torch.empty(rows, self.actor.out_dim, device=dev)
        if self.dist_kind == 0:
            H.call('dm_actor_loss', rows, A, H.fptr(logits), H.ptr(act_idx), H.fptr(agae), H.fptr(wgt), self.entropy_weight,
                   gw / rows, H.fptr(la), H.fptr(ent), H.fptr(dlogits), H.stream())
        else:
            H.call('dm_actor_loss_continuous', self.dist_kind, rows, A, H.fptr(logits), H.fptr(actions.contiguous()),
                   H.fptr(agae), H.fptr(wgt), self.entropy_weight, gw / rows, H.fptr(la), H.fptr(ent), H.fptr(dlogits),
                   H.stream())
        value2d = value.view(J, M)
        reward1 = rewards.view(J, M)[1:]
        if mbuf is None:
            mbuf = torch.zeros(METRIC_BUF_FLOATS, device=dev)
        s = _multi_sum([(lc, 1.0 / rows), (la, 1.0 / rows), (ent, 1.0 / rows), (value2d[0], 1.0 / M),
                        (value2d[:-1], 1.0 / rows), (reward1, 1.0 / rows)], dev, out=mbuf[8:14])
        std = _multi_sum([(reward1, 1.0 / max(rows - 1, 1), s[5:6], 2)], dev, out=mbuf[14:15])   # unbiased std (a2c.py:140)
        loss_critic_v, loss_actor_v = s[0], s[1]
        if log_only:
            loss_actor, loss_critic = loss_actor_v, loss_critic_v
        else:
            pa = dict(loss=loss_actor_v, x=feats, ldx=F_, rows=rows, acts=a_acts, dout=dlogits, ws=ws, overlap=overlap)
            pc = dict(loss=loss_critic_v, x=feats, ldx=F_, rows=J * M, acts=c_acts, dout=dvalue.view(J * M, 1), ws=ws,
                      overlap=overlap)
            loss_actor = _HeadLoss.apply(self.actor, pa, *self.actor.param_list())
            loss_critic = _HeadLoss.apply(self.critic, pc, *self.critic.param_list())
            self._last_packs = (pa, pc)
        metrics = dict(loss_critic=loss_critic_v, loss_actor=loss_actor_v, policy_entropy=s[2], policy_value=s[3],
                       policy_value_im=s[4], policy_reward=s[5], policy_reward_std=std[0])
        tensors = dict(value=value2d, value_target=vtgt, value_advantage=adv, value_advantage_gae=agae, value_weight=wgt)
        return (loss_actor, loss_critic), metrics, tensors


# ---------------------------------------------------------------------------------------------------------------
# Dreamer
# ---------------------------------------------------------------------------------------------------------------
class Dreamer(nn.Module):
    """dreamer.py:19-226."""

    def __init__(self, conf):
        super().__init__()
        assert conf.action_dim > 0, 'Need to set action_dim to match environment'
        if conf.probe_model != 'none':
            raise NotImplementedError('probe models are research heads outside the hot path')
        features_dim = conf.deter_dim + conf.stoch_dim * (conf.stoch_discrete or 1)
        self.conf = conf
        self.iwae_samples, self.imag_horizon = conf.iwae_samples, conf.imag_horizon
        self.wm = WorldModel(conf)
        self.ac = ActorCritic(in_dim=features_dim, out_actions=conf.action_dim, layer_norm=conf.layer_norm, gamma=conf.gamma,
                              lambda_gae=conf.lambda_gae, entropy_weight=conf.entropy, target_interval=conf.target_interval,
                              actor_grad=conf.actor_grad, actor_dist=conf.actor_dist)
        self.ac.sparse_cols = conf.stoch_dim * conf.stoch_discrete       # feature = [h | one-hot z] (rssm.py:83-84)
        self.probe_model = NoProbeHead()
        self.probe_gradients = conf.probe_gradients
        self._groups = None
        # conf.amp (defaults.yaml:55; train.py:166 runs the step under autocast): GEMM operands in bf16, fp32 accumulation,
        # fp32 storage.  Precision is an argument of every library call (dm_shape.flags bit DM_FLAG_BF16 via
        # WorldModel.shape(), dm_mlp_params.precision via MLP.struct()); there is no process-wide switch.
        self.amp = bool(getattr(conf, 'amp', False))
        for m in self.modules():
            if isinstance(m, MLP):
                m.precision = int(self.amp)
        self._overlap = None
        self.overlap_backward = True      # pre-launch the three backward passes on side streams (see _Overlap)
        # the world-model forward's tail (decoder, heads, losses) on the world-model stream: WorldModel._forward (A/B: DM_WM_TAIL=0)
        self.wm_tail_on_side = os.environ.get('DM_WM_TAIL', '1') != '0'

    # ---- optimizers (dreamer.py:60-87)
    def param_groups(self):
        return dict(wm=list(self.wm.parameters()), probe=list(self.probe_model.parameters()),
                    actor=list(self.ac.actor.parameters()), critic=list(self.ac.critic.parameters()))

    def init_optimizers(self, lr, lr_actor=None, lr_critic=None, eps=1e-5):
        groups = self.param_groups()
        # the auxiliary ActorCritic's actor never receives a gradient (dreamer.py:347-358 keeps only loss_critic): torch's AdamW
        # skips such parameters entirely, so they are frozen here too (its critic_target has requires_grad=False already)
        frozen = list(self.wm.ac_aux.actor.parameters()) if self.wm.ac_aux is not None else []
        self._opt = dict(wm=FusedAdamW(groups['wm'], lr=lr, eps=eps, frozen=frozen), probe=FusedAdamW(groups['probe'], lr=lr, eps=eps),
                         actor=FusedAdamW(groups['actor'], lr=lr_actor or lr, eps=eps),
                         critic=FusedAdamW(groups['critic'], lr=lr_critic or lr, eps=eps))
        # the backward passes write straight into these optimizers' gradient buffers (see _flat_views)
        self.wm._fused, self.ac.actor._fused, self.ac.critic._fused = self._opt['wm'], self._opt['actor'], self._opt['critic']
        self.prepare_streams()
        if self.probe_gradients:      # dreamer.py:67-71: three optimizers; the probe head's parameters belong to none of them
            return self._opt['wm'], self._opt['actor'], self._opt['critic']
        return self._opt['wm'], self._opt['probe'], self._opt['actor'], self._opt['critic']

    def prepare_streams(self, device=None):
        """Creates the step's side streams and gives each its hardware queue NOW (_Overlap.__init__).  Called by init_optimizers()
        and by dist.attach(model=...); a data-parallel trainer must reach one of them BEFORE its first collective (a barrier and a
        parameter broadcast count): a communicator that exists before these streams have run anything slowed every later step by
        +11 ... +19 ms in round 6's measurement (profiles/r06_force_dp.txt, run J) - the process warns when it sees that order.
        No-op on the CPU, when the streams exist already, and inside a graph capture."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != 'cuda' or not self.overlap_backward or torch.cuda.is_current_stream_capturing():
            return
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self._overlap is None or self._overlap.s_wm.device != device:
            _warn_if_communicator_exists()
            self._overlap = _Overlap(device)

    def grad_clip(self, grad_clip, grad_clip_ac=None):
        if getattr(self, '_opt', None) is None:
            raise RuntimeError('call init_optimizers() before grad_clip(): clipping runs on the optimizers\' flat buffers')
        o = self._opt
        mb = getattr(self, 'metric_buffer', None)
        if mb is not None and mb.device != o['wm'].flat_grad.device:
            mb = None
        if mb is not None and o['actor'].home is not None:
            mb.record_stream(o['actor'].home)      # (pipelined mode: two of its slots are written on the actor-critic stream)
        out = lambda name: None if mb is None else mb[METRIC_SLOTS[name]:METRIC_SLOTS[name] + 2]   # [norm, clip coefficient]
        if self.probe_gradients:      # dreamer.py:81-86
            return dict(grad_norm=o['wm'].clip_grad_norm(grad_clip, out('grad_norm')),
                        grad_norm_actor=o['actor'].clip_grad_norm(grad_clip_ac or grad_clip, out('grad_norm_actor')),
                        grad_norm_critic=o['critic'].clip_grad_norm(grad_clip_ac or grad_clip, out('grad_norm_critic')))
        return dict(grad_norm=o['wm'].clip_grad_norm(grad_clip, out('grad_norm')),
                    grad_norm_probe=o['probe'].clip_grad_norm(grad_clip, out('grad_norm_probe')),
                    grad_norm_actor=o['actor'].clip_grad_norm(grad_clip_ac or grad_clip, out('grad_norm_actor')),
                    grad_norm_critic=o['critic'].clip_grad_norm(grad_clip_ac or grad_clip, out('grad_norm_critic')))

    # ---- pipelined actor / critic optimizer (round 4) --------------------------------------------------------------------
    # The four optimizer groups are independent (dreamer.py:60-71).  With `pipeline_ac_optimizer = True` the gradient hand-over,
    # clip and AdamW step of the ACTOR and CRITIC groups are enqueued on the actor-critic stream, behind their backward pass,
    # instead of on the caller's stream - which therefore no longer waits for the actor-critic backward before the next
    # step's encoder and posterior loop: the tail of step n (actor-critic backward, ~2.5 ms after the world-model backward has
    # finished, profiles/r04_queues_f32.txt) overlaps the head of step n+1.  Same kernels, same per-stream order: bit-identical
    # parameters (test_pipelined_ac_optimizer_is_bit_identical).  The caller's stream waits for those AdamW steps where it
    # first reads actor / critic parameters (the rollout), in inference() and state_dict(), and in packed_metrics() for the two
    # gradient norms.  OFF by default: a trainer that touches actor / critic `.grad`s or reads `grad_norm_actor / _critic` on its
    # own stream between backward() and step() (GradScaler.unscale_ with amp; `.item()` on grad_clip()'s dict) must leave it off
    # or call join_optimizers() first.  bench.py switches it on (its loop reads no per-step scalars).
    pipeline_ac_optimizer = False

    def _set_ac_home(self, ov):
        o = getattr(self, '_opt', None)
        if o is None:
            return
        on = bool(self.pipeline_ac_optimizer) and ov is not None and all(o[k].dp is None for k in ('actor', 'critic')) \
            and not torch.cuda.is_current_stream_capturing() and not self.wm._arena.on      # (the step arena re-uses buffers)
        for k in ('actor', 'critic'):
            if o[k].home is not None and not on:
                o[k].join()
            o[k].home = ov.s_ac if on else None

    def _ac_pipelined(self):
        o = getattr(self, '_opt', None)
        return o is not None and o['actor'].home is not None

    def _await_ac_optimizers(self):
        """Before the caller's stream first touches what the actor-critic stream may still be using - actor / critic parameters
        (AdamW), the imagined trajectory and the heads' activations of the previous step (its backward pass) - it waits for
        that stream.  By now it holds nothing but the previous step's tail."""
        if self._ac_pipelined():
            self.join_optimizers()

    def join_optimizers(self):
        """The current stream waits for everything the pipelined optimizer groups have enqueued on their own stream."""
        o = getattr(self, '_opt', None)
        if o is not None:
            for k in ('actor', 'critic'):
                o[k].join()

    def state_dict(self, *args, **kwargs):
        self.join_optimizers()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.join_optimizers()      # (pipelined mode: an actor / critic AdamW step may still be writing the parameters)
        return super().load_state_dict(*args, **kwargs)

    def check_device_status(self):
        """Raises - ONCE - if a persistent posterior kernel of this process has given up in a spin loop (csrc/rssm_lds.hip: it needs
        every workgroup resident at once; a second process on the GPU can starve it).  The flag is host-visible memory the kernel
        writes, so this costs nothing; it is meaningful behind a device synchronisation - packed_metrics_host() and the
        trainer's logging sync are where it is called.  The call ACKNOWLEDGES the flag (dm_rssm_lds_status_ack): the step that
        gave up produced garbage and must not be trusted, so the trainer restores its last checkpoint and goes on IN THIS
        PROCESS - the library keeps the persistent kernel switched off from then on (dm_rssm_lds_gave_up) and every later call
        runs the launch chain, and later logging calls do not raise again."""
        st = H.lib().dm_rssm_lds_status_ack()
        if st != 0:
            raise RuntimeError(f'a persistent RSSM kernel gave up in a spin loop (status {st}): the outputs of that training step '
                               'are invalid - restore the last checkpoint; later steps of this process run the launch chain '
                               '(the persistent kernel stays off) and this error is not raised again')

    def packed_metrics(self):
        """(names, buffer, idx): every loss / metric scalar of the last training_step() (+ the gradient norms once grad_clip()
        has run) sits in ONE 1-D device tensor `buffer`; `idx` (a python list) are the slots of `names` in it, so
        `vals = buffer.tolist(); dict(zip(names, (vals[i] for i in idx)))` replaces the trainer's ~20 `.item()` syncs per
        logged step (train.py:204-214) with a single device-to-host copy.  No kernel runs here: the kernels of the step
        wrote their results straight into this buffer."""
        self.join_optimizers()      # (pipelined mode: the actor / critic gradient norms were written on the actor-critic stream)
        names = list(METRIC_SLOTS)
        return names, self.metric_buffer, [METRIC_SLOTS[n] for n in names]

    def packed_metrics_host(self):
        """{name: float} of packed_metrics() with ONE device-to-host copy (= the step's only sync), followed by
        check_device_status() - the logging call of a trainer (train.py:204-214)."""
        names, buf, idx = self.packed_metrics()
        vals = buf.tolist()
        self.check_device_status()
        return {n: vals[i] for n, i in zip(names, idx)}

    def init_state(self, batch_size):
        return self.wm.init_state(batch_size)

    # ---- inference (dreamer.py:92-111)
    def inference(self, obs, in_state, noise=None):
        """`noise` (beyond the reference signature): dict(u_post (1,B,S)) of explicit uniforms for the posterior draw."""
        assert 'action' in obs, 'Observation should contain previous action'
        act_shape = obs['action'].shape
        assert len(act_shape) == 3 and act_shape[0] == 1, f'Expected shape (1,B,A), got {act_shape}'
        self.join_optimizers()
        features, out_state = self.wm.forward(obs, in_state, None if noise is None else noise['u_post'])
        B = act_shape[1]
        feat = features.reshape(B, -1)
        shp = self.wm.shape(1, B, 1)
        ws = self.wm.workspace(shp, feat.device)
        logits, _ = self.ac.actor.fwd(feat, feat.shape[1], B, ws, save_acts=False)
        value, _ = self.ac.critic.fwd(feat, feat.shape[1], B, ws, save_acts=False)
        action_distr = _torch_actor_distribution(self.ac.actor_dist, logits.view(1, B, -1))
        return action_distr, out_state, dict(policy_value=value.mean())

    # ---- imagination (dreamer.py:188-216)
    def dream(self, in_state, imag_horizon, dynamics_gradients=False, u_act=None, u_prior=None, _pack=None):
        if dynamics_gradients:
            raise NotImplementedError('actor_grad=dynamics is not built (and not runnable in the reference, SURVEY 0.5)')
        h, z = in_state
        _require_cuda(h, 'in_state')
        Hh = int(imag_horizon)
        start = torch.cat((h, z), -1).contiguous()           # to_feature (rssm.py:83-84)
        return self._dream_from_features(start, Hh, u_act, u_prior, _pack)

    def _dream_from_features(self, start, Hh, u_act=None, u_prior=None, _pack=None, _start_in_arena=False, early=None):
        """start: (M,F) rows [h|z] (the world model's feature matrix is passed as is, no concat copy)."""
        c = self.conf
        M, dev = start.shape[0], start.device
        F_, A, S = self.wm.features_dim, c.action_dim, c.stoch_dim
        shp = self.wm.shape(1, M, Hh)                        # T*B = M rows for workspace sizing
        ws = self.wm.workspace(shp, dev)
        kind = self.ac.dist_kind
        # the rollout chain's buffers live in the world model's step arena (stable addresses -> its hipGraph is replayed);
        # uniforms for the one-hot actor / the categorical latents, standard-normal noise for continuous actors / Gaussian latents
        ar = self.wm._arena
        ua_shape = (Hh, M) if kind == 0 else (Hh, M, A)
        if u_act is not None and tuple(u_act.shape) != ua_shape:
            raise ValueError(f'actor noise has shape {tuple(u_act.shape)}, expected {ua_shape}')
        if ar.on:
            ua = ar.get('u_act', ua_shape, device=dev)
            up = ar.get('u_prior', (Hh, M, S), device=dev)
            if u_act is not None:
                ua.copy_(u_act)
            elif kind == 0:
                ua.uniform_()
            else:
                ua.normal_()
            if u_prior is not None:
                up.copy_(u_prior.reshape(Hh, M, S))
            elif c.stoch_discrete:
                up.uniform_()
            else:
                up.normal_()
            u_act, u_prior = ua, up
        else:
            if u_act is None:
                u_act = torch.rand(ua_shape, device=dev) if kind == 0 else torch.randn(ua_shape, device=dev)
            if u_prior is None:
                u_prior = (torch.rand if c.stoch_discrete else torch.randn)(Hh, M, S, device=dev)
            u_act, u_prior = u_act.float().contiguous(), u_prior.reshape(Hh, M, S).float().contiguous()
        if ar.on and not _start_in_arena:
            start = ar.get('dream_start', tuple(start.shape), device=dev).copy_(start)
        start = start.contiguous()
        feats = ar.get('dream_feats', (Hh + 1, M, F_), device=dev)
        actions = ar.get('dream_actions', (Hh, M, A), device=dev)
        act_idx = ar.get('dream_act_idx', (Hh, M), torch.int32, device=dev)
        cell_p = H.rssm_struct(self.wm.core.cell.ordered())
        actor_p = self.ac.actor.struct()
        a_acts = a_logits = None
        if _pack is not None:          # training: keep the actor activations of all H steps for the policy-gradient backward
            a_acts = ar.get('dream_actor_acts', (self.ac.actor.acts_floats(Hh * M),), device=dev)
            a_logits = ar.get('dream_actor_logits', (Hh * M, self.ac.actor.out_dim), device=dev)
        # The heads over the imagined states (reward, terminal: dreamer.py:212-213; critic_target, critic: a2c.py:85,113) are
        # per-row work on 40 000 rows, 4 ms behind a rollout whose last steps leave the chip half idle (profiles/r03_queues_*):
        # they run as two row windows - steps [0, split) and [split, H] - and with `early` (the training step's side streams)
        # the first window runs on the idle actor-critic stream as soon as the library reports those rows final, while the
        # rollout finishes on this one.  Same windows, same kernels either way: bit-identical to the single-stream order.
        rows = (Hh + 1) * M
        r_split = self.ac.split_steps(Hh + 1) * M
        # (not under conf.amp: there the world-model backward chain on the other side stream is the step's critical path and the
        #  early window only takes CUs from it - measured 23.5 vs 23.2 ms; fp32: 37.1 vs 37.3 ms, 7-column shard 11.36 vs 11.50)
        use_early = (early is not None and _pack is not None and not ar.on and r_split < rows and _HEADS_EARLY
                     and not getattr(c, 'amp', False))
        if use_early:
            H.call('dm_dream_rollout_marks', 1, (ctypes.c_int * 1)(r_split // M - 2),
                   (ctypes.c_void_p * 1)(int(early.ev_mark.cuda_event)))
        H.call('dm_dream_rollout', ctypes.byref(shp), M, H.fptr(start), ctypes.byref(cell_p), ctypes.byref(actor_p),
               H.fptr(u_act), H.fptr(u_prior), H.fptr(feats), H.fptr(actions), H.ptr(act_idx),
               H.fptr(a_acts), H.fptr(a_logits), H.ptr(ws), ws.numel(), H.stream())
        f2 = feats.view(rows, F_)
        Zc = c.stoch_dim * c.stoch_discrete            # the sampled one-hot latent columns of a feature row
        mu, tl = torch.empty(rows, 1, device=dev), torch.empty(rows, 1, device=dev)
        values = self.ac.value_buffers(rows, dev) if _pack is not None else None

        def heads(a, b, wsx):
            if b <= a:
                return
            self.wm.decoder.reward.model.fwd(f2, F_, b - a, wsx, save_acts=False, sparse_cols=Zc, window=(rows, a), out=mu)
            self.wm.decoder.terminal.model.fwd(f2, F_, b - a, wsx, save_acts=False, sparse_cols=Zc, window=(rows, a), out=tl)
            if values is not None:
                self.ac.forward_values(f2, F_, rows, a, b - a, wsx, values)
        if use_early:
            early.s_ac.wait_event(early.ev_mark)
            with torch.cuda.stream(early.s_ac):
                heads(0, r_split, early.ws_ac)
                early.ev_heads.record(early.s_ac)
            heads(r_split, rows, ws)
            torch.cuda.current_stream().wait_event(early.ev_heads)
        else:
            heads(0, r_split, ws)
            heads(r_split, rows, ws)
        term = torch.empty(rows, device=dev)
        H.call('dm_head_loss', 1, rows, H.fptr(tl), None, 0.0, 0.0, None, None, H.fptr(term), H.stream())
        if _pack is not None:
            _pack.update(act_idx=act_idx, ws=ws, actor_acts=a_acts, actor_logits=a_logits, values=values)
        return feats, actions, _Mean(mu.view(Hh + 1, M)), _Mean(term.view(Hh + 1, M))

    # ---- training step (dreamer.py:113-186)
    def training_step(self, obs, in_state, iwae_samples=None, imag_horizon=None, do_open_loop=False, do_image_pred=False,
                      do_dream_tensors=False, noise=None, forced_idx=None):
        """Extra keyword arguments beyond the reference signature: `noise` = dict(u_post (T,B,S), u_act (H,M), u_prior
        (H,M,S)) of explicit uniforms for the three sampler call sites, `forced_idx` (T,B,S) to teacher-force the posterior."""
        for k in ('action', 'reward', 'reset', 'terminal'):
            assert k in obs, f'`{k}` required in observation'
        iwae_samples = int(iwae_samples or self.iwae_samples)
        imag_horizon = int(imag_horizon or self.imag_horizon)
        T, B = obs['action'].shape[:2]
        noise = noise or {}
        I = iwae_samples
        u_post = noise.get('u_post')
        if u_post is not None:
            u_post = u_post.reshape(T, B * I, -1)

        if self._overlap is not None:
            # pre-launched backward passes of a step whose losses were never backpropagated may still be reading the arena -
            # and their launcher jobs may still be ENQUEUING: a stream wait only orders what is already in the stream, so the
            # jobs are awaited first (their exceptions belong to the backward() that owns them, not to this step)
            stale = [getattr(self.wm, '_last_pack', None)] + list(getattr(self.ac, '_last_packs', None) or ())
            for old_pk in stale:
                fut = old_pk.get('pre') if isinstance(old_pk, dict) else None
                if fut is not None:
                    try:
                        fut.result()
                    except Exception:
                        pass
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._overlap.s_wm)
            if not self._ac_pipelined():          # (pipelined mode waits for the actor-critic stream where the rollout starts:
                cur.wait_stream(self._overlap.s_ac)      # nothing the world-model forward writes is read by that stream)
        # every loss / metric scalar of this step lands in ONE device buffer (METRIC_SLOTS)
        mbuf = torch.zeros(METRIC_BUF_FLOATS, device=obs['action'].device)
        self.metric_buffer = mbuf
        tail = None
        dev0 = obs['action'].device
        if (self.overlap_backward and torch.is_grad_enabled() and self.wm_tail_on_side and dev0.type == 'cuda'
                and not torch.cuda.is_current_stream_capturing()):
            if self._overlap is None or self._overlap.s_wm.device != dev0:
                self._overlap = _Overlap(dev0)
            tail = self._overlap
        loss_model, features, states, out_state, metrics, tensors = \
            self.wm.training_step(obs, in_state, iwae_samples=iwae_samples, do_open_loop=do_open_loop,
                                  do_image_pred=do_image_pred, u_post=u_post, forced_idx=forced_idx,
                                  imag_horizon=imag_horizon, u_pred=noise.get('u_pred'), mbuf=mbuf, _internal=True, _tail=tail)
        pk = self.wm._last_pack
        ov = None
        if not (self.overlap_backward and torch.is_grad_enabled()):
            self._set_ac_home(None)
        if self.overlap_backward and torch.is_grad_enabled():
            dev = pk['feat'].device
            if self._overlap is None or self._overlap.s_wm.device != dev:
                self._overlap = _Overlap(dev)
            ov = self._overlap
            pk['overlap'] = ov
            self._set_ac_home(ov)
            gens = {}
            for owner in (self.wm, self.ac.actor, self.ac.critic):      # main thread: zero_grad() - or a backward() on
                if getattr(owner, '_fused', None) is not None:          # an OLDER step's losses - may come before the
                    gens[id(owner)] = owner._fused.claim_scratch()      # launcher thread has touched the buffers
            # world-model backward: on its own stream and workspace, concurrent with everything below
            need = pk['ws'].numel()
            if ov.ws_wm is None or ov.ws_wm.numel() < need:
                ov.ws_wm = torch.empty(need, dtype=torch.uint8, device=dev)
            # (with the forward's tail already on s_wm the backward simply follows it in stream order)
            ov.ev_wm_fwd.record(ov.s_wm if pk.get('tail') is not None else torch.cuda.current_stream())
            pk['pre'] = ov.submit(ov.s_wm, ov.ev_wm_fwd, lambda: _prelaunched(self.wm, lambda: self.wm._backward(
                pk, ov.ws_wm, scratch=gens.get(id(self.wm), True), defer_wgrad=True)))
        metrics, tensors = dict(metrics), tensors.copy()          # LazyTensors.copy(): image_rec stays a thunk
        loss_probe, metrics_probe, tensors_probe = self.probe_model.training_step(features.detach(), obs)
        metrics.update(**metrics_probe)
        tensors.update(**tensors_probe)

        # (T,B,I) => (TBI): the feature matrix [h|z] of all posterior states, detached (dreamer.py:149)
        dpk = {}
        if ov is not None:
            need = 4 * int(H.lib().dm_mlp_ws_floats((imag_horizon + 1) * T * B * I, MLP_HIDDEN, 4))
            if ov.ws_ac is None or ov.ws_ac.numel() < need:
                ov.ws_ac = torch.empty(need, dtype=torch.uint8, device=pk['feat'].device)
        self._await_ac_optimizers()      # (pipelined mode: the previous step's actor / critic AdamW ran on the actor-critic stream)
        self.ac.begin_step()             # the target-network refresh comes before the first use of critic_target (a2c.py:76-79)
        features_dream, actions_dream, rewards_dream, terminals_dream = \
            self._dream_from_features(pk['feat'], imag_horizon,
                                      noise.get('u_act') if self.ac.dist_kind == 0 else noise.get('eps_act'),
                                      noise.get('u_prior'), _pack=dpk, _start_in_arena=True, early=ov)
        (loss_actor, loss_critic), metrics_ac, tensors_ac = \
            self.ac.training_step(features_dream, actions_dream, rewards_dream.mean, terminals_dream.mean,
                                  act_idx=dpk['act_idx'], ws=dpk['ws'], actor_acts=dpk['actor_acts'],
                                  actor_logits=dpk['actor_logits'], overlap=ov, mbuf=mbuf, values=dpk['values'], _begun=True)
        if ov is not None:
            ov.ev_fwd.record(torch.cuda.current_stream())
            for mlp, hp in zip((self.ac.actor, self.ac.critic), self.ac._last_packs):
                hp['pre'] = ov.submit(ov.s_ac, ov.ev_fwd, lambda mlp=mlp, hp=hp: _prelaunched(mlp, lambda: mlp.bwd(
                    hp['x'], hp['ldx'], hp['rows'], hp['acts'], hp['dout'], ov.ws_ac, scratch=gens.get(id(mlp), True))))
        metrics.update(**metrics_ac)
        if I == 1:
            tensors.update(policy_value=tensors_ac['value'][0].view(T, B))
        else:                        # unflatten_batch(value[0], (T,B,I)).mean(-1), dreamer.py:159
            pv = torch.empty(T, B, device=pk['feat'].device)
            H.call('dm_reduce_i', T * B, I, 1, H.fptr(tensors_ac['value'][0].contiguous()), 0, H.fptr(pv), None, H.stream())
            tensors.update(policy_value=pv)
        # Diagnostics for tests / debugging (not part of the reference API).  The index tensors are copies; `actions`,
        # `dream_features`, `post` and `prior` are views of the step arena: valid until the next training_step().
        self.last_extras = dict(post_idx=pk['idx'].view(T, B * I, -1).clone(), act_idx=dpk['act_idx'].clone(),
                                actions=actions_dream, dream_features=features_dream, actor_logits=dpk.get('actor_logits'),
                                ac_tensors=tensors_ac, post=pk['post'], prior=pk['prior'], pred_idx=pk.get('pred_idx'))
        # Dream for a log sample (dreamer.py:163-180): T-1 imagined steps from the B first states, decoded to images
        dream_tensors = {}
        if do_dream_tensors:
            with torch.no_grad():
                kind, dpk2 = self.ac.dist_kind, {}
                # states[0, :, 0] (dreamer.py:170): the first of the I samples of every batch column at t = 0 = rows b*I
                f2, a2, r2, t2 = self._dream_from_features(
                    pk['feat'][:B * I:I].contiguous(), T - 1,
                    noise.get('u_act_log') if kind == 0 else noise.get('eps_act_log'), noise.get('u_prior_log'), _pack=dpk2)
                dl = self.wm.decoder.image.layers()
                dec_p = H.conv_struct([m.weight for m in dl], [m.bias for m in dl])
                shp, dev = pk['shp_e'], pk['feat'].device          # (T,B,1): the log dream has one row per (step, column)
                acts = torch.empty(int(H.lib().dm_conv_decoder_acts_floats(ctypes.byref(shp))), device=dev)
                image_dream = torch.empty(T, B, self.conf.image_channels, self.conf.image_size, self.conf.image_size, device=dev)
                ws = self.wm.workspace(shp, dev)
                H.call('dm_conv_decoder_mse_fwd', ctypes.byref(shp), H.fptr(f2), self.wm.features_dim, H.ptr(pk['image']),
                       ctypes.byref(dec_p), H.fptr(acts), None, H.fptr(image_dream), H.ptr(ws), ws.numel(), H.stream())
                _, _, t_ac2 = self.ac.training_step(f2, a2, r2.mean, t2.mean, log_only=True, act_idx=dpk2['act_idx'],
                                                    ws=dpk2['ws'], actor_acts=dpk2['actor_acts'],
                                                    actor_logits=dpk2['actor_logits'], values=dpk2['values'])
                dream_tensors = dict(action_pred=torch.cat([obs['action'][:1].float(), a2]), reward_pred=r2.mean,
                                     terminal_pred=t2.mean, image_pred=image_dream.view(T, B, *image_dream.shape[-3:]),
                                     **t_ac2)
                self.last_extras.update(dream_log_act_idx=dpk2['act_idx'].clone())
        if pk.get('tail') is not None:      # losses, metrics and tensors of the world model were written on the world-model stream
            cur = torch.cuda.current_stream()
            cur.wait_event(pk['tail'].ev_tail)
            # ... and ALLOCATED from that stream's pool: what leaves training_step() is read on the caller's stream from here on, so the
            # caching allocator must not hand such a block to the next s_wm allocation while a caller-stream read is still queued
            # (the actor-critic outputs are guarded the same way)
            leaving = [loss_model, pk.get('dec_acts')] + [v for v in dict.values(pk['tensors']) if torch.is_tensor(v)]
            for t in leaving:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        if self.probe_gradients:      # dreamer.py:183-186
            losses = (loss_model + loss_probe, loss_actor, loss_critic)
        else:
            losses = (loss_model, loss_probe, loss_actor, loss_critic)
        return losses, out_state, metrics, tensors, dream_tensors

    def __str__(self):
        count = lambda m: sum(p.numel() for p in m.parameters())
        s = [f'Model: {count(self)} parameters']
        for sub in (self.wm.encoder, self.wm.decoder, self.wm.core, self.ac, self.probe_model):
            s.append(f'  {type(sub).__name__:<15}: {count(sub)} parameters')
        return '\n'.join(s)
