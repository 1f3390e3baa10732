"""Flat-buffer AdamW + global-norm clipping on HIP kernels (reference: Dreamer.init_optimizers / grad_clip,
dreamer.py:60-87 = torch.optim.AdamW(lr, eps; betas (0.9,0.999), weight_decay 0.01 defaults) + clip_grad_norm_).

Each optimizer group (wm / probe / actor / critic) owns ONE contiguous fp32 parameter buffer, one gradient buffer and
two moment buffers; the nn.Parameters become views into the parameter buffer and their `.grad`s views into the gradient
buffer, so that
  * the norm, the in-place clip and the AdamW update are three streaming kernels per group instead of ~100 small ones;
  * data-parallel training all-reduces one buffer per group over RCCL (see pydreamer_amd/dist.py).
"""
import ctypes

import torch

from . import hip as H


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.01):
        params = list(params)
        if not params:
            raise ValueError('FusedAdamW got an empty parameter list')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plist = params
        dev = params[0].device
        if dev.type != 'cuda':
            raise H.DreamerHipError(f'FusedAdamW needs parameters on a gfx950 device, got {dev} '
                                    f'(call model.to(device) before init_optimizers; there is no CPU optimizer path)')
        # every parameter starts on a 256-byte boundary of the flat buffer: the GEMM operand loaders take their 16-byte
        # path only for 16-byte aligned weights, and one 1-element bias (reward head) would otherwise misalign every
        # tensor behind it (all RSSM weights).  Pad slots stay exactly zero (zero grad, zero moments, decay of zero).
        self._offsets, n = [], 0
        for p in params:
            self._offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev)
        self.flat_grad = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.norm_buf = torch.zeros(2, device=dev)        # [total_norm, clip_coef]
        self._ws = torch.empty(4096, device=dev)
        self.step_count = 0
        self.dp = None                                    # set by dist.attach(): (process_group, weight)
        self.fresh = False                                # flat_grad is all zero and nobody has written to it yet
        self.scratch = None                               # same layout as flat_grad, for pre-launched backward passes
        self.scratch_gen = 0                              # bumped every time the scratch buffer is handed out
        self._ids = {id(p) for p in params}
        with torch.no_grad():
            for p, off in zip(params, self._offsets):
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)

    def _grads_are_views(self):
        for p, off in zip(self._plist, self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                return False
        return True

    def _regather(self):
        """A caller replaced `.grad` (e.g. zero_grad(set_to_none=True) elsewhere): copy into the flat buffer and re-view."""
        with torch.no_grad():
            for p, off in zip(self._plist, self._offsets):
                k = p.numel()
                dst = self.flat_grad[off:off + k]
                if p.grad is None:
                    dst.zero_()
                elif p.grad.data_ptr() != dst.data_ptr():
                    dst.copy_(p.grad.reshape(-1))
                p.grad = dst.view(p.shape)

    def zero_grad(self, set_to_none=False):
        """Zeroes the flat gradient buffer and keeps the `.grad` views alive (set_to_none is ignored by design)."""
        if not self._grads_are_views():
            self._regather()
        self.flat_grad.zero_()
        self.fresh = True

    def scratch_views(self, plist):
        """Per-parameter views of the persistent scratch gradient buffer (pad slots stay zero forever), or None if
        `plist` is not exactly this group's parameter set."""
        if len(plist) != len(self._plist) or any(id(p) not in self._ids for p in plist):
            return None
        if self.scratch is None:
            self.scratch = torch.zeros_like(self.flat_grad)
            self._off_of = {id(p): off for p, off in zip(self._plist, self._offsets)}
        self.scratch_gen += 1
        return [self.scratch[self._off_of[id(p)]:self._off_of[id(p)] + p.numel()].view(p.shape) for p in plist]

    def claim_fresh_grads(self, plist):
        """For a backward pass that produces the gradients of EXACTLY this group's parameters with '=' semantics: if the
        buffer was zeroed since the last write, hand out the `.grad` views to be written in place (once); else None."""
        if not self.fresh or len(plist) != len(self._plist) or any(id(p) not in self._ids for p in plist):
            return None
        if not self._grads_are_views():
            return None
        self.fresh = False
        return [p.grad for p in plist]

    def clip_grad_norm(self, max_norm):
        """clip_grad_norm_ on the flat buffer: returns the pre-clip total norm as a 0-d device tensor (no host sync)."""
        if not self._grads_are_views():
            self._regather()
        if self.dp is not None:
            from . import dist as D
            D.allreduce_grads(self)
        H.call('dm_multi_tensor_norm_clip', H.fptr(self.flat_grad), self.numel, float(max_norm), H.fptr(self.norm_buf),
               H.fptr(self._ws), self._ws.numel() * 4, H.stream())
        H.call('dm_scale_inplace', H.fptr(self.flat_grad), self.numel, ctypes.c_void_p(self.norm_buf.data_ptr() + 4),
               H.stream())
        return self.norm_buf[0].clone()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not used by the trainer section (train.py:193-198)')
        if not self._grads_are_views():
            self._regather()
        g = self.param_groups[0]
        self.step_count += 1
        H.call('dm_adamw_step', H.fptr(self.flat_param), H.fptr(self.flat_grad), H.fptr(self.exp_avg),
               H.fptr(self.exp_avg_sq), self.numel, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'],
               self.step_count, None, H.stream())

    # checkpoint format: same top-level keys as torch optimizers ('state', 'param_groups') with flat moments
    def state_dict(self):
        return dict(state=dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone()),
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != 'params'}])

    def load_state_dict(self, sd):
        st = sd['state']
        self.step_count = int(st['step'])
        self.exp_avg.copy_(st['exp_avg'])
        self.exp_avg_sq.copy_(st['exp_avg_sq'])
        for k, v in sd['param_groups'][0].items():
            self.param_groups[0][k] = v
