"""Flat-buffer AdamW + global-norm clipping on HIP kernels (reference: Dreamer.init_optimizers / grad_clip,
dreamer.py:60-87 = torch.optim.AdamW(lr, eps; betas (0.9,0.999), weight_decay 0.01 defaults) + clip_grad_norm_).

Each optimizer group (wm / probe / actor / critic) owns ONE contiguous fp32 parameter buffer, one gradient buffer and
two moment buffers; the nn.Parameters become views into the parameter buffer and their `.grad`s views into the gradient
buffer, so that
  * the norm, the in-place clip and the AdamW update are three streaming kernels per group instead of ~100 small ones;
  * data-parallel training all-reduces one buffer per group over RCCL (see pydreamer_amd/dist.py).
"""
import contextlib
import ctypes

import torch

from . import hip as H


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.01, frozen=()):
        """frozen: parameters of this group that never receive a gradient (requires_grad=False, or sub-networks whose loss
        is not part of the group's objective - the auxiliary ActorCritic's actor, dreamer.py:267-279).  torch.optim.AdamW
        skips parameters whose .grad is None entirely (no moment update AND no weight decay); so does step() here."""
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
        self.fresh = False                                # flat_grad is (logically) all zero and nobody has written to it yet
        self._lazy_zero = False                           # ... but the memset was skipped (see zero_grad)
        self.scratch = None                               # the OTHER gradient buffer: pre-launched backward passes write here
        self.scratch_gen = 0                              # bumped every time the scratch buffer is handed out
        self._pending = False                             # a pre-launched backward result sits in `scratch`, not yet handed over
        self.early_reduce = None                          # (work handle) all-reduce of `scratch` already in flight (dist.py)
        self._reduced = False                             # flat_grad already holds the all-reduced gradient of this step
        # Pipelined mode (models.Dreamer.pipeline_ac_optimizer): the stream this group's backward pass ran on.  The gradient
        # hand-over, the clip and the AdamW step of the group are then enqueued THERE instead of on the caller's stream, so the
        # caller's stream does not wait for that backward pass before the next step's forward; `done` is recorded behind
        # step() and awaited by whoever reads the parameters next on another stream.
        self.home = None
        self.done = None
        self._gl_event = None
        self._ids = {id(p) for p in params}
        fz = {id(p) for p in frozen} | {id(p) for p in params if not p.requires_grad}
        self._frozen_idx = {i for i, p in enumerate(params) if id(p) in fz}
        self._spans, start = [], None          # maximal runs [begin, end) of the flat buffer that step() updates
        for i, (p, off) in enumerate(zip(params, self._offsets)):
            end = self._offsets[i + 1] if i + 1 < len(params) else n
            if i in self._frozen_idx:
                if start is not None:
                    self._spans.append((start, off))
                    start = None
            elif start is None:
                start = off
            _ = end
        if start is not None:
            self._spans.append((start, n))
        with torch.no_grad():
            for p, off in zip(params, self._offsets):
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)
        self._grad_views = {}                             # buffer data_ptr -> per-parameter views of that buffer
        self._point_grads(self.flat_grad)

    def _on_home(self):
        return torch.cuda.stream(self.home) if self.home is not None else contextlib.nullcontext()

    def order_after_current(self):
        """Pipelined mode: what the caller's stream has enqueued so far (e.g. the incoming scalar gradient of backward())
        happens before what this group enqueues next on its home stream."""
        if self.home is not None:
            if self._gl_event is None:
                self._gl_event = torch.cuda.Event()
            self._gl_event.record(torch.cuda.current_stream())
            self.home.wait_event(self._gl_event)

    def join(self):
        """The current stream waits for everything this group has enqueued on its home stream (no-op outside pipelined mode)."""
        if self.home is not None:
            torch.cuda.current_stream().wait_stream(self.home)

    def _views_of(self, buf):
        v = self._grad_views.get(buf.data_ptr())
        if v is None:
            v = [buf[off:off + p.numel()].view(p.shape) for p, off in zip(self._plist, self._offsets)]
            self._grad_views[buf.data_ptr()] = v
        return v

    def _point_grads(self, buf):
        for p, g in zip(self._plist, self._views_of(buf)):
            p.grad = g

    def _check_params_are_views(self):
        """anything that re-homes the parameters after init_optimizers() (model.to(...), .float(),
        load_state_dict(assign=True), a second init_optimizers()) would leave step() updating a buffer nobody reads."""
        base = self.flat_param.data_ptr()
        for p, off in zip(self._plist, self._offsets):
            if p.data_ptr() != base + 4 * off:
                raise H.DreamerHipError('a parameter no longer aliases the optimizer\'s flat buffer (moved / cast / re-assigned '
                                        'after init_optimizers()): call init_optimizers() again after moving the model')

    def _ensure_zeroed(self):
        if self._lazy_zero:
            self.flat_grad.zero_()
            self._lazy_zero = False

    def _grads_are_views(self):
        for p, off in zip(self._plist, self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                return False
        return True

    def _regather(self):
        """A caller replaced `.grad` (e.g. zero_grad(set_to_none=True) elsewhere): copy into the flat buffer and re-view."""
        self._ensure_zeroed()
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
        """Zeroes the flat gradient buffer and keeps the `.grad` views alive (set_to_none is ignored by design: the kernels
        write into these views).  When a pre-launched backward result is waiting in the other buffer (the trainer's order is
        training_step -> zero_grad -> backward, train.py:171-192) nothing is written at all: backward() will SWAP the two
        buffers, so the 101 MB memset and the 2 x 101 MB hand-over copy of round 1 are gone; the memset is done lazily if
        somebody needs real zeros first (clip / step / another backward path)."""
        self._check_params_are_views()
        if not self._grads_are_views():
            self._regather()
        if self._pending and not torch.cuda.is_current_stream_capturing():
            self._lazy_zero = True
        else:
            with self._on_home():
                self.flat_grad.zero_()
            self._lazy_zero = False
        self.fresh = True
        self._reduced = False

    def claim_scratch(self):
        """Called by training_step() on the CALLER's thread before the backward pass is handed to the launcher thread:
        the generation is bumped here, synchronously, so a backward() on the losses of an older training_step() is refused
        deterministically (the launcher thread may not have touched the buffer yet).  Returns the generation claimed."""
        self.scratch_gen += 1
        self._pending = True
        return self.scratch_gen

    def scratch_views(self, plist, bump=True):
        """Per-parameter views of the other gradient buffer (pad slots stay zero forever), or None if `plist` is not
        exactly this group's parameter set.  bump=False: the generation was claimed by claim_scratch()."""
        if len(plist) != len(self._plist) or any(id(p) not in self._ids for p in plist):
            return None
        if self.scratch is None:
            self.scratch = torch.zeros_like(self.flat_grad)
        if bump:
            self.scratch_gen += 1
        self._pending = True
        self.early_reduce = None
        by_id = {id(p): v for p, v in zip(self._plist, self._views_of(self.scratch))}
        return [by_id[id(p)] for p in plist]

    def adopt_scratch(self, gl):
        """Hand-over of a pre-launched backward result (called from loss.backward()): with '=' semantics (zero_grad since the
        last write) the two buffers swap roles - no copy; otherwise (gradient accumulation) it is added.  `gl` is the
        incoming scalar gradient as a 1-element device tensor (1.0 unless a GradScaler is active): the in-place scale
        kernel returns at once when it is exactly 1."""
        with self._on_home():
            self._adopt_scratch(gl)

    def _adopt_scratch(self, gl):
        self._pending = False
        work, self.early_reduce = self.early_reduce, None
        if work is not None:
            work.wait()                                   # the all-reduce of `scratch` issued right after its backward
        if self.fresh and not torch.cuda.is_current_stream_capturing():
            self.flat_grad, self.scratch = self.scratch, self.flat_grad
            self._point_grads(self.flat_grad)
            self._lazy_zero = False
            H.call('dm_scale_inplace', H.fptr(self.flat_grad), self.numel, H.fptr(gl), H.stream())
            self._reduced = work is not None
        elif self.fresh:                                  # inside a graph capture: fixed addresses, so copy
            self._ensure_zeroed()
            torch.mul(self.scratch, gl, out=self.flat_grad)
        else:                                             # accumulation onto gradients that are already there
            self._ensure_zeroed()
            if work is not None and self.dp is not None and not self._reduced:
                from . import dist as D
                D.allreduce_grads(self)                   # what is there has not been reduced yet; the addend has
                self._reduced = True
            self.flat_grad.addcmul_(self.scratch, gl.expand_as(self.scratch))
        self.fresh = False

    def claim_fresh_grads(self, plist):
        """For a backward pass that produces the gradients of EXACTLY this group's parameters with '=' semantics: if the
        buffer was zeroed since the last write, hand out the `.grad` views to be written in place (once); else None."""
        if not self.fresh or len(plist) != len(self._plist) or any(id(p) not in self._ids for p in plist):
            return None
        if not self._grads_are_views():
            return None
        self._ensure_zeroed()
        self.fresh = False
        return [p.grad for p in plist]

    def clip_grad_norm(self, max_norm, out=None):
        """clip_grad_norm_ on the flat buffer: returns the pre-clip total norm as a 0-d device tensor (no host sync).
        out: optional 2-float device slice receiving [norm, clip coefficient] (the step's metric buffer)."""
        with self._on_home():
            return self._clip_grad_norm(max_norm, out)

    def _clip_grad_norm(self, max_norm, out):
        if not self._grads_are_views():
            self._regather()
        self._ensure_zeroed()
        if self.dp is not None and not self._reduced:
            from . import dist as D
            D.allreduce_grads(self)
        self._reduced = False
        nb = self.norm_buf if out is None else out
        H.call('dm_multi_tensor_norm_clip', H.fptr(self.flat_grad), self.numel, float(max_norm), H.fptr(nb),
               H.fptr(self._ws), self._ws.numel() * 4, H.stream())
        H.call('dm_scale_inplace', H.fptr(self.flat_grad), self.numel, ctypes.c_void_p(nb.data_ptr() + 4), H.stream())
        return nb[0].clone() if out is None else nb[0]

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not used by the trainer section (train.py:193-198)')
        with self._on_home():
            self._step()
            if self.home is not None:
                if self.done is None:
                    self.done = torch.cuda.Event()
                self.done.record(self.home)

    def _step(self):
        self._check_params_are_views()
        if not self._grads_are_views():
            self._regather()
        self._ensure_zeroed()
        g = self.param_groups[0]
        self.step_count += 1
        for b, e in self._spans:
            o = 4 * b
            H.call('dm_adamw_step', ctypes.c_void_p(self.flat_param.data_ptr() + o), ctypes.c_void_p(self.flat_grad.data_ptr() + o),
                   ctypes.c_void_p(self.exp_avg.data_ptr() + o), ctypes.c_void_p(self.exp_avg_sq.data_ptr() + o), e - b, g['lr'],
                   g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.step_count, None, H.stream())

    # ---- checkpoint format: EXACTLY torch.optim.AdamW's (per-parameter 'state' entries, 'params' index lists), so that
    # tools.mlflow_save_checkpoint / mlflow_load_checkpoint (tools.py:164-197) move `optimizer_{i}_state_dict` between a
    # reference run and this build in both directions.  The flat moments are sliced at the parameter offsets.
    def state_dict(self):
        self.join()       # (pipelined mode: the moments may still be being written on the group's own stream)
        g = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        for k, v in dict(amsgrad=False, foreach=None, maximize=False, capturable=False, differentiable=False, fused=None).items():
            g.setdefault(k, v)
        g['params'] = list(range(len(self._plist)))
        state = {}
        if self.step_count > 0:                      # torch creates the per-parameter state lazily at the first step
            for i, (p, off) in enumerate(zip(self._plist, self._offsets)):
                if i in self._frozen_idx:         # torch keeps no state for parameters that never had a gradient
                    continue
                k = p.numel()
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[off:off + k].view(p.shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[off:off + k].view(p.shape).clone())
        return dict(state=state, param_groups=[g])

    def load_state_dict(self, sd):
        self.join()       # an AdamW step in flight on the group's own stream must not race with the copies below
        groups = sd['param_groups']
        if len(groups) != 1:
            raise ValueError(f'FusedAdamW holds one parameter group, the state dict has {len(groups)}')
        st = sd['state']
        if 'exp_avg' in st and not any(isinstance(k, int) for k in st):      # round-1 flat format of this package
            self.step_count = int(st['step'])
            self.exp_avg.copy_(st['exp_avg'])
            self.exp_avg_sq.copy_(st['exp_avg_sq'])
        else:
            ids = groups[0].get('params', list(range(len(self._plist))))
            if len(ids) != len(self._plist):
                raise ValueError(f'optimizer state has {len(ids)} parameters, this group has {len(self._plist)}')
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = set()
            for pid, p, off in zip(ids, self._plist, self._offsets):
                e = st.get(pid, st.get(str(pid)))
                if e is None:
                    continue
                if tuple(e['exp_avg'].shape) != tuple(p.shape):
                    raise ValueError(f'optimizer state entry {pid} has shape {tuple(e["exp_avg"].shape)}, parameter has {tuple(p.shape)}')
                k = p.numel()
                self.exp_avg[off:off + k].copy_(e['exp_avg'].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(e['exp_avg_sq'].reshape(-1))
                steps.add(int(float(e['step'])))
            if len(steps) > 1:
                raise ValueError(f'per-parameter step counts differ ({sorted(steps)}); one flat AdamW step count is kept')
            self.step_count = steps.pop() if steps else 0
        for k, v in groups[0].items():
            if k != 'params':
                self.param_groups[0][k] = v
