"""hipGraph capture of the trainer's inner section (reference train.py:171-192: training_step -> zero_grad -> backward x4).

One Atari-literal gradient step is ~1 000 kernel launches, a quarter of them the 50-row products of the sequential T-step,
BPTT and imagination chains.  Capturing the section once and replaying it removes the host from the loop; what is captured is
exactly the eager code path (same kernels, same order, same side streams for the overlapped backward passes), so results are
bit-identical to the eager step.  Measured on MI355X / ROCm 7.2 it is SLOWER than eager (a graph with concurrent branches
replays at ~11 us per node) and therefore optional; the library-level alternative - linear graphs of the three
launch chains only, csrc/chain_graph.hip - is GPU-neutral and saves host time only.

Left outside the graph on purpose:
  * the critic-target refresh (a2c.py:68-70: every `target_interval` steps, host-side condition) - done eagerly by
    `__call__` before the replay;
  * grad-clip + AdamW (train.py:193-198) - a dozen streaming launches; keeping them eager keeps the data-parallel
    all-reduce (pydreamer_amd/dist.py), the step counter and learning-rate changes out of the captured graph.

Static-shape contract: `obs` tensors and `in_state` of every call must have the shapes / dtypes of the example given at
construction (they are copied into the graph's input buffers).  The returned losses / metrics / tensors / out_state are
the graph's output buffers: they are overwritten by the next call, clone what must survive it.
"""
import torch

from . import hip as H


class GraphedTrainStep:
    def __init__(self, model, optimizers, obs, in_state, noise=None, warmup_iters=2, **step_kwargs):
        dev = obs['action'].device
        if dev.type != 'cuda':
            raise H.DreamerHipError('GraphedTrainStep needs the batch on a gfx950 device (there is no CPU path)')
        if model.wm.ac_aux is not None:
            raise NotImplementedError('aux_critic refreshes its target network on a host-side step counter inside '
                                      'training_step(); capture would freeze that decision - run it eagerly')
        self.model, self.optimizers = model, [o for o in optimizers]
        self.static_obs = {k: v.clone() for k, v in obs.items()}
        self.static_state = tuple(x.clone() for x in in_state)
        self.static_noise = None if noise is None else {k: v.clone() for k, v in noise.items()}
        self.step_kwargs = dict(step_kwargs)
        ac = model.ac

        def section():
            out = model.training_step(self.static_obs, self.static_state, noise=self.static_noise, **self.step_kwargs)
            for opt in self.optimizers:
                opt.zero_grad()
            for loss in out[0]:
                loss.backward()
            return out

        # warm-up on a side stream (lazy workspaces, side streams and the caching allocator settle before capture)
        saved_steps = ac.train_steps
        ac.defer_target_update = True
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, int(warmup_iters))):
                    section()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                out = section()
        finally:
            ac.defer_target_update = False
            ac.train_steps = saved_steps
        losses, self.out_state, self.metrics, self.tensors, self.extra = out
        self.losses = tuple(x.detach() for x in losses)
        # the captured graph bakes in raw pointers of lazily-grown buffers it does not own (workspaces of the
        # world model and of the overlap streams, the optimizers' second gradient buffer).  Holding references here means a
        # later, larger eager call allocates NEW buffers instead of freeing the captured ones under the graph's feet.
        ov = model._overlap
        self._keepalive = [model.wm._ws, getattr(ov, 'ws_wm', None), getattr(ov, 'ws_ac', None),
                           (model.wm._pipe or {}).get('ws_chain'), (model.wm._pipe or {}).get('ws_dec'),
                           [(o.flat_grad, o.scratch) for o in self.optimizers]]

    def __call__(self, obs, in_state, noise=None):
        """Same return value as Dreamer.training_step(); the four backward passes have already run (gradients are in the
        optimizers' flat buffers), so the caller continues with grad_clip() and optimizer.step()."""
        for k, dst in self.static_obs.items():
            src = obs[k]
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f'obs[{k!r}] is {tuple(src.shape)} {src.dtype}, the graph was captured for '
                                 f'{tuple(dst.shape)} {dst.dtype}')
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        for dst, src in zip(self.static_state, in_state):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.static_noise is not None:
            if noise is None:
                raise ValueError('this graph was captured with explicit sampler noise: pass `noise` on every call')
            missing = [k for k in self.static_noise if k not in noise]
            if missing:
                raise ValueError(f'noise is missing {missing} (the graph was captured with {sorted(self.static_noise)})')
            for k, dst in self.static_noise.items():
                if noise[k].dtype != dst.dtype:
                    raise ValueError(f'noise[{k!r}] is {noise[k].dtype}, the graph was captured with {dst.dtype}')
                dst.copy_(noise[k].reshape(dst.shape), non_blocking=True)
        elif noise is not None:
            raise ValueError('this graph was captured WITHOUT explicit sampler noise (it draws its own): `noise` would be '
                             'silently ignored - capture with an example `noise` dict to supply it per call')
        ac = self.model.ac
        if ac.train_steps % ac.target_interval == 0:        # a2c.py:68-70, host-side condition kept out of the graph
            ac.update_critic_target()
        ac.train_steps += 1
        self.graph.replay()
        tensors = self.tensors.copy() if hasattr(self.tensors, 'lazy') else self.tensors    # lazy entries re-read the new step
        return self.losses, self.out_state, self.metrics, tensors, self.extra
