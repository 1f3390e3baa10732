"""GPU: replay reader -> device ring -> training_step.

The ring (pinned staging, copies on the consumer's stream - staged by next() or ahead of time by prefetch() -, slot recycling)
must hand the step exactly the bytes the
reader produced: several optimizer steps fed by `DeviceRing` are bit-identical to the same steps fed by plain
synchronous `.to(device)` copies of an identically seeded reader - with more steps than ring slots, so slots recycle
while earlier steps may still be running."""
import numpy as np
import pytest
import torch

from oracle import dreamer_oracle as O
from pydreamer_amd import replay as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _write_episodes(tmp_path, action_dim):
    rs = np.random.RandomState(4)
    repo = R.LocalEpisodeRepository(str(tmp_path))
    for ep, n in enumerate([31, 44, 27]):
        d = dict(image=rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8), action=rs.randint(0, action_dim, n),
                 reward=rs.randn(n).astype(np.float32), terminal=np.zeros(n, bool), reset=np.zeros(n, bool))
        d['terminal'][-1] = True
        repo.save_data(d, ep, ep)
    return repo


def _source(repo, oconf, seed):
    for b in R.SequentialReplay(repo, oconf.batch_length, oconf.batch_size, allow_mid_reset=True, seed=seed):
        yield R.preprocess_batch(b, oconf.action_dim, clip_rewards='tanh')


def _train(model, conf, batches, noises, nsteps, after_step=None):
    opts = model.init_optimizers(conf.adam_lr, conf.adam_lr_actor, conf.adam_lr_critic, conf.adam_eps)
    state = model.init_state(conf.batch_size)
    out = []
    for s in range(nsteps):
        obs = batches()
        assert obs['image'].dtype == torch.uint8 and obs['image'].is_cuda
        losses, state, metrics, tensors, _ = model.training_step(obs, state, noise=noises[s])
        if after_step is not None:
            after_step()
        for opt in opts:
            opt.zero_grad()
        for loss in losses:
            loss.backward()
        model.grad_clip(conf.grad_clip, conf.grad_clip_ac)
        for opt in opts:
            opt.step()
        out.append(torch.stack([l.detach().reshape(()) for l in losses]))
    return torch.stack(out).cpu(), opts[0].flat_param.clone().cpu()


def test_device_ring_feeds_training_step(hip, tmp_path):
    from tests.test_gpu_training_step import _build, _hip_conf
    oconf = O.tiny_conf()
    conf = _hip_conf(oconf)
    repo = _write_episodes(tmp_path, oconf.action_dim)
    nsteps = 9                                                 # > 2 * ring depth: every slot is recycled
    noises = [{k: v.to(DEV) for k, v in O.make_noise(oconf, seed=100 + s).items()} for s in range(nsteps)]

    ring = R.DeviceRing(_source(repo, oconf, seed=7), DEV, depth=3)
    a = _train(_build(oconf, O.make_params(oconf, seed=2)), conf, ring.next, noises, nsteps)
    ring.close()
    # the trainer's order: the next batch is staged right after training_step() returned, while the backward passes run
    ring = R.DeviceRing(_source(repo, oconf, seed=7), DEV, depth=3)
    c = _train(_build(oconf, O.make_params(oconf, seed=2)), conf, ring.next, noises, nsteps, after_step=ring.prefetch)
    ring.close()
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])

    # the in-place path: the planner writes the frame windows straight into the ring's pinned slots (ReplayFeed)
    feed = R.ReplayFeed(R.SequentialReplay(repo, oconf.batch_length, oconf.batch_size, allow_mid_reset=True, seed=7), oconf.action_dim,
                        clip_rewards='tanh')
    ring = R.DeviceRing(feed, DEV, depth=3)
    d = _train(_build(oconf, O.make_params(oconf, seed=2)), conf, ring.next, noises, nsteps, after_step=ring.prefetch)
    ring.close()
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])

    it = _source(repo, oconf, seed=7)
    b = _train(_build(oconf, O.make_params(oconf, seed=2)), conf,
               lambda: {k: torch.from_numpy(v).to(DEV) for k, v in next(it).items()}, noises, nsteps)
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
