"""CPU (-m "not gpu"): the replay reader (pydreamer_amd/replay.py; reference data.py:128-304, preprocessing.py:87-180).

The reference's data.py cannot be imported here (mlflow is absent), so these are property tests of the restated
algorithm ("parity unpinned" for this file): truncated-BPTT continuity per batch column, resets at file starts, random
resets only at window starts, partial-window carry, HWCT episodes, file-name grammar, preprocessing outputs."""
import os

import numpy as np
import pytest

from pydreamer_amd import replay as R


def _episode(n, ep, rs, transposed=False, action_dim=4):
    img = rs.randint(0, 256, (n, 8, 8, 3)).astype(np.uint8)
    d = dict(action=rs.randint(0, action_dim, n), reward=(ep * 1000 + np.arange(n)).astype(np.float32),      # reward encodes (episode, step)
             terminal=np.zeros(n, bool), reset=np.zeros(n, bool))
    d['terminal'][-1] = True
    if transposed:
        d['image_t'] = img.transpose(1, 2, 3, 0)          # THWC -> HWCT, generator.py:246-249
    else:
        d['image'] = img
    return d, img


@pytest.fixture
def repo(tmp_path):
    rs = np.random.RandomState(0)
    r = R.LocalEpisodeRepository(str(tmp_path))
    imgs = {}
    for ep, n in enumerate([37, 50, 23, 64]):
        d, img = _episode(n, ep + 1, rs, transposed=(ep % 2 == 1))
        path = r.save_data(d, ep, ep)
        imgs[ep + 1] = img
        assert R.parse_episode_name(path)[:2] == (ep, ep)
    return r, imgs


def test_file_name_grammar():
    assert R.parse_episode_name('ep000012_000014-r35-0421.npz') == (12, 14, 421)
    assert R.parse_episode_name('x/y/ep000007-r-3-0099.npz') == (7, 7, 99)
    assert R.parse_episode_name('20210101T000000-0500.npz') == (0, 0, 500)


@pytest.mark.parametrize('allow_mid_reset', [False, True])
def test_tbtt_windows_are_contiguous(repo, allow_mid_reset):
    r, imgs = repo
    T, B = 10, 3
    ds = R.SequentialReplay(r, T, B, skip_first=True, allow_mid_reset=allow_mid_reset, seed=1)
    it = iter(ds)
    prev = None
    seen_carry = False
    for step in range(60):
        b = next(it)
        assert b['reward'].shape == (T, B) and b['image'].shape == (T, B, 8, 8, 3) and b['image'].dtype == np.uint8
        for col in range(B):
            rew, reset = b['reward'][:, col], b['reset'][:, col]
            ep, st = (rew // 1000).astype(int), (rew % 1000).astype(int)
            for t in range(T):
                if reset[t]:
                    assert rew[t] == 0.0                               # a file starts with reset and zero reward
                    continue
                if t > 0 and not reset[t]:
                    # inside a window steps are consecutive steps of ONE episode
                    if not reset[t - 1]:
                        assert ep[t] == ep[t - 1] and st[t] == st[t - 1] + 1
                # frames belong to the (episode, step) the reward encodes (also for HWCT files)
                assert np.array_equal(b['image'][t, col], imgs[ep[t]][st[t]])
            if not allow_mid_reset:
                assert not reset[1:].any()                              # windows never straddle files
            elif reset[1:].any():
                seen_carry = True
            # continuity ACROSS windows of the same column: either the episode continues or the window starts with a reset
            if prev is not None and not reset[0]:
                pr = prev['reward'][-1, col]
                if prev['reset'][-1, col]:                              # the previous window ended ON a file start (step 0, reward zeroed)
                    assert st[0] == 1
                else:
                    assert ep[0] == int(pr // 1000) and st[0] == int(pr % 1000) + 1
        prev = b
    assert seen_carry == allow_mid_reset
    # action_next is the action of the following step, zero at the end of a file (data.py:246)
    assert b['action_next'].shape == b['action'].shape


def test_random_resets_only_at_window_starts(repo):
    r, _ = repo
    ds = R.SequentialReplay(r, 5, 2, skip_first=False, reset_interval=10, seed=3)
    it = iter(ds)
    extra = 0
    for _ in range(80):
        b = next(it)
        assert not b['reset'][1:].any()
        extra += int((b['reset'][0] & (b['reward'][0] != 0)).sum())     # a reset that is not a file start
    assert extra > 0


def test_buffer_size_keeps_newest_files(repo):
    r, _ = repo
    ds = R.SequentialReplay(r, 10, 1, buffer_size=70, seed=0)
    assert [f.episode_to for f in ds.files] == [3]                      # newest first: 64 steps fit, 64 + 23 does not
    ds = R.SequentialReplay(r, 10, 1, buffer_size=100, seed=0)
    assert [f.episode_to for f in ds.files] == [3, 2]                   # 87 < 100, the next file (50 steps) would pass it


def test_preprocess_batch(repo):
    r, _ = repo
    b = next(iter(R.SequentialReplay(r, 6, 2, seed=5)))
    out = R.preprocess_batch(b, action_dim=4, clip_rewards='tanh')
    assert out['image'].dtype == np.uint8 and out['image'].shape == (6, 2, 8, 8, 3) and out['image'].flags['C_CONTIGUOUS']
    assert out['action'].shape == (6, 2, 4) and out['action'].dtype == np.float32 and np.all(out['action'].sum(-1) == 1)
    assert np.allclose(out['reward'], np.tanh(b['reward'])) and out['terminal'].dtype == np.float32 and out['reset'].dtype == bool
