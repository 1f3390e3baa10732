"""Generate the replay-reader golden vectors from the REAL reference (build container only; /root/reference must exist).

    python oracle/gen_replay_golden.py            # writes tests/golden/replay_reader.npz

What runs is the reference's own `DataSequential.__iter__ / iter_single / iter_file / randomize_resets`
(pydreamer/data.py:128-304) and `Preprocessor.apply` (pydreamer/preprocessing.py:87-180), unmodified, over episode files
this script writes with the reference's own `save_npz` (tools.py:200-207) in the generator's on-disk format
(`image_t` HWCT for half of them, generator.py:246-249).

Two things stand between `import pydreamer.data` and this container, both stated here because they limit the claim:
  * data.py:11-13 imports two names from `mlflow` at module scope and `mlflow_load_npz` (tools.py:149-154) does
    `import mlflow`; mlflow is not installed.  EMPTY placeholder modules are put into sys.modules so that those import
    statements succeed; no attribute of them is ever called (`ArtifactRepository` is only a type annotation,
    `get_artifact_repository` is only used by MlflowEpisodeRepository, which is not instantiated).
  * the episode source is a local `EpisodeRepository` subclass (data.py:42-50 is the abstract interface the reference
    defines for exactly this) whose FileInfo.artifact_repo is a 3-line object with `_download_file(name, dst)` = file copy.
The reader algorithm, the file loader, the preprocessing and the random stream (numpy's global legacy RandomState,
seeded per case) are the reference's.

The fixture is data: the episode arrays (inputs), per case the constructor arguments + seed, and the first K batches the
reference produced (raw reader output and Preprocessor output).  tests/test_replay_cpu.py replays it through
pydreamer_amd/replay.py.
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

ACTION_DIM = 5
EPISODES = [      # (file index from, to, steps incl. resets rows, inner reset positions, stored transposed)
    (0, 0, 37, (), False),
    (1, 2, 83, (41,), True),              # two episodes in one file (generator.py chunks)
    (3, 3, 52, (), True),
    (4, 6, 121, (40, 80), False),
    (7, 7, 64, (), True),
    (8, 8, 19, (), False),                # shorter than batch_length 20: exercised by the "too short file" branch
]
CASES = [         # name, kwargs of DataSequential, numpy seed, batches, clip_rewards
    ('default', dict(batch_length=10, batch_size=3), 1, 14, None),
    ('no_skip_first', dict(batch_length=10, batch_size=4, skip_first=False), 2, 12, 'tanh'),
    ('mid_reset', dict(batch_length=12, batch_size=3, allow_mid_reset=True), 3, 16, None),
    ('random_resets', dict(batch_length=8, batch_size=3, reset_interval=16), 4, 16, None),
    ('random_resets_mid', dict(batch_length=8, batch_size=2, reset_interval=12, allow_mid_reset=True), 5, 20, 'log1p'),
    ('buffer_size', dict(batch_length=10, batch_size=3, buffer_size=200), 6, 12, None),
    ('long_window', dict(batch_length=20, batch_size=2, allow_mid_reset=True), 7, 12, None),
]


def make_episodes():
    rs = np.random.RandomState(1234)
    eps = []
    for (a, b, n, inner, transposed) in EPISODES:
        reset = np.zeros(n, bool)
        reset[0] = True
        for p in inner:
            reset[p] = True
        terminal = np.zeros(n, bool)
        for p in list(inner) + [n]:
            terminal[p - 1] = True
        d = dict(action=rs.randint(0, ACTION_DIM, n).astype(np.int64),
                 reward=np.round(rs.rand(n) * 3, 2).astype(np.float32),     # non-negative: log1p is defined
                 terminal=terminal, reset=reset,
                 image=rs.randint(0, 256, (n, 8, 8, 3)).astype(np.uint8))
        eps.append((a, b, transposed, d))
    return eps


def main():
    for name in ('mlflow', 'mlflow.store', 'mlflow.store.artifact', 'mlflow.store.artifact.artifact_repo',
                 'mlflow.store.artifact.artifact_repository_registry'):
        sys.modules[name] = types.ModuleType(name)           # empty placeholders: import statements only (see header)
    sys.modules['mlflow.store.artifact.artifact_repo'].ArtifactRepository = object
    sys.modules['mlflow.store.artifact.artifact_repository_registry'].get_artifact_repository = None
    sys.path.insert(0, REF)
    from pydreamer import data as RD
    from pydreamer import tools as RT
    from pydreamer.preprocessing import Preprocessor

    tmp = tempfile.mkdtemp()

    class CopyRepo:                                           # stands where an mlflow ArtifactRepository would
        def _download_file(self, name, dst):
            shutil.copy(os.path.join(tmp, name), dst)

    class LocalRepo(RD.EpisodeRepository):
        parse = RD.MlflowEpisodeRepository.parse_episode_name
        build = RD.MlflowEpisodeRepository.build_episode_name

        def save_data(self, data, episode_from, episode_to):
            n_episodes = data['reset'].sum()
            fname = self.build(episode_from, episode_to, data['reward'].sum(), len(data['reset']) - n_episodes)
            RT.save_npz(data, os.path.join(tmp, fname))
            return fname

        def list_files(self):
            repo = CopyRepo()
            out = []
            for f in sorted(os.listdir(tmp)):
                a, b, steps = self.parse(f)
                out.append(RD.FileInfo(path=f, episode_from=a, episode_to=b, steps=steps, artifact_repo=repo))
            return out

    repo = LocalRepo()
    out = {}
    names = []
    for i, (a, b, transposed, d) in enumerate(make_episodes()):
        stored = dict(d)
        if transposed:
            stored['image_t'] = stored.pop('image').transpose(1, 2, 3, 0)       # THWC -> HWCT as generator.py:246-249
        names.append(repo.save_data(stored, a, b))
        for k, v in stored.items():
            out[f'episode{i}/{k}'] = v
    out['episode_files'] = np.array(names)
    out['action_dim'] = np.int64(ACTION_DIM)
    # the file-name grammar on the reference's own parser (data.py:103-122)
    probe = ['ep000012_000014-r35-0421.npz', 'x/y/ep000007-r-3-0099.npz', '20210101T000000-0500.npz', 'ep000003_000004-2-r7-0100.npz'] + names
    out['name_probe'] = np.array(probe)
    out['name_probe_parsed'] = np.array([repo.parse(p) for p in probe], np.int64)
    # ... and the builder (data.py:97-101), with and without the generator's chunk sequence number
    build_args = [(3, 4, 7.4, 100, None), (3, 4, 7.6, 100, 2), (0, 0, -3.0, 9, None), (12, 14, 35.49, 421, 0), (123456, 123457, 0.0, 12345, None)]
    out['name_build_args'] = np.array([[a, b, r, n, -1 if c is None else c] for a, b, r, n, c in build_args], np.float64)
    out['name_build'] = np.array([repo.build(a, b, r, n, chunk_seq=c) for a, b, r, n, c in build_args])

    case_names = []
    for (cname, kw, seed, nb, clip) in CASES:
        np.random.seed(seed)
        ds = RD.DataSequential(repo, **kw)
        out[f'case/{cname}/files_kept'] = np.array([f.path for f in ds.files])
        out[f'case/{cname}/stats_steps'] = np.int64(ds.stats_steps)
        pre = Preprocessor(image_key='image', action_dim=ACTION_DIM, clip_rewards=clip)
        it = iter(ds)
        raw, prep = [], []
        for _ in range(nb):
            b = next(it)
            raw.append({k: np.array(v) for k, v in b.items()})
            prep.append(pre.apply({k: np.array(v) for k, v in b.items()}))
        for k in raw[0]:
            out[f'case/{cname}/raw/{k}'] = np.stack([r[k] for r in raw])
        for k in prep[0]:
            keep = prep[:2] if k == 'image' else prep             # float images: the first two batches are enough (size)
            out[f'case/{cname}/prep/{k}'] = np.stack([p[k] for p in keep])
        out[f'case/{cname}/kwargs'] = np.array(repr(kw))
        out[f'case/{cname}/seed'] = np.int64(seed)
        out[f'case/{cname}/clip_rewards'] = np.array(clip or '')
        case_names.append(cname)
        print(cname, {k: out[f'case/{cname}/raw/{k}'].shape for k in raw[0]})
    out['cases'] = np.array(case_names)

    # randomize_resets alone (data.py:280-300), on a long reset vector
    resets = np.zeros(400, bool)
    resets[[0, 90, 250]] = True
    np.random.seed(11)
    rr = np.stack([ds.randomize_resets(resets, 25, 10) for _ in range(8)])
    out['randomize_resets/resets'] = resets
    out['randomize_resets/out'] = rr
    path = os.path.join(ROOT, 'tests', 'golden', 'replay_reader.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
