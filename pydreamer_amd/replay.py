"""Replay reader -> device ring: the caller side of the hot path.

The reference feeds `train.py` from `DataSequential` (data.py:128-304: B independent episode streams cut into
truncated-BPTT windows of `batch_length`, files chosen at random forever) through `Preprocessor.apply`
(preprocessing.py:87-180) in DataLoader workers, and the trainer moves every batch to the device as float32.
Here the same batches are produced as what the HIP path consumes directly:

  * `LocalEpisodeRepository`  - episode `.npz` files in local directories (the reference lists mlflow artifacts,
    data.py:53-122; the file-name grammar `ep{from}_{to}-r{reward}-{steps}.npz` is the same);
  * `SequentialReplay`        - a window planner: per batch column a cursor (episode, row) that cuts consecutive
    `batch_length`-row windows, planned as (episode, start, stop) pieces and copied once, straight into the (T, B, ...) arrays
    the caller hands it (the pinned slot of the device ring).  Same batches as DataSequential for the same files, arguments and
    seed (random start in a column's first file, tails carried into the next file with `allow_mid_reset`, artificial resets
    every `reset_interval`, `buffer_size` / `reload_interval` file selection); the random stream is an explicit
    `numpy.random.RandomState` (the reference uses the global one);
  * `ReplayFeed`              - planner + preprocessing writing IN PLACE into a ring slot;
  * `preprocess_batch`        - the hot-path subset of Preprocessor.apply: one-hot float32 actions, float32 reward with
    `clip_rewards`, float32 terminal, bool reset - and the image is LEFT AS uint8 (T,B,H,W,C): x/255-0.5 and HWC->CHW
    happen inside the first conv's patch loader on the GPU (N1), so a batch crosses PCIe at 1 byte per pixel value;
  * `DeviceRing`              - a background thread fills pinned host buffers; the consumer enqueues the H2D copies on its
    own stream at a point where that stream is idle (`prefetch()` after `training_step()`), into a ring of device batches.

parity: pinned against the reference's own DataSequential + Preprocessor run in the build container
(oracle/gen_replay_golden.py -> tests/golden/replay_reader.npz; tests/test_replay_cpu.py replays it: same episode files,
same arguments, same seed of numpy's legacy random stream => identical batches, byte for byte, for 7 reader configurations).
The limits of that pin, as the generator's header states them: data.py imports mlflow at module scope and mlflow is absent,
so the import statements were satisfied with empty placeholder modules (nothing in them is called), and the episode source
was a local subclass of the reference's abstract EpisodeRepository instead of MlflowEpisodeRepository.  `DeviceRing` has no
reference counterpart (the reference uses a DataLoader + `.to(device)`) and is covered by its own tests.
"""
import os
import queue
import threading
import time
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class FileInfo:
    path: str
    episode_from: int
    episode_to: int
    steps: int

    def load_data(self):
        with open(self.path, 'rb') as f:
            fdata = np.load(f)
            return {k: fdata[k] for k in fdata}


def parse_episode_name(fname):
    """data.py:103-122."""
    fname = os.path.basename(fname).split('.')[0]
    steps = fname.split('-')[-1]
    steps = int(steps) if steps.isnumeric() else 0
    if fname.startswith('ep'):
        body = fname.split('ep')[1].split('-')[0]
        ep_from, ep_to = body.split('_')[0], body.split('_')[-1]
        return (int(ep_from) if ep_from.isnumeric() else 0, int(ep_to) if ep_to.isnumeric() else 0, steps)
    return (0, 0, steps)


class LocalEpisodeRepository:
    def __init__(self, dirs):
        self.dirs = [dirs] if isinstance(dirs, str) else list(dirs)

    def list_files(self):
        files = []
        for d in self.dirs:
            for name in sorted(os.listdir(d)):
                if name.endswith('.npz'):
                    a, b, steps = parse_episode_name(name)
                    files.append(FileInfo(os.path.join(d, name), a, b, steps))
        return files

    def count_steps(self):
        """data.py:90-94: (files, steps, episodes) of the repository."""
        files = self.list_files()
        return len(files), sum(f.steps for f in files), (max(f.episode_to for f in files) + 1) if files else 0

    @staticmethod
    def build_episode_name(episode_from, episode, reward, steps, chunk_seq=None):
        """data.py:97-101 (chunk_seq: the generator's sequence number of a partial episode file)."""
        if chunk_seq is None:
            return f'ep{episode_from:06}_{episode:06}-r{reward:.0f}-{steps:04}.npz'
        return f'ep{episode_from:06}_{episode:06}-{chunk_seq}-r{reward:.0f}-{steps:04}.npz'

    def save_data(self, data, episode_from, episode_to, chunk_seq=None):
        """data.py:62-69 naming; written with np.savez_compressed like tools.py:200-207."""
        n_episodes = int(data['reset'].sum())
        steps = len(data['reset']) - n_episodes
        name = self.build_episode_name(episode_from, episode_to, float(data['reward'].sum()), steps, chunk_seq)
        path = os.path.join(self.dirs[0], name)
        np.savez_compressed(path, **data)
        return path


class _Episode:
    """One episode file, prepared once: frames as (T,H,W,C) (the generator may store them time-last, data.py:237-239), the
    action that FOLLOWS each step, and the two per-file fix-ups of data.py:256-260 (a file starts with a reset and carries no
    reward on its first row).  `marks` flags the rows where a window gets an artificial reset (scatter_resets)."""
    __slots__ = ('fields', 'rows', 'marks')

    def __init__(self, arrays):
        if 'image' not in arrays and 'image_t' in arrays:
            arrays['image'] = arrays.pop('image_t').transpose(3, 0, 1, 2)
        arrays['action_next'] = np.concatenate([arrays['action'][1:], np.zeros_like(arrays['action'][:1])])
        self.rows = arrays['reward'].shape[0]
        reset = arrays['reset'].copy() if 'reset' in arrays else np.zeros(self.rows, bool)
        reward = arrays['reward'].copy()
        reset[0], reward[0] = True, 0.0
        arrays['reset'], arrays['reward'] = reset, reward
        self.fields = arrays
        self.marks = None


class _Column:
    """Cursor of one batch column: where in which episode its next window starts, how long that window is (a window that
    completes a carried tail is shorter), and the tail of the previous episode waiting to be completed."""
    __slots__ = ('episode', 'pos', 'want', 'tail', 'random_start')

    def __init__(self, random_start):
        self.episode, self.pos, self.want, self.tail, self.random_start = None, 0, 0, None, random_start


class SequentialReplay:
    """Truncated-BPTT window planner over episode files: every batch column walks its own random sequence of episodes and
    cuts consecutive windows of `batch_length` rows out of them, so the recurrent state a trainer carries from batch to batch
    (train.py:168-178) stays aligned with the data.  The batches are those of the reference's DataSequential (data.py:128-304)
    BYTE FOR BYTE for the same files, arguments and seed of numpy's legacy random stream - it draws in the same order: file
    choice, random start inside a column's first file, reset scatter of the file - which tests/test_replay_cpu.py checks
    against batches written by the reference itself.

    What is different is where the rows go.  A window is planned as one or two (episode, start, stop) pieces and copied ONCE,
    straight into the (T, B, ...) arrays the caller hands to `fill()` - for the training loop the pinned slot of the device
    ring (`ReplayFeed`), from where the frames cross PCIe as uint8 - instead of slice dicts -> np.concatenate -> np.stack ->
    another copy.  `__iter__` (tests, small tools) allocates fresh arrays per batch and fills those."""

    def __init__(self, repository, batch_length, batch_size, skip_first=True, reload_interval=0, buffer_size=0, reset_interval=0,
                 allow_mid_reset=False, seed=0, check_nonempty=True):
        self.repository = repository
        self.batch_length, self.batch_size = int(batch_length), int(batch_size)
        self.buffer_size = buffer_size                       # keep the newest files whose step counts fit (0: all)
        self.reload_interval = reload_interval               # seconds between re-listings of the repository (online training)
        self.reset_interval, self.allow_mid_reset = reset_interval, allow_mid_reset
        self.rs = np.random.RandomState(seed)
        self.columns = [_Column(bool(skip_first)) for _ in range(self.batch_size)]
        self.relist()
        if check_nonempty and not self.files:
            raise ValueError(f'no episode files in {getattr(repository, "dirs", repository)}')

    # ---- which files are in play
    def relist(self):
        listed = sorted(self.repository.list_files(), key=lambda f: -f.episode_to)      # newest first (data.py:164)
        kept, steps = [], 0
        for f in listed:
            steps += f.steps
            if not self.buffer_size or steps < self.buffer_size:
                kept.append(f)
        self.files, self.listed_steps, self.listed_at = kept, steps, time.time()

    def relist_due(self):
        return bool(self.reload_interval) and time.time() - self.listed_at > self.reload_interval

    # the names the reference trainer reads on its DataSequential (train.py:68-79, 227-229; data.py:147-170)
    @property
    def stats_steps(self):
        return self.listed_steps

    reload_files = relist
    should_reload_files = relist_due

    # ---- artificial resets: a long episode is cut into 1..steps/interval+1 backprop spans at random window-aligned rows
    def scatter_resets(self, resets, reset_interval, batch_length):
        """Same draws, same result as data.py:280-300."""
        if not resets[0]:
            raise ValueError('an episode file must start with a reset')
        starts = np.flatnonzero(resets).tolist() + [len(resets)]
        marks = np.zeros_like(resets)
        for a, b in zip(starts[:-1], starts[1:]):
            spans = self.rs.randint(1, (b - a) // reset_interval + 2)
            if spans > 1:
                gaps = np.sort(self.rs.choice(b - a - batch_length * spans, spans - 1))
                marks[a + gaps + np.arange(1, spans) * batch_length] = True
        return marks

    # ---- one column, one window
    def _open(self, col):
        """Next episode of a column: a random file; unreadable ones and ones shorter than a window are passed over (they cost
        the draw that chose them, nothing else)."""
        if self.relist_due():
            self.relist()
        info = self.files[self.rs.randint(len(self.files))]
        random_start, col.random_start = col.random_start, False
        try:
            ep = _Episode(info.load_data())
        except Exception as e:
            print('replay: skipping unreadable episode file', info.path, e)
            return
        L = self.batch_length
        if ep.rows < L:
            return
        col.pos = self.rs.randint(ep.rows - L + 1) if random_start else 0
        col.want = L - (col.tail[2] - col.tail[1]) if col.tail is not None else L
        ep.marks = self.scatter_resets(ep.fields['reset'], self.reset_interval, L) if self.reset_interval else None
        col.episode = ep

    def _plan(self, col):
        """The pieces [(episode, start, stop), ...] of the column's next window (their lengths add up to batch_length)."""
        while True:
            if col.episode is None:
                self._open(col)
                continue
            ep = col.episode
            if col.pos >= ep.rows:                           # the file ended on a window boundary
                col.episode = None
                continue
            stop = min(col.pos + col.want, ep.rows)
            piece = (ep, col.pos, stop)
            if stop - col.pos < col.want:                    # the file's last rows do not fill the window
                if col.tail is not None:
                    raise ValueError('an episode file is too short to complete the window carried over from the previous one')
                col.tail = piece if self.allow_mid_reset else None
                col.episode = None
                continue
            pieces = [piece] if col.tail is None else [col.tail, piece]
            col.tail, col.pos, col.want = None, stop, self.batch_length
            return pieces

    def _copy(self, pieces, out, b):
        t = 0
        for ep, a, z in pieces:
            n = z - a
            for k, dst in out.items():
                src = ep.fields.get(k)
                if src is None:
                    if k != 'terminal':
                        raise KeyError(f'an episode file lacks the field {k!r} the batch was laid out with')
                    dst[t:t + n, b] = 0                       # files written before `terminal` existed: no terminal rows
                else:
                    dst[t:t + n, b] = src[a:z]
            if ep.marks is not None and ep.marks[a:z].any():
                if ep.fields['reset'][a:z].any():
                    raise ValueError('an artificial reset fell into a window that holds a real one')
                out['reset'][t, b] = True                     # at the piece's first row: the longest backprop span
            t += n

    # ---- batches
    def fill(self, out=None):
        """Write the next batch into `out` ({key: (T, B, ...) array}; only the keys present are written - all of the files' keys
        when out is None or empty, in freshly allocated arrays).  Returns out."""
        plans = [self._plan(col) for col in self.columns]
        if not out:
            first = plans[0][0][0].fields
            out = {} if out is None else out
            for k, v in first.items():
                out[k] = np.empty((self.batch_length, self.batch_size) + v.shape[1:], v.dtype)
            if 'terminal' not in out:      # the first file predates `terminal` but another piece of this batch carries it: same
                for pieces in plans:       # column as ReplayFeed lays out (zeros for the files without it, _copy)
                    src = next((ep.fields['terminal'] for ep, _, _ in pieces if 'terminal' in ep.fields), None)
                    if src is not None:
                        out['terminal'] = np.empty((self.batch_length, self.batch_size) + src.shape[1:], src.dtype)
                        break
        for b, pieces in enumerate(plans):
            self._copy(pieces, out, b)
        return out

    def __iter__(self):
        while True:
            yield self.fill()


def preprocess_batch(batch, action_dim, clip_rewards=None, image_key='image'):
    """Hot-path subset of Preprocessor.apply (preprocessing.py:87-180), images left uint8 (T,B,H,W,C)."""
    T, B = batch['reward'].shape[:2]
    out = {}
    img = batch[image_key]
    assert img.dtype == np.uint8 and img.ndim == 5, f'expected uint8 (T,B,H,W,C) frames, got {img.dtype} {img.shape}'
    out['image'] = np.ascontiguousarray(img)
    for k in ('action', 'action_next'):
        if k in batch:
            a = batch[k]
            if a.ndim == 2:
                a = np.eye(action_dim, dtype=np.float32)[a]
            assert a.ndim == 3
            out[k] = a.astype(np.float32)
    out['terminal'] = batch.get('terminal', np.zeros((T, B))).astype(np.float32)
    r = batch.get('reward', np.zeros((T, B))).astype(np.float32)
    if clip_rewards == 'tanh':
        r = np.tanh(r)
    elif clip_rewards == 'log1p':
        r = np.log1p(r)
    elif clip_rewards:
        raise ValueError(clip_rewards)
    out['reward'] = r
    out['reset'] = batch.get('reset', np.zeros((T, B), bool)).astype(bool)
    return out


class ReplayFeed:
    """Planner + hot-path preprocessing writing IN PLACE into a slot of host arrays (DeviceRing hands it the numpy views of a
    pinned slot): the uint8 frame windows and the reset flags go from the episode arrays straight into the slot - one copy
    between the file and PCIe - and the small per-step fields (actions, reward, terminal) through a reused scratch batch.
    Field for field what `preprocess_batch(next(iter(replay)), ...)` returns (tests/test_replay_cpu.py compares the two)."""

    def __init__(self, replay, action_dim, clip_rewards=None, image_key='image'):
        if clip_rewards not in (None, '', False, 'tanh', 'log1p'):
            raise ValueError(clip_rewards)
        self.replay, self.action_dim, self.clip_rewards, self.image_key = replay, int(action_dim), clip_rewards, image_key
        if not replay.files:
            raise ValueError('ReplayFeed needs at least one episode file to lay out its slots (the repository is empty)')
        probe = _Episode(replay.files[0].load_data()).fields      # shapes only; draws nothing from the random stream
        T, B = replay.batch_length, replay.batch_size
        # `terminal` always has a column: an episode without the field contributes zeros (SequentialReplay._copy), so a
        # repository that mixes files with and without it still yields the flags of those that carry them
        self._small = {k: np.empty((T, B) + probe[k].shape[1:], probe[k].dtype)
                       for k in ('action', 'action_next', 'reward') if k in probe}
        self._small['terminal'] = np.empty((T, B), probe['terminal'].dtype if 'terminal' in probe else np.float32)
        img = probe[image_key]
        if img.dtype != np.uint8 or img.ndim != 4:
            raise ValueError(f'expected uint8 (T,H,W,C) frames in the episode files, got {img.dtype} {img.shape}')
        self._spec = {'image': ((T, B) + img.shape[1:], np.uint8), 'action': ((T, B, self.action_dim), np.float32),
                      'action_next': ((T, B, self.action_dim), np.float32), 'terminal': ((T, B), np.float32),
                      'reward': ((T, B), np.float32), 'reset': ((T, B), np.bool_)}

    def spec(self):
        """{field: (shape, dtype)} of a slot."""
        return dict(self._spec)

    def _actions(self, src, dst):
        if src.ndim == 2:                                     # integer actions -> one-hot rows
            dst[...] = 0.0
            np.put_along_axis(dst, src[..., None].astype(np.int64), 1.0, axis=2)
        else:
            dst[...] = src

    def fill(self, slot):
        raw = dict(self._small)
        raw[self.image_key] = slot['image']
        raw['reset'] = slot['reset']
        self.replay.fill(raw)
        self._actions(raw['action'], slot['action'])
        self._actions(raw['action_next'], slot['action_next'])
        slot['terminal'][...] = raw['terminal']
        r = raw['reward'].astype(np.float32)
        if self.clip_rewards == 'tanh':
            r = np.tanh(r)
        elif self.clip_rewards == 'log1p':
            r = np.log1p(r)
        slot['reward'][...] = r
        return slot


class DeviceRing:
    """Pinned staging + asynchronous H2D into a ring of device-resident batches, WITHOUT a stream of its own.

    A producer thread pulls numpy batches from `source` (an iterator of preprocess_batch outputs) into one of `depth` pinned
    host slots - host work only.  The H2D copies are enqueued by the CONSUMER, on whatever stream is current, in one of two
    places: `prefetch()` stages the next batch (call it right after `training_step()` returned: the caller's stream has
    nothing left to do then while the backward passes run on their own streams, so the ~31 MB transfer is hidden), and
    `next()` returns the staged batch, staging it first if nobody prefetched (the copy then sits in front of the step).
    Ordering needs no events on the device side: a device slot is overwritten by a copy on the caller's stream `depth`
    batches after it was handed out, and everything that read it (the step's kernels, the side-stream backward passes that
    `loss.backward()` joins) is ordered before that on the same stream.  A pinned slot is rewritten by the producer only
    after a host-side wait for the event recorded behind its copy.

    Why no copy stream (rounds 1-2 had one): ROCm multiplexes HIP streams onto four hardware queues by default, and the
    step already uses the caller's stream, two backward streams and the library's weight-gradient side stream.  A copy
    stream is the fifth; when it lands on the queue of a stream that is parked behind an event for most of a step (the
    actor-critic backward stream is), the transfer for step n+2 completes a step late - measured as a bimodal H2D-included
    step, 37.3 or 44-47 ms depending on the process."""

    def __init__(self, source, device, depth=4):
        # source: an iterator of {key: numpy array} batches (copied into the pinned slot), or a feed with spec() / fill(slot)
        # (ReplayFeed) that writes the batch INTO the pinned slot
        self.feed = source if hasattr(source, 'fill') and hasattr(source, 'spec') else None
        self.source = None if self.feed is not None else iter(source)
        self.device, self.depth = torch.device(device), max(3, depth)
        self.free = queue.Queue()               # pinned slots the producer may fill
        self.filled = queue.Queue()             # pinned slots holding a batch, in source order (None: exhausted / failed)
        self.host = None                        # per pinned slot: {key: pinned tensor}
        self.copied = [None] * self.depth       # per pinned slot: event behind its last H2D copy
        self.dev = None                         # per device slot: {key: device tensor}
        self.next_dev = 0
        self.staged = None
        self.error = None
        self.done = False
        for i in range(self.depth):
            self.free.put(i)
        self.thread = threading.Thread(target=self._produce, daemon=True, name='dm-replay')
        self.thread.start()

    def _produce_feed(self):
        spec = self.feed.spec()
        self.host = [None] * self.depth
        while True:
            i = self.free.get()
            if i is None:
                return
            if self.host[i] is None:
                with torch.cuda.device(self.device):        # (a new thread's current device is 0)
                    self.host[i] = {k: torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dt)).dtype).pin_memory()
                                    for k, (shape, dt) in spec.items()}
            elif self.copied[i] is not None:                 # the pinned buffer is the source of an asynchronous copy until this fires
                self.copied[i].synchronize()
            self.feed.fill({k: t.numpy() for k, t in self.host[i].items()})
            self.filled.put(i)

    def _produce(self):
        try:
            if self.feed is not None:
                return self._produce_feed()
            for batch in self.source:
                i = self.free.get()
                if i is None:
                    return
                if self.host is None:
                    self.host = [None] * self.depth
                if self.host[i] is None:
                    # a new thread's current device is 0: pin under THIS ring's device, or every rank creates a context on GPU 0
                    with torch.cuda.device(self.device):
                        self.host[i] = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().pin_memory() for k, v in batch.items()}
                else:
                    ev = self.copied[i]
                    if ev is not None:          # the pinned buffer is still the source of an asynchronous copy until this fires
                        ev.synchronize()
                    for k, v in batch.items():
                        self.host[i][k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
                self.filled.put(i)
            self.filled.put(None)               # source exhausted: next() raises StopIteration after the last batch
        except Exception as e:                  # surfaced by next()
            self.error = e
            self.filled.put(None)

    def __iter__(self):
        return self

    __next__ = lambda self: self.next()

    def prefetch(self):
        """Stage the next batch: enqueue its H2D copies on the current stream (no-op if one is staged already)."""
        if self.staged is not None or self.done:
            return
        i = self.filled.get()
        if i is None:
            self.done = True
            return
        host = self.host[i]
        with torch.cuda.device(self.device):
            if self.dev is None:
                self.dev = [{k: torch.empty_like(h, device=self.device) for k, h in host.items()} for _ in range(self.depth)]
            d = self.dev[self.next_dev]
            self.next_dev = (self.next_dev + 1) % self.depth
            for k in host:
                d[k].copy_(host[k], non_blocking=True)
            ev = self.copied[i] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.copied[i] = ev
        self.free.put(i)                        # the producer waits for `ev` before touching the pinned buffers again
        self.staged = d

    def next(self):
        self.prefetch()
        if self.staged is None:
            if self.error is not None:
                raise RuntimeError('replay producer failed') from self.error
            raise StopIteration
        d, self.staged = self.staged, None
        return d

    def close(self):
        self.free.put(None)
