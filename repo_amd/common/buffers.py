"""Sequence replay buffer: NumPy ring on the host + pinned staging for async H2D.

Ring semantics follow the reference's SequenceReplayBuffer
(/root/reference/common/buffers.py:128-202): flat arrays, `push` advances a write head that
wraps (:146-154); `sample(B, L)` draws B start indices with np.random.choice(len - L),
builds a (B, L) index matrix, transposes it to time-major, rotates by the write head once
the ring is full so a sequence never straddles the head (:156-166); save/load as .npz of the
instance dict, marking the last stored transition done on load (:193-202).

MI355X additions:
 * `prefetch/acquire/release` gather a batch into one of two page-locked staging slots and issue
   the copies on a side HIP stream (hipMemcpyAsync under torch's non_blocking copy), so the 30.7 MB
   uint8 batch of update k+1 crosses PCIe while update k computes.
 * `enable_device_mirror(device)`: with 288 GB of HBM the whole ring (1e6 frames x 12 KB = 12.3 GB)
   fits on the GPU many times over, and the host-side gather of 2500 scattered 12 KB frames (10+ ms,
   single-threaded) is slower than the update it feeds.  The mirror keeps a device copy of the ring
   that is topped up with the frames pushed since the last batch (one or two contiguous H2D copies),
   and a batch becomes ONE device gather kernel per field driven by the same host-drawn indices.
   The host arrays stay the source of truth (save/load, reference semantics unchanged).
Frames stay uint8 on the device; normalisation is fused into the first convolution's loader.
"""
import os

import numpy as np
import torch


_FIELDS = ("observations", "actions", "rewards", "dones")
# host threads of the pinned-path batch gather (repo_host_gather_rows); REPO_GATHER_THREADS overrides
_GATHER_THREADS = int(os.environ.get("REPO_GATHER_THREADS", "0")) or max(1, min(8, (os.cpu_count() or 2) // 2))


def _gather_rows(src, inds, out):
    """out[i] = src[inds[i]] (the reference's `ring[batch_inds]`, common/buffers.py:186-191) by the C helper: a few
    host threads copy the rows straight into `out` (a page-locked staging slot) with the GIL released.  Falls back
    to np.take when the library is not built (host-only use of the buffer)."""
    try:
        from .._lib import RepoHipError, check, lib
        L = lib()
    except Exception:  # noqa: BLE001  (no librepo_hip.so: the buffer still works as a plain host ring)
        np.take(src, inds, axis=0, out=out)
        return
    assert src.flags.c_contiguous and out.flags.c_contiguous and out.dtype == src.dtype and out.shape[1:] == src.shape[1:]
    idx = np.ascontiguousarray(inds, dtype=np.int64)
    row_bytes = src[0].nbytes if src.shape[0] else 0
    if len(idx) == 0 or row_bytes == 0:
        return
    rc = L.repo_host_gather_rows(src.ctypes.data, src.shape[0], row_bytes, idx.ctypes.data, len(idx), out.ctypes.data,
                                 _GATHER_THREADS)
    if rc == -2:  # REPO_E_SHAPE: what NumPy reports for the same mistake
        raise IndexError("replay gather: index out of range")
    check(rc, "repo_host_gather_rows")


def _time_major_indices(starts, seq_len):
    """Flat (seq_len * batch) ring indices of `batch` windows, time-major: entry t*batch + b is
    starts[b] + t (the reference stacks (B, L) aranges and transposes: common/buffers.py:159-160)."""
    return (np.asarray(starts)[None, :] + np.arange(seq_len)[:, None]).reshape(-1)


class SequenceReplayBuffer:
    def __init__(self, capacity, obs_shape, act_shape, obs_type=np.float32, act_type=np.float32):
        self.capacity = capacity
        self.observations = np.zeros((capacity,) + tuple(obs_shape), dtype=obs_type)
        self.actions = np.zeros((capacity,) + tuple(act_shape), dtype=act_type)
        self.rewards = np.zeros((capacity, 1), dtype=np.float32)
        self.dones = np.zeros((capacity, 1), dtype=np.float32)
        self.pos = 0        # write head
        self.full = False   # the head has wrapped at least once
        self._pushed_total = 0  # monotonic count of stored transitions (device mirror bookkeeping)
        self._mirror = None

    # the per-transition arrays, in the order sample() returns them (the multitask buffer prepends `tasks`)
    _RING_FIELDS = _FIELDS

    def _rings(self):
        return tuple(getattr(self, f) for f in self._RING_FIELDS)

    # exactly the keys the reference's save() writes (its instance dict, common/buffers.py:193-194)
    @property
    def _DATA_KEYS(self):
        return ("capacity",) + _FIELDS + ("pos", "full") + tuple(f for f in self._RING_FIELDS if f not in _FIELDS)

    def __len__(self):
        return self.capacity if self.full else self.pos

    def push(self, obs, act, rew, done):
        """Store one transition at the write head and advance it (wraps; reference :146-154)."""
        at = self.pos
        for ring, value in zip((self.observations, self.actions, self.rewards, self.dones), (obs, act, rew, done)):
            ring[at] = np.asarray(value)
        self._pushed_total += 1
        self.pos = (at + 1) % self.capacity
        self.full = self.full or self.pos == 0

    def _unrotate(self, flat_inds):
        """Window offsets are drawn relative to the OLDEST stored transition; once the ring has wrapped
        that is the write head, so no window straddles it (reference :161-163)."""
        return (flat_inds + self.pos) % len(self) if self.full else flat_inds

    def _sample_inds(self, batch_size, seq_len):
        starts = np.random.choice(len(self) - seq_len, size=batch_size)  # the reference's ONE RNG draw (:158)
        return self._unrotate(_time_major_indices(starts, seq_len))

    def sample(self, batch_size, seq_len):
        """-> (obs, act, rew, done), each (seq_len, batch_size, ...), host arrays."""
        picked = self._get_samples(self._sample_inds(batch_size, seq_len))
        return tuple(a.reshape(seq_len, batch_size, *a.shape[1:]) for a in picked)

    def iterate(self, batch_size, seq_len):
        """One shuffled pass over non-overlapping windows (reference :168-184, including its quirks: the
        window starts are rotated by the write head before shuffling AND the gathered indices once more,
        and a trailing group of exactly `batch_size` windows is dropped by the exclusive range)."""
        starts = np.arange(0, len(self) - seq_len, seq_len)
        starts = self._unrotate(starts)
        np.random.shuffle(starts)
        for first in range(0, len(starts) - batch_size, batch_size):
            inds = self._unrotate(_time_major_indices(starts[first:first + batch_size], seq_len))
            yield [a.reshape(seq_len, batch_size, *a.shape[1:]) for a in self._get_samples(inds)]

    def _get_samples(self, batch_inds):
        return tuple(r[batch_inds] for r in self._rings())

    def save(self, path):
        np.savez(path, **{k: getattr(self, k) for k in self._DATA_KEYS})

    def load(self, path):
        """Adopt a saved ring; the transition just behind the write head is marked terminal because the
        episode it belonged to was cut by the save (reference :196-202)."""
        with np.load(path) as z:
            for key in self._DATA_KEYS:
                setattr(self, key, z[key])
        self.capacity, self.pos, self.full = int(self.capacity), int(self.pos), bool(self.full)
        if len(self) > 0:
            self.dones[self.pos - 1] = 1
        self.invalidate_mirror()

    def adopt_offline(self, paths, truncate_size):
        """Offline datasets (reference algorithms/repo/dreamer.py:566-596): every file is a saved ring;
        read each in chronological order, keep its first `truncate_size` transitions, end it with a
        terminal, and make the concatenation THE ring (capacity = total, full, head at 0)."""
        parts = {f: [] for f in _FIELDS}
        for path in paths:
            with np.load(path) as z:
                head, wrapped = int(z["pos"]), bool(z["full"])
                n_stored = len(z["observations"]) if wrapped else head
                oldest = head if wrapped else 0
                order = (oldest + np.arange(min(n_stored, int(truncate_size)))) % max(len(z["observations"]), 1)
                piece = {f: z[f][order] for f in _FIELDS}
            piece["dones"][-1, :] = 1
            for f in _FIELDS:
                parts[f].append(piece[f])
        for f in _FIELDS:
            setattr(self, f, np.concatenate(parts[f]))
        self.capacity = len(self.observations)
        self.pos, self.full = 0, True
        self.invalidate_mirror()

    # ------------------------------------------------------------------ device-resident mirror
    def enable_device_mirror(self, device):
        """Ask for batches to be gathered on `device` from a device copy of the ring (allocated and
        filled lazily at the first prefetch)."""
        self._mirror = {"device": torch.device(device), "bufs": None, "synced": 0}

    def _mirror_on(self, device):
        m = getattr(self, "_mirror", None)
        return m is not None and m["device"] == torch.device(device)

    def invalidate_mirror(self):
        """Call after writing the host arrays directly (load(), tests): forces a full re-upload."""
        self._pushed_total = getattr(self, "_pushed_total", 0) + self.capacity
        if getattr(self, "_mirror", None) is not None:
            self._mirror["synced"] = self._pushed_total - self.capacity - 1  # => everything is stale
            bufs = self._mirror["bufs"]
            if bufs is not None and bufs[0].shape[0] != self.capacity:
                self._mirror["bufs"] = None  # load() / adopt_offline() changed the capacity: reallocate the mirror

    def _mirror_flush(self, stream):
        m = self._mirror
        srcs = self._rings()
        if m["bufs"] is None:
            m["bufs"] = tuple(torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype, device=m["device"]) for a in srcs)
            m["synced"] = self._pushed_total - self.capacity - 1
        stale = self._pushed_total - m["synced"]
        if stale <= 0:
            return
        if stale >= self.capacity:
            pieces = [(0, len(self))]
        else:  # the last `stale` ring slots ending at the write head
            a = (self.pos - stale) % self.capacity
            pieces = [(a, self.pos)] if a < self.pos else [(a, self.capacity), (0, self.pos)]
        with torch.cuda.stream(stream):
            for lo, hi in pieces:
                if hi > lo:
                    for d, src in zip(m["bufs"], srcs):
                        d[lo:hi].copy_(torch.from_numpy(src[lo:hi]))
        m["synced"] = self._pushed_total

    # ------------------------------------------------------------------ device staging
    def _staging(self, batch_size, seq_len, device):
        key = (batch_size, seq_len, str(device))
        st = getattr(self, "_stage", None)
        if st is None or st["key"] != key:
            pin = device.type == "cuda"

            def host(shape, dtype):
                t = torch.empty(shape, dtype=dtype)
                return t.pin_memory() if pin else t

            n = seq_len * batch_size
            slots = []
            for _ in range(2):
                # frames keep the ring's dtype (uint8 on the device); every other field is float32
                h = tuple(host((n,) + r.shape[1:], torch.from_numpy(r[:0]).dtype if r is self.observations
                               else torch.float32) for r in self._rings())
                d = tuple(torch.empty(t.shape, dtype=t.dtype, device=device) for t in h)
                slots.append({"host": h, "dev": d, "event": None,
                              "idx_host": host((n,), torch.int64),
                              "idx_dev": torch.empty(n, dtype=torch.int64, device=device)})
            st = {"key": key, "slots": slots, "next": 0,
                  "stream": torch.cuda.Stream(device=device) if pin and not self._mirror_on(device) else None}
            self._stage = st
        return st

    def prefetch(self, batch_size, seq_len, device):
        """Sample a batch on the host into the next pinned slot and start its host->device
        copy on the side stream.  Returns a handle for acquire()/release().  Call it right
        AFTER enqueuing update k so the gather and the PCIe copy of batch k+1 overlap it."""
        st = self._staging(batch_size, seq_len, device)
        idx = st["next"]
        st["next"] ^= 1
        slot = st["slots"][idx]
        inds = self._sample_inds(batch_size, seq_len)
        if self._mirror_on(device):
            # Device-mirror mode: the batch is gathered by acquire() ON THE CONSUMER'S STREAM (a ~30 us
            # kernel), so no staging stream exists at all -- an extra stream costs more than it hides here
            # (HIP shares 4 hardware queues among all streams).  prefetch only draws the
            # indices into this slot's pinned vector.
            if slot["event"] is not None:
                slot["event"].synchronize()  # this slot's index upload of two batches ago
            slot["idx_host"].copy_(torch.from_numpy(inds.astype(np.int64)))
            slot["pending"] = True
            return idx
        if slot["event"] is not None:
            slot["event"].synchronize()  # previous copy out of this pinned slot has finished
        for h, src in zip(slot["host"], self._rings()):
            _gather_rows(src, inds, h.numpy())
        if st["stream"] is not None:
            consumed = slot.get("consumed")
            if consumed is not None:
                st["stream"].wait_event(consumed)  # the update that last read this device slot
            else:
                st["stream"].wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(st["stream"]):
                for h, d in zip(slot["host"], slot["dev"]):
                    d.copy_(h, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st["stream"])
            slot["event"] = ev
        else:
            for h, d in zip(slot["host"], slot["dev"]):
                d.copy_(h)
        return idx

    def acquire(self, handle, batch_size, seq_len, device):
        """Make the current stream wait for the slot's copy; returns the device tensors
        (obs uint8 (L,B,C,H,W), actions (L,B,A), rewards (L,B,1), dones (L,B,1); the multitask buffer: tasks
        (L,B,num_tasks) first)."""
        st = self._staging(batch_size, seq_len, device)
        slot = st["slots"][handle]
        if self._mirror_on(device):
            if slot.get("pending"):
                cur = torch.cuda.current_stream(device)
                self._mirror_flush(cur)
                consumed = slot.get("consumed")
                if consumed is not None:
                    # the update that last READ this slot's device tensors may run on other streams
                    # (the agent's world-model lane and its weight-gradient side stream)
                    cur.wait_event(consumed)
                slot["idx_dev"].copy_(slot["idx_host"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(cur)
                slot["event"] = ev
                for src, d in zip(self._mirror["bufs"], slot["dev"]):
                    torch.index_select(src, 0, slot["idx_dev"], out=d)
                slot["pending"] = False
            return tuple(d.view(seq_len, batch_size, *d.shape[1:]) for d in slot["dev"])
        if slot["event"] is not None and st["stream"] is not None:
            torch.cuda.current_stream(device).wait_event(slot["event"])
        return tuple(d.view(seq_len, batch_size, *d.shape[1:]) for d in slot["dev"])

    def release(self, handle, batch_size, seq_len, device):
        """Record that everything enqueued so far on the current stream has consumed the slot."""
        st = self._staging(batch_size, seq_len, device)
        if self._mirror_on(device) or st["stream"] is not None:
            # call this on the stream that joins every reader of the slot (Dreamer.train_agent records it on
            # the world-model lane after the update is enqueued); the next gather / copy INTO the slot waits
            # for it, whichever stream that gather runs on
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            st["slots"][handle]["consumed"] = ev

    def sample_to_device(self, batch_size, seq_len, device):
        """prefetch + acquire in one call (no overlap; for simple callers and tests)."""
        h = self.prefetch(batch_size, seq_len, device)
        return self.acquire(h, batch_size, seq_len, device)

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_stage", None)
        return d


class MultitaskSequenceReplayBuffer(SequenceReplayBuffer):
    """The reference's multitask ring (/root/reference/common/buffers.py:205-225): a `tasks` array (capacity,
    num_tasks) beside the four rings, `push(task, obs, act, rew, done)`, and every batch led by its task one-hots:
    sample() -> (tasks (L,B,num_tasks), obs, act, rew, done).  The pinned staging path and the HBM mirror carry the
    fifth field like the others."""

    _RING_FIELDS = ("tasks",) + _FIELDS

    def __init__(self, capacity, num_tasks, obs_shape, act_shape, obs_type=np.float32, act_type=np.float32):
        super().__init__(capacity, obs_shape, act_shape, obs_type, act_type)
        self.tasks = np.zeros((self.capacity, num_tasks), dtype=act_type)

    def push(self, task, obs, act, rew, done):
        self.tasks[self.pos] = np.asarray(task)
        super().push(obs, act, rew, done)

    def adopt_offline(self, paths, truncate_size):
        raise NotImplementedError("offline datasets carry no task labels (the reference's load_offline_data is "
                                  "single-task: algorithms/repo/dreamer.py:566-596)")
