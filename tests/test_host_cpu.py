"""Host-side rows of SURVEY.md section 8 that need no GPU: the replay sampler (a1), offline-data
adoption (f3) and the C-ABI library's exported surface (b) -- each against fixtures produced by the
REFERENCE itself (tests/golden/gen_golden_host.py)."""
import ctypes
import os
import tempfile

import numpy as np

from tests.golden import gen_golden_host as gh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _make(cap):
    from repo_amd.common.buffers import SequenceReplayBuffer

    return SequenceReplayBuffer(cap, gh.OBS_SHAPE, gh.ACT_SHAPE, obs_type=np.uint8)


def test_sampler_bit_exact_vs_reference_golden():
    """push / sample (before and after the ring wraps) / iterate / save -> load -> sample reproduce the
    reference's outputs bit for bit under the same np.random seeds (common/buffers.py:146-202)."""
    want = np.load(os.path.join(GOLD, "buffer_sample.npz"))
    got = {}
    with tempfile.TemporaryDirectory() as td:
        gh.drive_sampler(_make, got, td)
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        assert got[k].dtype == want[k].dtype, (k, got[k].dtype, want[k].dtype)
        assert got[k].shape == want[k].shape, (k, got[k].shape, want[k].shape)
        assert np.array_equal(got[k], want[k]), k
    # the fixture really exercises what it claims to
    assert want["partial/pos_full_len"][1] == 0 and want["wrapped/pos_full_len"][1] == 1
    assert want["loaded/dones"][int(want["loaded/pos_full_len"][0]) - 1, 0] == 1
    assert want["partial/obs"].shape == (6, 4) + gh.OBS_SHAPE and want["partial/obs"].dtype == np.uint8


def test_saved_file_has_reference_keys():
    """save() writes exactly the reference's instance-dict keys, so either side loads the other's file."""
    b = _make(9)
    for tr in gh.host_stream(1, 5):
        b.push(*tr)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "buffer.npz")
        b.save(path)
        with np.load(path) as z:
            assert sorted(z.files) == sorted(["capacity", "observations", "actions", "rewards", "dones", "pos", "full"])


def test_offline_adoption_matches_reference_golden():
    """Dreamer.load_offline_data's buffer surgery (dreamer.py:566-596), one file at a time (the order of
    several files is glob's, as in the reference) and both together, with and without truncation."""
    want = np.load(os.path.join(GOLD, "offline_data.npz"))
    with tempfile.TemporaryDirectory() as td:
        files = gh.write_offline_files(_make, td)
        for trunc in (1000, 14):
            segs = {}
            for f in files:
                b = _make(4)
                b.adopt_offline([os.path.join(td, f)], trunc)
                for k in ("observations", "actions", "rewards", "dones"):
                    w = want[f"t{trunc}/{f}/{k}"]
                    g = getattr(b, k)
                    assert g.dtype == w.dtype and np.array_equal(g, w), (trunc, f, k)
                assert [b.capacity, b.pos, int(b.full)] == list(want[f"t{trunc}/{f}/cap_pos_full"])
                segs[f] = b
            both = _make(4)
            both.adopt_offline([os.path.join(td, f) for f in files], trunc)
            cat = np.concatenate([segs[f].observations for f in files])
            assert np.array_equal(both.observations, cat) and both.capacity == len(cat) and both.full and both.pos == 0
            # every segment ends with a terminal
            ends = np.cumsum([len(segs[f].observations) for f in files]) - 1
            assert np.all(both.dones[ends, 0] == 1)
            # ... and the adopted ring samples like any other
            np.random.seed(0)
            o, a, r, d = both.sample(3, 5)
            assert o.shape == (5, 3) + gh.OBS_SHAPE


def test_host_gather_rows_equals_numpy_take():
    """repo_host_gather_rows (the pinned path's batch gather) == ring[batch_inds]
    (common/buffers.py:186-191) for frames and for the narrow fields, any thread count; a bad index is an
    IndexError, nothing is copied out of range."""
    import pytest

    from repo_amd.common import buffers as B

    rs = np.random.RandomState(5)
    frames = rs.randint(0, 256, (300, 3, 64, 64)).astype(np.uint8)
    acts = rs.uniform(-1, 1, (300, 6)).astype(np.float32)
    inds = rs.randint(0, 300, 700)
    old = B._GATHER_THREADS
    try:
        for th in (1, 3, 8):
            B._GATHER_THREADS = th
            for src in (frames, acts):
                out = np.zeros((len(inds),) + src.shape[1:], dtype=src.dtype)
                B._gather_rows(src, inds, out)
                assert np.array_equal(out, src[inds]), (th, src.dtype)
        with pytest.raises(IndexError):
            B._gather_rows(frames, np.array([0, 300]), np.zeros((2, 3, 64, 64), np.uint8))
        with pytest.raises(IndexError):
            B._gather_rows(frames, np.array([-1]), np.zeros((1, 3, 64, 64), np.uint8))
    finally:
        B._GATHER_THREADS = old


def test_c_abi_exports_every_declared_symbol():
    """librepo_hip.so loads without a GPU and exports every prototype of include/repo_hip.h
    (no compute call is made here)."""
    import torch  # noqa: F401  (the library links against the HIP runtime torch ships)

    from repo_amd._lib import LIB_PATH, parse_header

    assert os.path.exists(LIB_PATH), "build first: python -m repo_amd.build"
    L = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    assert len(protos) >= 39
    for name in protos:
        assert hasattr(L, name), name
    L.repo_abi_version.restype = ctypes.c_int
    L.repo_strerror.restype = ctypes.c_char_p
    L.repo_strerror.argtypes = [ctypes.c_int]
    assert L.repo_abi_version() >= 1
    for code in range(-1, -8, -1):
        assert L.repo_strerror(code)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no external launcher starts two ranks itself (torch.distributed.run
    children, rendezvous on 127.0.0.1) -- checked here on CPU with the workload switched off."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"rendezvous_ranks": 2, "world_size": 2}
    # an external launcher's WORLD_SIZE that disagrees with --gpus is an error, not a silent single-rank run
    env2 = dict(env, RANK="0", WORLD_SIZE="3", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                        capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE=3" in r2.stderr
