"""The pin recipe must stay runnable: both golden generators are re-run here against /root/reference (build
container only -- the reference never travels to the GPU box) into a temp dir and every regenerated fixture must
equal the committed one.  (gen_golden_host.py crashed at HEAD and no test noticed.)"""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/algorithms"),
                                reason="needs the reference checkout (build container only)")


def _run(script, out, *flags):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, script), "--out", str(out), *flags], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _same_npz(a, b):
    x, y = np.load(a), np.load(b)
    assert sorted(x.files) == sorted(y.files), (a, sorted(set(x.files) ^ set(y.files)))
    for k in x.files:
        if x[k].dtype.kind in "USb" or x[k].dtype.kind in "iu":
            assert np.array_equal(x[k], y[k]), (a, k)
        else:  # thread-count dependent summation order: observed 0, allowed a few ulps of fp32
            np.testing.assert_allclose(x[k], y[k], rtol=2e-6, atol=1e-9, err_msg=f"{a}:{k}")


def test_host_generator_reproduces_committed_fixtures(tmp_path):
    _run("gen_golden_host.py", tmp_path)
    for f in ("buffer_sample.npz", "offline_data.npz"):
        _same_npz(tmp_path / f, os.path.join(GOLDEN, f))
    assert filecmp.cmp(tmp_path / "checkpoint_manifest.json", os.path.join(GOLDEN, "checkpoint_manifest.json"), shallow=False)


def test_update_generator_reproduces_committed_fixtures(tmp_path):
    _run("gen_golden.py", tmp_path, "--skip-c1")
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert made == ["dreamer_tiny.npz", "finetune_tiny.npz", "mt_dreamer_tiny.npz", "mt_repo_tiny.npz", "repo_odd.npz",
                    "repo_tiny.npz", "tia_coefs.npz", "tia_tiny.npz", "tia_zeros.npz"]
    for f in made:
        _same_npz(tmp_path / f, os.path.join(GOLDEN, f))
