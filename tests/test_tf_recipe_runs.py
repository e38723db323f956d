"""The committed tests/golden/tf_*.npz are what tests/golden/make_tf_goldens.py produces TODAY from /root/reference: the
recipe is re-run (the reference's own Python on the stand-in, both precisions, ~10 s) into a scratch directory and every
array of every fixture must come out equal to the committed file (integers and strings exactly, floats to the last digits: 2e-6 /
1e-12 relative for the float32 / float64 runs).  Catches a fixture edited by hand, a recipe or a
name map (tensorflowasr_amd.checkpoint.keras_names_to_abi / chunk_checkpoint_keys_to_abi -- the recipe fails on any Keras
variable it cannot place) that drifted, and a stand-in change that moved a result.  Replaces round 4's
test_tf_recipe_mock.py, whose stand-in models returned zeros.  Needs /root/reference: skips on the GPU box."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "asr", "models")), reason="needs the reference checkout")
def test_committed_tf_fixtures_are_reproduced_by_the_recipe(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_tf_goldens.py")], capture_output=True, text=True, timeout=1200,
                       cwd=ROOT, env=dict(os.environ, MI355ASR_TF_GOLDEN_OUT=str(tmp_path)))
    assert r.returncode == 0, r.stderr[-3000:]
    names = sorted(f for f in os.listdir(GOLDEN) if f.startswith("tf_") and f.endswith(".npz") and f != "tf_config2_b64.npz")
    # (tf_config2_b64.npz has its own recipe, make_tf_config2_b64.py: test_tf_goldens.py re-runs two of its utterances)
    assert names == sorted(os.listdir(tmp_path)) and len(names) == 11, (names, sorted(os.listdir(tmp_path)))
    for n in names:
        a, b = np.load(os.path.join(GOLDEN, n)), np.load(os.path.join(tmp_path, n))
        assert sorted(a.files) == sorted(b.files), n
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (n, k)
            if a[k].size == 0:
                continue
            if a[k].dtype.kind == "f":        # BLAS may sum in another order on another core count: last-digit slack only
                tol = (1e-12 if a[k].dtype == np.float64 else 2e-6) * max(1.0, float(np.abs(a[k]).max()))
                assert float(np.abs(a[k].astype(np.float64) - b[k]).max()) <= tol, (n, k)
            else:
                assert np.array_equal(a[k], b[k]), (n, k)
