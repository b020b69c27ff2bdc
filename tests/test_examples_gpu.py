"""The end-to-end example (text file -> front-end -> sampler -> the driver's files) runs and writes what it says."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["ggs", "pcgs"])
def test_run_dataset_example_on_cats(tmp_path, scheme):
    out = tmp_path / "run"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_dataset.py"), os.path.join(ROOT, "tests", "golden", "datasets", "cats.txt"),
                        "--scheme", scheme, "--topics", "20", "--iterations", "40", "--seed", "4711", "--out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "23 documents, 303 types, 7788 tokens" in r.stdout          # the bundled corpus
    assert r.stdout.count("topic ") == 20
    from ldagroupedgibbssampler_amd import formats as F
    phi = F.read_binary_double_matrix(20, 303, str(out / "phi_20_303_00040.BINARY"))
    assert np.allclose(phi.sum(axis=1), 1.0, atol=1e-9)
    lines = (out / "log-likelihood.txt").read_text().splitlines()
    assert len(lines) == 40 and lines[0].startswith("1\t-")
    counts = np.loadtxt(out / "type_topic_counts.csv", delimiter=",", dtype=np.int64)
    assert counts.shape == (20, 303) and counts.sum() == 7788
    assert (out / "Theta_DxK_23_20_00040.csv").exists()
