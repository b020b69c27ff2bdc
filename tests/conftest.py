import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library reads its GGS_DEBUG_* knobs (kernel selection, proof margins: what several tests force) only with this opt-in
    os.environ["GGS_DEBUG"] = "1"


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def native():
    """The product C-ABI wrapper; the .so must already be built (no fallback)."""
    from ldagroupedgibbssampler_amd import native as N
    N._lib.load()
    return N


@pytest.fixture(scope="session")
def cats():
    """Integer encoding of the reference's bundled cats.txt (tests/golden/cats_corpus.npz,
    produced by tests/golden/make_fixtures.py)."""
    import numpy as np
    from ldagroupedgibbssampler_amd.corpus import Corpus
    d = np.load(os.path.join(ROOT, "tests", "golden", "cats_corpus.npz"))
    return Corpus(d["doc_ptr"].astype(np.int64), d["tokens"].astype(np.int32), int(d["num_types"]))
