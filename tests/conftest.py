"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs on the build container (no GPU): oracle vs golden vectors, host logic, C-ABI
symbol checks, gloo world_size-2 sharding tests.  `-m gpu` runs on an MI355X: parity of the HIP
engine (called through the C-ABI) against the oracle and the golden vectors.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Per-function golden vectors captured from the compiled reference (oracle/make_golden.py)."""
    with np.load(GOLDEN / "functions.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_edt_standalone():
    """Square-grid cases captured from the reference's STAND-ALONE scatter EDT file,
    Submodule_2/Accelereated_Euclidean_Distance_Transform.c:1,36 (oracle/make_golden.py standalone_edt)."""
    with np.load(GOLDEN / "edt_standalone.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


SA_CASES = ["sq64_sparse", "sq120_dense", "sq200_max", "sq37_single", "sq250_fine", "sq400_max", "sq16_empty"]


@pytest.fixture(scope="session")
def orc():
    import oracle

    oracle.build(ref=False)
    return oracle


def bits(a):
    """Bit pattern view for exact float comparison."""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
