import os
import subprocess
import sys

import pytest

# Load order with PyTorch: torch wheels bundle their own libamdhip64; whichever HIP runtime is mapped
# first serves the whole process.  Import torch before libqdrant_amd.so so that both share torch's
# runtime (the other order leaves torch without devices).  See INTEGRATION.md.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: build it on demand (gcc only, a second)
    so = os.path.join(ROOT, "oracle", "libqdrant_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("qdrant_oracle.c", "qdrant_oracle_hnsw.c", "qdrant_oracle_links.c", "qdrant_oracle.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    return oracle_ffi
