import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "syke-pic_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """The C-ABI library is built in-tree by __graft_entry__.build(); a checkout that has not been built yet
    (the .so is git-ignored) builds it once here (hipcc cross-compiles gfx950 without a GPU)."""
    so = ROOT / "syke-pic_amd" / "sykepic_hip" / "libsykepic_hip.so"
    if not so.is_file():
        import subprocess
        subprocess.run(["bash", str(ROOT / "syke-pic_amd" / "csrc" / "build.sh")], check=True)
    return so
