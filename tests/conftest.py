import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "slam-process_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the reference's Chinese axis labels have no glyphs in the image's default font; the raster is unaffected
    config.addinivalue_line("filterwarnings", "ignore:Glyph .* missing from font:UserWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "debugDoc"))


needs_reference = pytest.mark.skipif(not has_reference(), reason="reference tree not present on this machine")
