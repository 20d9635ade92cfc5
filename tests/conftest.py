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


def pytest_generate_tests(metafunc):
    """Every GPU test runs twice: with the fused decode + correct kernel (the batched calls' default) and with the two
    separate kernels.  Contexts read SLAMHIP_FUSED when they are created, so the pipelines a test builds follow it."""
    if metafunc.definition.get_closest_marker("gpu") is not None:
        metafunc.fixturenames.append("_kernel_mode")
        metafunc.parametrize("_kernel_mode", ["fused", "split"], indirect=True)


@pytest.fixture
def _kernel_mode(request, monkeypatch):
    monkeypatch.setenv("SLAMHIP_FUSED", "1" if request.param == "fused" else "0")
    from slamhip import batch
    batch.clear_cache()
    yield request.param
    batch.clear_cache()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "debugDoc"))


needs_reference = pytest.mark.skipif(not has_reference(), reason="reference tree not present on this machine")


def png_axes_box(img):
    """(left, right, top, bottom) of a heatmap figure's axes in pixels, from its four spines: each is a dark line two
    pixels wide centred on the axes edge (the colour bar's spines come after the mesh's)."""
    import numpy as np
    g = img[..., :3].astype(int).sum(-1)
    dark = g < 200
    rows = np.flatnonzero(dark.sum(1) > img.shape[1] // 3)
    cols = np.flatnonzero(dark.sum(0) > img.shape[0] // 3)
    pairs_c = [c for c in cols if c + 1 in cols]
    pairs_r = [r for r in rows if r + 1 in rows]
    return float(pairs_c[0] + 1), float(pairs_c[1] + 1), float(pairs_r[0] + 1), float(pairs_r[-1] + 1)


def png_cell_centres(box, x_edges, y_edges):
    """Pixel (column, row) index arrays of the mesh's cell centres inside the axes box."""
    import numpy as np
    left, right, top, bottom = box
    xe, ye = np.asarray(x_edges), np.asarray(y_edges)
    xc, yc = (xe[:-1] + xe[1:]) / 2, (ye[:-1] + ye[1:]) / 2
    px = np.floor(left + (xc - xe[0]) / (xe[-1] - xe[0]) * (right - left)).astype(int)
    py = np.floor(bottom - (yc - ye[0]) / (ye[-1] - ye[0]) * (bottom - top)).astype(int)
    return px, py


def mesh_cells_agree(got_rgb, want_rgb):
    """Cell colours sampled from two renderings of one mesh: equal everywhere except along the few rows and columns
    a dashed grid line of the figure runs through (the reference draws ax.grid over the mesh, excel_heatmap_v3.py:464).
    Returns (fraction equal, rows with differences, columns with differences)."""
    import numpy as np
    same = (np.asarray(got_rgb).astype(int) == np.asarray(want_rgb).astype(int)).all(axis=-1)
    bad_rows = np.flatnonzero((~same).sum(1) > 3)
    bad_cols = np.flatnonzero((~same).sum(0) > 3)
    rest = same.copy()
    rest[bad_rows, :] = True
    rest[:, bad_cols] = True
    return float(same.mean()), bad_rows, bad_cols, bool(rest.all())
