"""CPU: the tile enumeration of the all-pairs kernels, checked on the host before any launch depends on it.

csrc/gemm_hl7.h walks the upper triangle of a symmetric problem on 256 x 192 tiles in an L2-friendly order (groups of tile
rows, ragged head, XCD interleave of the block ids, and - round 5 - a cache of the previous tile's group); a tile left out is
a silently missing minimum, a tile outside the triangle wasted work or an out-of-range block.  tests/native/tile_order_check.hip
compiles the kernel header's own (host + device) functions for the host and enumerates whole launches: every id inside the
triangle, every needed tile exactly once, the cache irrelevant to the answer; the 256 x 128 order of csrc/gemm_hl2.h as well.
Sizes: one row, below one tile, tile edges +-1, ragged (5 493 rows), 28 / 83 images of 784 patches (config 2), and the 585-image
sub-block of config 5 (1792 x 2389 tiles)."""
import os
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_every_tile_of_the_triangle_exactly_once(tmp_path):
    src = os.path.join(ROOT, "tests", "native", "tile_order_check.hip")
    exe = str(tmp_path / "tile_order_check")
    inc = os.path.join(ROOT, "anomaly-clustering_amd", "csrc")
    r = subprocess.run([HIPCC, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", inc, src, "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    sizes = [1, 100, 191, 192, 193, 255, 256, 257, 784, 5493, 28 * 784, 83 * 784, 585 * 784]
    r = subprocess.run([exe] + [str(s) for s in sizes], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert [ln.split()[1] for ln in lines if ln.startswith("ok")] == [str(s) for s in sizes]
