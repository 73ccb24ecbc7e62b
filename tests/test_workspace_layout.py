"""The workspace layout's overlay argument as an executable check (no GPU): tests/host_checks/layout_invariants.cpp includes the
library's own rnnt_host.h (make_layout) and verifies, over a grid of shapes, that the record table can only ever fall on lattice
blocks of EARLIER samples, that a group's records fit the head, where the packed layout's row scales live, and that
get_workspace_size is monotone across the one-group / eight-group switch.  Reference for what the workspace replaces:
/root/reference/src/rnnt_entrypoint.cpp:96-128."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (host-only compile)")
def test_layout_invariants(tmp_path):
    exe = str(tmp_path / "layout_invariants")
    src = os.path.join(ROOT, "tests", "host_checks", "layout_invariants.cpp")
    build = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-x", "hip", src, "-o", exe], capture_output=True, text=True,
                           timeout=900)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "0 failures" in run.stdout, run.stdout[-3000:]


def test_workspace_sizes_through_the_c_abi():
    """What callers see: the long-utterance configuration shrank, small problems kept their size."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
    from warprnnt_pytorch import _lib
    c4 = _lib.workspace_bytes(1500, 301, 64, True, 4)
    ref_c4 = (3 * 1500 * 301 + 2) * 64 * 4                    # the reference's formula
    assert 0.70e9 < c4 < 0.80e9 and c4 < 2.3 * ref_c4          # round 5: 1.18e9 = 3.4x
    assert _lib.workspace_bytes(150, 21, 128, True, 4) < 20e6
    assert _lib.workspace_bytes(200, 41, 1024, True, 2) < 0.32e9
