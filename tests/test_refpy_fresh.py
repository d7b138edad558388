"""Where the reference is present (the build container), re-running tests/golden/make_refpy_golden.py must
reproduce the committed refpy fixtures exactly: they ARE the imported reference's outputs, not hand-edited."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference (not on the GPU box)")
def test_refpy_fixtures_are_reproduced_by_the_imported_reference(tmp_path):
    out = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(G, "make_refpy_golden.py"), "--out", out], cwd=out,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = [f for f in os.listdir(G) if f.startswith("refpy_") and f.endswith(".npz")]
    assert len(names) >= 6
    for f in names:
        a, b = np.load(os.path.join(out, f)), np.load(os.path.join(G, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in b.files:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (f, k)
    na = torch.load(os.path.join(out, "refpy_nets.pt"), weights_only=False)
    nb = torch.load(os.path.join(G, "refpy_nets.pt"), weights_only=False)
    for mod in ("warp", "camera_mlp"):
        assert all(torch.equal(na["v2"][mod][k], nb["v2"][mod][k]) for k in nb["v2"][mod])
