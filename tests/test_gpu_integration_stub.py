"""The ctypes stub printed in INTEGRATION.md section 2 -- the binding a maintainer of the reference would add -- is executed as
written (only the library path is made absolute) and compared with the package's own binding."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_integration_md_stub_runs_and_matches(gpu):
    from gyroflow_amd import cv, _lib
    from scipy.spatial.transform import Rotation
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"## 2\. ctypes stub.*?```python\n(.*?)```", text, re.S)
    assert m, "INTEGRATION.md section 2 has no python block"
    code = m.group(1).replace('C.CDLL("libgfremap.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(8)
    n, sh, sw, dh, dw = 3, 120, 160, 90, 140
    frames = torch.from_numpy(rng.integers(0, 256, (n, sh, sw, 3), dtype=np.uint8)).to(gpu)
    K = np.array([[110., 0, 80], [0, 110., 60], [0, 0, 1]]); P = np.array([[100., 0, 70], [0, 100., 45], [0, 0, 1]])
    D = [0.04, -0.01, 0.003, 0.0]
    Rs = np.stack([Rotation.from_euler("zyx", [2 * i, -3 + i, 1.0], degrees=True).as_matrix() for i in range(n)])
    got = ns["undistort_rotate"](frames, K, D, Rs, P, dw, dh, border=1)
    torch.cuda.synchronize()
    xf = cv.xforms_to_device(np.stack([_lib.xform_to_array(_lib.make_frame_xform(K, R, P)) for R in Rs]), gpu)
    want = cv.undistort_rotate(frames, D, xf, (dw, dh), borderMode=1)
    assert torch.equal(got, want)
