"""The ctypes binding stub printed in INTEGRATION.md section 2 is executed as written (from the repo root) and its
`conv3x3_nhwc` is checked against torch's conv2d -- the document cannot drift from the ABI."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_binding_stub_runs():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "class umr_gemm_desc" in b]
    assert len(stub) == 1
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)   # the stub loads "unmore_amd/lib/libumr.so" relative to the repo root
    try:
        exec(stub[0], ns)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(0)
    nb, H, W, ci, co = 2, 24, 40, 64, 128
    x = torch.randn(nb, H, W, ci, generator=g).cuda().bfloat16()
    w = (torch.randn(co, ci, 3, 3, generator=g) * (9 * ci) ** -0.5).cuda().bfloat16()
    bias = torch.randn(co, generator=g).cuda()
    wp = w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()     # [co][ky][kx][ci], as include/umr.h documents
    y = ns["conv3x3_nhwc"](x, wp, bias)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(y.float(), ref, atol=3e-2, rtol=3e-2)
