"""GPU: the ctypes stub printed in INTEGRATION.md (section B) is extracted from the document and run
against the CPU oracle, so the integration guide cannot rot."""
import os
import re

import numpy as np
import pytest

from conftest import REPO
from new_bloom_filter_repo_amd.synthetic import make_mask

pytestmark = pytest.mark.gpu


def test_integration_md_stub_matches_oracle(oracle):
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = doc[doc.index("```python\nimport ctypes, math, numpy as np"):]
    code = code[len("```python\n"):code.index("```\n")]
    code = code.replace('ctypes.CDLL("librbf_hip.so")',
                        'ctypes.CDLL("%s")' % os.path.join(REPO, "new_bloom_filter_repo_amd", "librbf_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    x = make_mask(3, 57600, 0.06)
    k, l = oracle.optimal_params(len(x), x.sum() / len(x))
    bit_array, witness = ns["gpu_compress_loops"](x, k, l)
    bm, wit, *_ = oracle.compress(x)
    assert np.array_equal(bit_array, bm) and witness == wit
