"""The containment soak of tests/test_gpu_soak.py at 10 x its length (one-off record, not part of the suite):
two-pipeline C2 steps in bf16 and fp16, and the fp16x2 mode at both geometries, beside SDPA on another stream; every
step must be bit-equal to the quiet result.

    python tools/long_soak.py [steps]          # default 20000 (fp16x2: a third of it)
"""
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_soak as soak                        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
soak.STEPS = steps
patch = pytest.MonkeyPatch()
for precision in ('bf16', 'fp16'):
    start = time.perf_counter()
    soak.test_soak_encode_two_pipelines_beside_sdpa(patch, precision)
    print(f'{precision}: {steps} two-pipeline C2 steps beside SDPA, 0 differ from the quiet one-pipeline result ({time.perf_counter() - start:.0f} s)', flush=True)
patch.undo()
