"""Per-workgroup timeline of linear_kernel from a PPG_LIN_TIMING build.

    make -C ppgs_amd/csrc EXTRA=-DPPG_LIN_TIMING          (after touching ppg_device.h)
    PPGS_AMD_LIN_TIMING=2 PPGS_AMD_LIN_TIMING_OUT=gpurun_out/lin.bin python bench.py --no-cpu --steps 1 --warmup 0
    python tools/lin_timing.py gpurun_out/lin.bin

PPGS_AMD_LIN_TIMING = kernel class (2 = Q/K/V projection, 4 = out-proj + LN), layer 0.
Stamps (s_memtime of thread 0): 0 start, 1 prologue issued, 2 first tile ready,
3+2s MFMAs of step s issued, 4+2s tile s+1 ready, 13 epilogue issued, 14 stores retired.
"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
live = raw[raw[:, 0] > 0]
t = live[:, :15].astype(np.int64)
dur = t[:, 14] - t[:, 0]
print(f'{len(live)} workgroups; duration (ticks) pct 0/50/90/100:', np.percentile(dur, [0, 50, 90, 100]).astype(int))
prev = 0
for k in range(1, 15):
    ok = (t[:, k] > 0) & (t[:, prev] > 0)
    if not ok.any():
        continue
    d = t[ok, k] - t[ok, prev]
    label = {1: 'prologue issue', 2: 'first tile wait', 13: 'epilogue issue', 14: 'store drain'}.get(
        k, f'step {(k - 3) // 2} mfma' if k % 2 else f'step {(k - 4) // 2} wait')
    print(f'  {label:18s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}')
    prev = k
