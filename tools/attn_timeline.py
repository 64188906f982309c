"""Timeline of one attention launch from a PPG_ATTN_TIMING build:
    PPGS_AMD_ATTN_TIMING=1 PPGS_AMD_ATTN_TIMING_OUT=gpurun_out/attn.bin python bench.py --no-cpu --no-alt --steps 1 --warmup 0
    python tools/attn_timeline.py gpurun_out/attn.bin
Per workgroup: start, end (s_memrealtime, 10 ns ticks, chip-wide), valid keys, HW_ID."""
import sys
import numpy as np
rec = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
rec = rec[rec[:, 1] > 0]
t0 = rec[:, 0].min()
start = (rec[:, 0] - t0).astype(np.int64) / 100.0      # us
end = (rec[:, 1] - t0).astype(np.int64) / 100.0
valid = (rec[:, 2] & 0xffff).astype(np.int64)
pro = ((rec[:, 2] >> 16) & 0xffffff).astype(np.int64) / 100.0      # prologue: start -> first tile-loop iteration
loop = ((rec[:, 2] >> 40) & 0xffffff).astype(np.int64) / 100.0 - pro   # the tile loop
dur = end - start
print(f'{len(rec)} workgroups; launch span {end.max():.1f} us')
for v in np.unique(valid):
    m = valid == v
    print(f'  valid {v:4d}: {m.sum():4d} WGs  duration p10/50/90 {np.round(np.percentile(dur[m], [10, 50, 90]), 1)} us  '
          f'start p10/50/90 {np.round(np.percentile(start[m], [10, 50, 90]), 1)}  end p50/max {np.median(end[m]):.1f} {end[m].max():.1f}')
    print(f'              prologue p10/50/90 {np.round(np.percentile(pro[m], [10, 50, 90]), 2)} us  tile loop {np.round(np.percentile(loop[m], [10, 50, 90]), 2)} us  '
          f'epilogue {np.round(np.percentile((dur - pro - loop)[m], [10, 50, 90]), 2)} us')
edges = np.arange(0, end.max() + 2, 2.0)
running = [(int(((start < t + 1) & (end > t + 1)).sum())) for t in edges]
print('workgroups resident every 2 us:', running)
