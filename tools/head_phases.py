"""Phase stamps of the head kernel at 32 x 1000 frames, from a timing build (make VARIANT=timing EXTRA=-DPPG_H32_TIMING;
PPGS_AMD_LIB=ppgs_amd/libppgs_amd_timing.so PPGS_AMD_H32_TIMING=1 python tools/head_phases.py [precision]):
the engine prints workgroup 0's s_memtime differences per wave when it is destroyed."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
state = ppgs_amd.weights.seeded_state_dict(seed=1)
model = E.Engine(state, 0, precision)
feats = torch.randn(32, 80, 1000).half().cuda()
lengths = [1000] * 32
for _ in range(50):
    model.encode(feats, lengths)
torch.cuda.synchronize()
model.profile(True)
for _ in range(50):
    model.encode(feats, lengths)
torch.cuda.synchronize()
for name, value in model.profile_read().items():
    print(f'{name}: {value[0] / max(value[1], 1) * 1e3:.1f} us x {value[1] / 50:.0f} per step')
del model
