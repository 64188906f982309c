"""Fixed cost of a timed block: K steps between synchronize brackets, for several K (the driver's line is K = 20)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppgs_amd
from ppgs_amd import engine as E

state = ppgs_amd.weights.seeded_state_dict(seed=1234)
model = E.Engine(state, 0, 'bf16')
gen = torch.Generator().manual_seed(1234)
audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
lengths = [1000] * 32

def step():
    mel = ppgs_amd.preprocess.mel.from_audios(audio)
    return model.encode(mel, lengths)

for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for _ in range(20): step()
    torch.cuda.synchronize()
res = {}
for rep in range(5):
    for K in (1, 2, 5, 10, 20, 40, 80, 160, 320):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(K): step()
        t_enq = time.perf_counter() - t
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        res.setdefault(K, []).append((dt, t_enq))
for K, v in res.items():
    best = min(x[0] for x in v); med = sorted(x[0] for x in v)[len(v) // 2]
    print(f'K {K:4d}: block {med * 1e3:8.3f} ms (best {best * 1e3:8.3f}), per step {med / K * 1e3:.4f} ms, host enqueue {sorted(x[1] for x in v)[len(v)//2] / K * 1e6:7.1f} us/step')
k1, k2 = 320, 20
a = (sorted(x[0] for x in res[k1])[2] - sorted(x[0] for x in res[k2])[2]) / (k1 - k2)
print(f'steady state {a * 1e3:.4f} ms/step; fixed cost of a 20-step block {(sorted(x[0] for x in res[20])[2] - 20 * a) * 1e6:.0f} us')
# the pieces: an empty synchronize, one frontend alone, a step's enqueue time
torch.cuda.synchronize(); t = time.perf_counter(); torch.cuda.synchronize(); print(f'empty synchronize {(time.perf_counter() - t) * 1e6:.1f} us')
for name, fn in (('frontend only', lambda: ppgs_amd.preprocess.mel.from_audios(audio)),):
    xs = []
    for _ in range(20):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); xs.append(time.perf_counter() - t)
    print(f'{name}: launch + kernel + synchronize {sorted(xs)[10] * 1e6:.1f} us (kernel ~37 us)')
