"""Probe: does running the two halves of the C2 batch as two independent
pipelines on two HIP streams (windows never interact) beat one stream?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppgs_amd
from ppgs_amd import engine as E

state = ppgs_amd.weights.seeded_state_dict(seed=1234)
g = torch.Generator().manual_seed(1234)
audio = (0.1 * torch.randn(32, 1, 160000, generator=g)).cuda()
for parts in (1, 2, 4):
    engines = [E.Engine(state, 0, 'bf16') for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = list(audio.chunk(parts))
    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for e, s, a in zip(engines, streams, chunks):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                mel = ppgs_amd.preprocess.mel.from_audios(a)
                outs.append(e.encode(mel, [1000] * a.shape[0]))
        for s in streams:
            cur.wait_stream(s)
        return outs
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'parts={parts}: {dt*1e3:.3f} ms/step  {32000/dt/1e6:.2f} Mfps')
