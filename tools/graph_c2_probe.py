"""C2 (32 x 1000 frames, bf16): the encoder's launch sequence eager against ONE hipGraph replay per step
(Engine.graphed: the two pipelines' fork / join captured with it), the mel frontend before it either way.

    python tools/graph_c2_probe.py          # prints one JSON record
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402


def timed(fn, steps, prewarm_s=1.0):
    end = time.perf_counter() + prewarm_s
    while time.perf_counter() < end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - start) / steps


def main():
    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, 0, 'bf16')
    audio = (0.1 * torch.randn(32, 1, 160000, generator=torch.Generator().manual_seed(1234))).cuda()
    lengths = [1000] * 32
    mel = ppgs_amd.preprocess.mel.from_audios(audio)
    reference = model.encode(mel, lengths).clone()
    run = model.graphed(32, 1000, lengths)
    out = run(mel).clone()
    record = {'graph_equals_eager': bool((out == reference).all()), 'ms_per_step': {}}

    def eager_step():
        model.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths)

    def graph_step():
        run(ppgs_amd.preprocess.mel.from_audios(audio))

    for _ in range(2):
        for name, fn in (('eager', eager_step), ('graph', graph_step), ('eager_encoder_only', lambda: model.encode(mel, lengths)),
                         ('graph_encoder_only', lambda: run())):
            record['ms_per_step'].setdefault(name, []).append(timed(fn, 400))
    print(json.dumps(record))


if __name__ == '__main__':
    main()
