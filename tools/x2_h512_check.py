"""fp16x2 at the w2v2fb geometry (768 input channels, hidden 512, two heads of 256), configs[2] size (16 x 1000 frames):
distance of the posteriors from the fp32 mode's and from the CPU oracle's (two utterances), and the time of the
PPG-network leg in fp32 / fp16x2 / fp16 / bf16.

    python tools/x2_h512_check.py          # prints one JSON record
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


def timed(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - start) / steps


def main():
    from oracle import ppg_oracle as O
    state = ppgs_amd.weights.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
    feats = torch.randn(16, 768, 1000, generator=torch.Generator().manual_seed(3)).half().cuda()
    lengths = [1000] * 16
    record = {'config': 'configs[2] PPG-network leg: (16, 768, 1000) fp16 features -> hidden-512 network', 'ms_per_step': {}}
    outputs = {}
    for precision in ('fp32', 'fp16x2', 'fp16', 'bf16'):
        model = E.Engine(state, 0, precision)
        outputs[precision] = model.encode(feats, lengths).float().cpu()
        record['ms_per_step'][precision] = timed(lambda: model.encode(feats, lengths))
        if precision == 'fp16x2':       # HIP-event time per kernel class (profile mode serialises the two pipelines' launches)
            model.profile(True)
            for _ in range(5):
                model.encode(feats, lengths)
            torch.cuda.synchronize()
            record['fp16x2_kernel_ms_per_step_stream_summed'] = {n: v[0] / 5 for n, v in model.profile_read().items()}
            model.profile(False)
        del model
    reference = O.from_features(state, feats[:2].cpu().float(), torch.tensor(lengths[:2]))
    record['max_abs_vs_oracle_2_utterances'] = {p: float((o[:2] - reference).abs().max()) for p, o in outputs.items()}
    record['max_abs_vs_fp32_mode'] = {p: float((o - outputs['fp32']).abs().max()) for p, o in outputs.items() if p != 'fp32'}
    record['finite'] = {p: bool(torch.isfinite(o).all()) for p, o in outputs.items()}
    print(json.dumps(record))


if __name__ == '__main__':
    main()
