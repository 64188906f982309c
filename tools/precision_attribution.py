"""Where the 16-bit modes lose accuracy: the oracle with the engine's operand
roundings emulated one stage at a time (CPU, test infrastructure only).

    python tools/precision_attribution.py [--frames 1000] [--batch 4]

For each format (bf16, fp16) and each rounding point of the HIP kernels --
feat (gathered features), w (all weights), x0 / x1 / x2 (residual-stream copies
used as GEMM operands; the residual itself stays fp32), qkv, p (softmax
numerators), ao, h (FFN hidden) -- print the max-abs posterior error against the
fp32 oracle with ONLY that stage rounded, then with all of them, on the seeded
and on the sharpened checkpoint.  Mel features of 0.1*randn audio, seed 1234.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ppg_oracle as O          # noqa: E402
from ppgs_amd import weights as W           # noqa: E402

STAGES = ('feat', 'w', 'x0', 'qkv', 'p', 'ao', 'x1', 'h', 'x2')


def rounder(dtype, stages):
    def quant(stage, tensor):
        return tensor.to(dtype).float() if stage in stages else tensor
    return quant


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--frames', type=int, default=1000)
    parser.add_argument('--batch', type=int, default=4)
    args = parser.parse_args()
    generator = torch.Generator().manual_seed(1234)
    audio = 0.1 * torch.randn(args.batch, 1, args.frames * 160, generator=generator)
    feats = O.mel_from_audios(audio)
    lengths = torch.full((args.batch,), args.frames, dtype=torch.long)
    for name, state in (('seeded', W.seeded_state_dict(seed=1234)),
                        ('sharp', W.seeded_state_dict(seed=4321, sharpen=2.0))):
        ref = O.from_features(state, feats, lengths)
        print(f'--- {name} checkpoint: posterior max {float(ref.max()):.3f}, '
              f'mean top-1 {float(ref.amax(1).mean()):.3f}')
        for label, dtype in (('bf16', torch.bfloat16), ('fp16', torch.float16)):
            row = []
            for stage in STAGES:
                out = O.from_features(state, feats, lengths, quant=rounder(dtype, {stage}))
                row.append(f'{stage} {float((out - ref).abs().max()):.2e}')
            out = O.from_features(state, feats, lengths, quant=rounder(dtype, set(STAGES)))
            agree = float((out.argmax(1) == ref.argmax(1)).float().mean())
            print(f'{label}: ' + '  '.join(row))
            print(f'{label}: ALL {float((out - ref).abs().max()):.2e}  argmax agreement {agree:.5f}')


if __name__ == '__main__':
    main()
