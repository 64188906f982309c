"""End-to-end file pipeline throughput: WAV files on disk -> .pt PPG files.

    python tools/bench_files.py [--files 256] [--seconds 10] [--workers 8]

Includes everything the hot-path bench leaves out: WAV decode, collate,
pinned H2D, frontend + encoder, D2H, torch.save -- the PCIe- and
filesystem-inclusive rate of ppgs_amd.from_files_to_files.
Synthetic 16 kHz audio, seeded random weights.
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
from scipy.io import wavfile                              # noqa: E402

import ppgs_amd                                           # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--files', type=int, default=256)
    parser.add_argument('--seconds', type=float, default=10.0)
    parser.add_argument('--workers', type=int, default=8)
    parser.add_argument('--max-frames', type=int, default=32000)
    parser.add_argument('--breakdown', action='store_true', help='time the stages of the pipeline on their own as well')
    args = parser.parse_args()
    rng = np.random.default_rng(1234)
    with tempfile.TemporaryDirectory() as root:
        checkpoint = os.path.join(root, 'seeded.pt')
        torch.save(ppgs_amd.weights.seeded_state_dict(seed=1234), checkpoint)
        files, outs, frames = [], [], 0
        for i in range(args.files):
            seconds = args.seconds * rng.uniform(0.5, 1.5)
            samples = int(seconds * 16000)
            audio = (0.1 * rng.standard_normal(samples)).astype(np.float32)
            path = os.path.join(root, f'{i:05d}.wav')
            wavfile.write(path, 16000, audio)
            files.append(path)
            outs.append(os.path.join(root, f'{i:05d}-ppg.pt'))
            frames += samples // 160
        # warm-up (engine creation, plan cache is per batch shape anyway)
        ppgs_amd.from_files_to_files(
            files[:8], outs[:8], checkpoint=checkpoint, num_workers=args.workers,
            gpu=0, max_frames=args.max_frames)
        torch.cuda.synchronize()
        start = time.perf_counter()
        ppgs_amd.from_files_to_files(
            files, outs, checkpoint=checkpoint, num_workers=args.workers, gpu=0,
            max_frames=args.max_frames)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
        sample = torch.load(outs[-1])
        assert sample.shape[0] == 40
        if args.breakdown:
            from ppgs_amd import core, engine as E
            t0 = time.perf_counter()
            dl = core.loader(files, num_workers=max(args.workers // 2, 1), max_frames=args.max_frames, gpu=0)
            t1 = time.perf_counter()
            batches = list(dl)
            t2 = time.perf_counter()
            model = core.engine_for('mel', checkpoint, 0) if hasattr(core, 'engine_for') else None
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for padded, lengths, _ in batches:
                audio = padded.cuda(non_blocking=True)
                mel = ppgs_amd.preprocess.mel.from_audios(audio)
                out = model.encode(mel, [int(n) // 160 for n in lengths])
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            host = [torch.empty((40, 1000)) for _ in range(64)]
            t5 = time.perf_counter()
            E.pt_write_batch([os.path.join(root, f'w{i}.pt') for i in range(64)], torch.zeros(64, 40, 1000), [1000] * 64, threads=max(args.workers // 2, 1)) if hasattr(E, 'pt_write_batch') else None
            t6 = time.perf_counter()
            print(f'breakdown: loader init (headers, packing) {t1 - t0:.3f} s, decode of {len(batches)} batches {t2 - t1:.3f} s, '
                  f'H2D + frontend + encoder {t4 - t3:.3f} s, 64 x (40, 1000) .pt writes {t6 - t5:.4f} s')
    print(f'{args.files} files, {frames} frames in {elapsed:.2f} s -> '
          f'{frames / elapsed / 1e6:.2f} M frames/s end to end '
          f'({args.workers} workers, max_frames {args.max_frames})')


if __name__ == '__main__':
    main()
