"""Frontend launch time against the batch size: slope = steady-state time per row, intercept = launch + prologue."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from ppgs_amd import engine


def main():
    torch.manual_seed(0)
    for batch in (1, 4, 16, 32, 64, 128):
        audio = torch.randn(batch, 160000, device='cuda') * 0.1
        for _ in range(3):
            engine.frontend(audio)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            engine.frontend(audio)
        b.record()
        torch.cuda.synchronize()
        print(f'batch {batch:4d}: {a.elapsed_time(b) / 50 * 1e3:8.1f} us per launch')


if __name__ == '__main__':
    main()
