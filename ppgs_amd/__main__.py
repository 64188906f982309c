"""python -m ppgs_amd: inference CLI with the reference's flags
(ppgs/__main__.py:12-59)."""
import argparse
from pathlib import Path

import ppgs_amd


def parse_args():
    parser = argparse.ArgumentParser(
        description='Phonetic posteriorgram inference')
    parser.add_argument(
        '--audio_files', nargs='+', type=Path, required=True,
        help='Paths to audio files')
    parser.add_argument(
        '--output_files', type=Path, required=True, nargs='+',
        help='The one-to-one corresponding output files')
    parser.add_argument(
        '--representation', type=str, default=ppgs_amd.REPRESENTATION,
        help='Representation to use for inference')
    parser.add_argument('--checkpoint', type=Path, help='The checkpoint file')
    parser.add_argument(
        '--num-workers', type=int, default=0,
        help='Number of CPU threads for multiprocessing')
    parser.add_argument(
        '--gpu', type=int,
        help='The index of the GPU to use for inference. '
             'Defaults to the current HIP device.')
    parser.add_argument(
        '--max-frames', type=float, default=ppgs_amd.MAX_INFERENCE_FRAMES,
        help='Maximum number of frames in a batch')
    parser.add_argument(
        '--legacy-mode', action='store_true',
        help='Use legacy (unchunked) inference')
    return parser.parse_args()


if __name__ == '__main__':
    ppgs_amd.from_files_to_files(**vars(parse_args()))
