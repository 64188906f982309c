"""Host-side batch scheduler: length bucketing and frame-budget packing.

Counterpart of reference ppgs/data/{dataset,sampler,collate}.py for the
inference path: ``Dataset.buckets`` (dataset.py:109-128), ``Sampler.batch``
(sampler.py:46-82) and ``Collate`` (collate.py:20-27).  Two packers:

* ``mode='reference'`` reproduces the reference sampler bit-for-bit (seeded
  shuffle inside BUCKETS buckets, greedy ``(n+1) * maxlen <= max_frames``,
  shuffled batch order) -- pinned by tests/golden/g8_packing.npz.
* ``mode='sorted'`` (default for inference here) keeps the same budget
  invariant but packs in length order, so batches are length-homogeneous and
  padding is minimal; per-utterance results do not depend on batch order.
"""
import math
import warnings

import numpy as np
import torch

from . import config


def frames_of(num_samples, sample_rate=config.SAMPLE_RATE):
    """Frames of a file (reference ppgs/data/dataset.py:188-190)."""
    return int(num_samples * (config.SAMPLE_RATE / sample_rate)) // config.HOPSIZE


def filter_lengths(lengths, max_frames, names=None):
    """Indices of items that fit the budget; longer ones are skipped with a
    warning (reference ppgs/data/dataset.py:193-198)."""
    keep = []
    for index, length in enumerate(lengths):
        if length <= max_frames:
            keep.append(index)
        else:
            name = names[index] if names is not None else index
            warnings.warn(
                f'File {name} of length {length} '
                f'exceeds max_frames of {max_frames}. Skipping.')
    return keep


def plan_rows(length, frames, chunk=config.CHUNK_LENGTH,
              overlap=config.CHUNK_OVERLAP):
    """Token rows one item of `length` valid frames occupies in the engine's
    plan of a batch padded to `frames` (ppg_plan_windows, the reference's chunk
    rule of transformer.py:49-64): every window of the batch grid that still
    holds a valid frame of the item, each rounded up to 16 rows."""
    if length <= 0:
        return 0
    if frames <= chunk:
        return -(-frames // 16) * 16
    stride = chunk - 2 * overlap
    rows = 0
    for index in range(-(-min(length, frames) // stride)):
        start = index * stride
        rows += -(-(min(start + chunk, frames + overlap) - start) // 16) * 16
    return rows


def row_budget(max_frames, tile_rows=160, gpu=None):
    """Rows per batch for the packer on an engine pipeline: whole rounds of the
    layer kernel's workgroups (one tile of `tile_rows` token rows per CU and
    round; a batch of one tile more than a round costs a second round), as
    many rounds as the frame budget's ~1.25 rows per frame fill.  None without
    a GPU or for an unbounded budget."""
    if not torch.cuda.is_available() or math.isinf(max_frames):
        return None
    device = torch.cuda.current_device() if gpu is None else gpu
    per_round = torch.cuda.get_device_properties(device).multi_processor_count * tile_rows
    return max(1, math.ceil(1.25 * max_frames / per_round)) * per_round


def pack_batches(lengths, max_frames=config.MAX_INFERENCE_FRAMES,
                 mode='sorted', seed=config.RANDOM_SEED, epoch=0,
                 buckets=config.BUCKETS, max_rows=None):
    """Lists of indices such that len(batch) * max(len in batch) <= max_frames
    (a single item longer than the budget forms its own batch, as in the
    reference sampler).  `max_rows` (mode 'sorted' only; see row_budget) also
    bounds the token rows of a batch's plan."""
    lengths = np.asarray(lengths, dtype=np.int64)
    count = len(lengths)
    if count == 0:
        return []
    if mode == 'sorted':
        order = np.argsort(lengths, kind='stable')
        batches, batch, longest, rows = [], [], 0, 0
        for index in order[::-1]:           # longest first: max is the first item
            length = int(lengths[index])
            longest = max(longest, length)
            item_rows = plan_rows(length, longest) if max_rows else 0
            if batch and ((len(batch) + 1) * longest > max_frames or
                          (max_rows and rows + item_rows > max_rows)):
                batches.append(batch)
                batch, longest = [int(index)], length
                rows = plan_rows(length, length) if max_rows else 0
            else:
                batch.append(int(index))
                rows += item_rows
        if batch:
            batches.append(batch)
        return batches
    if mode != 'reference':
        raise ValueError(f'unknown packing mode {mode}')
    # np.argsort's default (unstable) order on ties, as reference
    # Dataset.buckets (dataset.py:115)
    order = np.argsort(lengths)
    size = count // buckets
    pairs = np.stack((order, lengths[order])).T
    bucket_list = [pairs[i:i + size] for i in range(0, count, size)]
    if len(bucket_list) == buckets + 1:
        residual = bucket_list.pop()
        bucket_list[-1] = np.concatenate((bucket_list[-1], residual), axis=0)
    generator = torch.Generator()
    generator.manual_seed(seed + epoch)
    batches = []
    for bucket in bucket_list:
        bucket = bucket[torch.randperm(len(bucket), generator=generator).tolist()]
        batch, longest = [], 0
        for index, length in bucket:
            longest = max(longest, int(length))
            if batch and (len(batch) + 1) * longest > max_frames:
                batches.append(batch)
                batch, longest = [int(index)], int(length)
            else:
                batch.append(int(index))
        if batch:
            batches.append(batch)
    shuffle = torch.randperm(len(batches), generator=generator).tolist()
    return [batches[i] for i in shuffle]


def padding_efficiency(lengths, batches):
    """valid frames / (B * maxlen) summed over batches."""
    lengths = np.asarray(lengths)
    valid = sum(int(lengths[b].sum()) for b in batches)
    padded = sum(len(b) * int(lengths[b].max()) for b in batches)
    return valid / max(padded, 1)


def collate(audios):
    """Zero-pad (1, samples) tensors to (B, 1, maxlen) + sample lengths
    (reference ppgs/data/collate.py:20-27, 49-51)."""
    lengths = torch.tensor([a.shape[-1] for a in audios], dtype=torch.long)
    padded = torch.zeros((len(audios), 1, int(lengths.max())), dtype=torch.float)
    for i, a in enumerate(audios):
        padded[i, 0, :a.shape[-1]] = a[0]
    return padded, lengths


###############################################################################
# Cost model -- used for multi-GPU sharding and bench FLOPs
###############################################################################


def chunk_lengths(frames):
    """Window lengths Tc of a `frames`-long utterance
    (reference ppgs/model/transformer.py:49-64)."""
    chunk, overlap = config.CHUNK_LENGTH, config.CHUNK_OVERLAP
    if frames <= chunk:
        return [frames]
    stride = chunk - 2 * overlap
    return [
        min(i * stride + chunk, frames + overlap) - i * stride
        for i in range(math.ceil(frames / stride))]


def flops(frames, input_channels=config.INPUT_CHANNELS,
          hidden=config.HIDDEN_CHANNELS, ffn=config.FFN_CHANNELS,
          layers=config.NUM_HIDDEN_LAYERS, outputs=config.OUTPUT_CHANNELS,
          kernel=config.KERNEL_SIZE):
    """Algorithmic FLOPs (2 x MAC, GEMM/conv work only) of one utterance."""
    per_frame = (
        2 * input_channels * kernel * hidden +
        layers * (2 * hidden * 3 * hidden + 2 * hidden * hidden +
                  4 * hidden * ffn) +
        2 * hidden * kernel * outputs)
    windows = chunk_lengths(frames)
    attention = layers * 4 * hidden * sum(t * t for t in windows)
    return per_frame * sum(windows) + attention
