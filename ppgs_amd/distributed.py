"""Utterance-sharded multi-GPU inference: one process per GPU.

The path shards by utterance with no data-path collective (utterances, and
even their windows, are independent; weights are replicated).  The only
exchange is the final gather of results to rank 0 over
``torch.distributed`` (backend 'nccl' = RCCL over xGMI on the GPU box,
'gloo' in the CPU tests): one all_gather of per-rank frame counts, then a
gatherv of the (sum frames, 40) payloads -- grouped point-to-point sends, only
rank 0 receives, each peer over its own link.  Not present in the reference,
which is single-device.
"""
import os

import torch
import torch.distributed as dist

from . import config, data


def shard_lpt(costs, world_size):
    """Longest-processing-time-first assignment: list of index lists, one per
    rank; deterministic (ties broken by index)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for index in order:
        rank = min(range(world_size), key=lambda r: (loads[r], r))
        shards[rank].append(index)
        loads[rank] += costs[index]
    return shards


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpu_slice(local_rank, local_world, allowed, gpu_nodes, node_cpus):
    """The cores rank `local_rank` of `local_world` is pinned to (pure function; bind_cpus feeds it from sysfs).

    allowed: cores the process may use; gpu_nodes[r]: NUMA node of rank r's GPU (-1 unknown); node_cpus[n]: cores of
    node n.  A rank whose GPU's node is known gets an equal slice of that node's allowed cores, shared with the other
    ranks whose GPUs hang off the same node; otherwise (unknown node, or a node with fewer than two cores per rank) an
    equal slice of all allowed cores.  Empty list: leave the affinity alone."""
    allowed = sorted(allowed)
    local_world = max(int(local_world), 1)
    if len(allowed) < 2 * local_world:
        return []
    mine = gpu_nodes[local_rank] if local_rank < len(gpu_nodes) else -1
    if mine >= 0 and mine in node_cpus:
        ranks = [r for r in range(local_world) if r < len(gpu_nodes) and gpu_nodes[r] == mine]
        cpus = sorted(set(node_cpus[mine]) & set(allowed))
        if len(cpus) >= 2 * len(ranks):
            per = len(cpus) // len(ranks)
            slot = ranks.index(local_rank)
            return cpus[slot * per:(slot + 1) * per]
    per = len(allowed) // local_world
    return allowed[local_rank * per:(local_rank + 1) * per]


def gpu_numa_node(bdf, sysfs='/sys'):
    """NUMA node of the PCI device `bdf` (dddd:bb:dd.f), -1 when sysfs does not tell."""
    try:
        with open(os.path.join(sysfs, 'bus', 'pci', 'devices', bdf, 'numa_node')) as handle:
            return int(handle.read())
    except (OSError, ValueError):
        return -1


def numa_cpus(node, sysfs='/sys'):
    try:
        with open(os.path.join(sysfs, 'devices', 'system', 'node', f'node{node}', 'cpulist')) as handle:
            return _cpulist(handle.read())
    except (OSError, ValueError):
        return set()


def bind_cpus(local_rank, local_world, device=None, sysfs=None, bdfs=None, apply=True):
    """Pin this rank's process (its launch thread, and the reader / writer threads it spawns later) to its
    own slice of host cores: the cores of the NUMA node its GPU hangs off when sysfs tells, split among the
    ranks that share the node -- otherwise an equal slice of the cores the process may use.  Eight ranks
    each running a ~25-launch / 0.7 ms Python loop plus I/O threads on one box otherwise migrate over all
    cores and across sockets.  Returns the sorted core list (empty: nothing was changed).
    PPGS_AMD_BIND_CPUS=0 disables.  `sysfs` (default /sys, or PPGS_AMD_SYSFS_ROOT), `bdfs` (the PCI addresses of
    the ranks' GPUs, default from the HIP device properties) and `apply=False` exist for tests/test_host.py, which
    runs the function against a faked 2-node x 4-GPU tree."""
    if os.environ.get('PPGS_AMD_BIND_CPUS', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return []
    sysfs = sysfs or os.environ.get('PPGS_AMD_SYSFS_ROOT', '/sys')
    allowed = sorted(os.sched_getaffinity(0))
    local_world = max(int(local_world), 1)
    if bdfs is None:
        bdfs = []
        if torch.cuda.is_available():
            count = torch.cuda.device_count()
            for rank in range(local_world):
                try:
                    props = torch.cuda.get_device_properties((rank if device is None else device) % count)
                    bdfs.append(f'{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0')
                except (AttributeError, RuntimeError, AssertionError):
                    bdfs.append(None)
    gpu_nodes = [gpu_numa_node(bdf, sysfs) if bdf else -1 for bdf in bdfs]
    node_cpus = {node: numa_cpus(node, sysfs) for node in set(gpu_nodes) if node >= 0}
    if sysfs != '/sys' and node_cpus:
        # (a faked tree describes a machine that is not this one: its core numbers are the universe)
        allowed = sorted(set().union(*node_cpus.values()))
    cpus = cpu_slice(local_rank, local_world, allowed, gpu_nodes, node_cpus)
    if cpus and apply:
        os.sched_setaffinity(0, cpus)
    return cpus


def init(backend=None):
    """Initialise the default process group from the torchrun environment
    and bind this rank to its GPU.  Returns (rank, world_size)."""
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world_size > 1:
        bind_cpus(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world_size)))
    if world_size > 1 and not dist.is_initialized():
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend, rank=rank, world_size=world_size)
    return rank, world_size


def gather_ragged(local, frame_counts_local, dst=0):
    """Gather per-rank packed results to rank `dst` (a gatherv: only `dst`
    receives, and exactly the bytes each rank holds).

    local: (frames_local, C) tensor holding this rank's utterances
    back-to-back (frame-major); frame_counts_local: list of their lengths.
    Returns on rank dst a list (one per rank) of lists of (C, frames)
    tensors, None elsewhere.  One small all_gather_object of the frame counts,
    then one point-to-point transfer per non-empty peer: grouped
    ncclSend / ncclRecv on the GPU box (each peer on its own xGMI link to
    `dst`), isend / irecv under gloo.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    if world_size == 1:
        return [_split(local, frame_counts_local)]
    rank = dist.get_rank()
    counts = [None] * world_size
    dist.all_gather_object(counts, [int(c) for c in frame_counts_local])
    totals = [sum(c) for c in counts]
    local = local.contiguous()
    if dist.get_backend() != 'nccl' and local.is_cuda:
        local = local.cpu()                 # (gloo moves host memory: the aliased-GPU dry run of bench.py)
    ops, buffers = [], {}
    if rank == dst:
        for src in range(world_size):
            if src == dst or totals[src] == 0:
                continue
            buffers[src] = torch.empty(
                (totals[src], local.shape[1]), dtype=local.dtype, device=local.device)
            ops.append(dist.P2POp(dist.irecv, buffers[src], src))
    elif totals[rank] > 0:
        ops.append(dist.P2POp(dist.isend, local, dst))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    if rank != dst:
        return None
    out = []
    for src in range(world_size):
        if src == dst:
            out.append(_split(local, counts[src]))
        elif totals[src] == 0:
            out.append([])
        else:
            out.append(_split(buffers[src], counts[src]))
    return out


def _split(packed, counts):
    out, offset = [], 0
    for count in counts:
        out.append(packed[offset:offset + count].T.contiguous())
        offset += count
    return out


def from_audios_sharded(audios, compute=None, gpu=None, max_frames=32000,
                        checkpoint=None, representation=config.REPRESENTATION,
                        frames=None):
    """PPGs of a list of (1, samples) utterances using every rank.

    The utterances are LPT-assigned by the algorithmic FLOPs of their chunked
    forward; rank r packs its shard into padded batches under `max_frames`,
    runs them, and the results are gathered to rank 0, which returns a list of
    (40, frames) tensors in input order (other ranks return None).

    `audios` is either the list itself (every rank passes the same list) or,
    with `frames` (the frame count of every utterance, known to all ranks), a
    callable ``audios(index) -> (1, samples)`` that a rank calls ONLY for the
    utterances of its own shard -- no rank then ever holds another rank's
    audio.  `compute(padded_audio, sample_lengths) -> (B, 40, T)` is the
    per-batch forward; by default the HIP engine.
    """
    rank = dist.get_rank() if dist.is_initialized() else 0
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    if frames is None:
        sequence = audios
        frames = [a.shape[-1] // config.HOPSIZE for a in sequence]

        def audios(index):
            return sequence[index]
    elif not callable(audios):
        raise TypeError('with frames=, audios must be a callable index -> audio')
    shards = shard_lpt([data.flops(f) for f in frames], world_size)
    mine = shards[rank]
    if compute is None:
        from . import core, preprocess

        def compute(padded, lengths):
            features = preprocess.mel.from_audios(padded, lengths, gpu=gpu)
            return core.from_features(
                features, lengths // config.HOPSIZE,
                representation=representation, checkpoint=checkpoint, gpu=gpu)
    results = {}
    for batch in data.pack_batches([frames[i] for i in mine], max_frames, max_rows=data.row_budget(max_frames, gpu=gpu)):
        indices = [mine[j] for j in batch]
        padded, lengths = data.collate([audios(i)[:1] for i in indices])
        out = compute(padded, lengths)
        for row, index in enumerate(indices):
            results[index] = out[row, :, :frames[index]]
    channels = config.OUTPUT_CHANNELS
    if mine:
        local = torch.cat([results[i].T for i in mine], dim=0)
    else:
        device = 'cuda' if (torch.cuda.is_available() and (
            not dist.is_initialized() or dist.get_backend() == 'nccl')) else 'cpu'
        local = torch.zeros((0, channels), device=device)
    gathered = gather_ragged(local.contiguous(), [frames[i] for i in mine])
    if gathered is None:
        return None
    ordered = [None] * len(frames)
    for r, shard in enumerate(shards):
        for index, ppg in zip(shard, gathered[r]):
            ordered[index] = ppg
    return ordered


def from_files_to_files_sharded(audio_files, output_files, runner=None, **kwargs):
    """File-to-file inference with every rank: the file list is LPT-sharded
    by chunk-aware FLOP cost (lengths from the WAV headers), rank r runs
    ``ppgs_amd.from_files_to_files`` on its shard and writes its own outputs.
    No collective is needed (reference ppgs/core.py:207-272 is single
    device); a barrier at the end makes completion global.  ``runner(files,
    outputs)`` replaces the per-rank call (tests)."""
    from . import load
    rank = dist.get_rank() if dist.is_initialized() else 0
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    frames = []
    for file in audio_files:
        samples, rate = load.info(file)
        frames.append(data.frames_of(samples, rate))
    shards = shard_lpt([data.flops(max(f, 1)) for f in frames], world_size)
    mine = shards[rank]
    files = [audio_files[i] for i in mine]
    outputs = [output_files[i] for i in mine]
    if files:
        if runner is not None:
            runner(files, outputs)
        else:
            from . import core
            core.from_files_to_files(files, outputs, **kwargs)
    if dist.is_initialized():
        dist.barrier()
    return mine
