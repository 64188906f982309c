"""Multi-rank path on CPU: world_size 2 (and 4, with an empty shard) over gloo.  The per-batch compute is
the CPU oracle here (test infrastructure); sharding, packing, the two gather
collectives and the reassembly are the product code under test."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def worker(rank, port, lengths, result_path):
    os.environ.update(
        MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
        WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import ppg_oracle
    from ppgs_amd import distributed, weights
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    state = weights.seeded_state_dict(seed=1234, num_layers=1)
    generator = torch.Generator().manual_seed(5)
    audios = [0.1 * torch.randn(1, n * 160, generator=generator) for n in lengths]
    seen = []

    def compute(padded, sample_lengths):
        seen.append(padded.shape[0])
        return ppg_oracle.from_audio(state, padded)

    # every rank builds ONLY the utterances of its shard (deterministic per index):
    # touching another rank's audio is an error
    shards = distributed.shard_lpt([__import__('ppgs_amd').data.flops(n) for n in lengths], WORLD)
    touched = []

    def audio_of(index):
        assert index in shards[rank], (rank, index)
        touched.append(index)
        return audios[index]

    out = distributed.from_audios_sharded(audio_of, compute=compute, max_frames=400, frames=list(lengths))
    assert sorted(touched) == sorted(shards[rank])
    # the gatherv by itself: ragged payloads, one rank with nothing to send
    payload = torch.arange(12, dtype=torch.float32).reshape(4, 3) + 100 * rank if rank == 1 else torch.zeros((0, 3))
    gathered = distributed.gather_ragged(payload, [3, 1] if rank == 1 else [])
    if rank == 0:
        assert gathered[0] == [] and [tuple(t.shape) for t in gathered[1]] == [(3, 3), (3, 1)]
        assert torch.equal(gathered[1][0], (torch.arange(9, dtype=torch.float32).reshape(3, 3) + 100).T)
    else:
        assert gathered is None
    if rank == 0:
        torch.save({'out': out, 'batches': seen}, result_path)
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_rank_sharded_inference_matches_single_process(tmp_path):
    lengths = [120, 40, 75, 33, 90, 61, 18]
    result = tmp_path / 'result.pt'
    mp.spawn(worker, args=(free_port(), lengths, str(result)), nprocs=WORLD, join=True)
    got = torch.load(result)
    assert len(got['out']) == len(lengths)

    # single-process reference with the same packing rule per shard
    from oracle import ppg_oracle
    from ppgs_amd import data, distributed, weights
    state = weights.seeded_state_dict(seed=1234, num_layers=1)
    generator = torch.Generator().manual_seed(5)
    audios = [0.1 * torch.randn(1, n * 160, generator=generator) for n in lengths]
    shards = distributed.shard_lpt([data.flops(n) for n in lengths], WORLD)
    assert all(shards)                                    # both ranks had work
    for shard in shards:
        for batch in data.pack_batches([lengths[i] for i in shard], 400):
            indices = [shard[j] for j in batch]
            padded, _ = data.collate([audios[i] for i in indices])
            ref = ppg_oracle.from_audio(state, padded)
            for row, index in enumerate(indices):
                ppg = got['out'][index]
                assert ppg.shape == (40, lengths[index])
                assert np.abs(ppg.numpy() - ref[row, :, :lengths[index]].numpy()).max() < 1e-6


def files_worker(rank, port, root, count):
    os.environ.update(
        MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
        WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    from ppgs_amd import distributed
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    files = [os.path.join(root, f'{i}.wav') for i in range(count)]
    outs = [os.path.join(root, f'{i}.pt') for i in range(count)]

    def runner(mine, outputs):          # stands in for the GPU run of the shard
        for src, dst in zip(mine, outputs):
            torch.save(torch.tensor([rank]), dst)

    distributed.from_files_to_files_sharded(files, outs, runner=runner)
    dist.destroy_process_group()


def test_two_rank_file_sharding(tmp_path):
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    lengths = [16000 * k for k in (1, 7, 3, 2, 9, 4)]
    for i, n in enumerate(lengths):
        wavfile.write(tmp_path / f'{i}.wav', 16000, np.zeros(n, np.float32))
    mp.spawn(files_worker, args=(free_port(), str(tmp_path), len(lengths)), nprocs=WORLD, join=True)
    owners = [int(torch.load(tmp_path / f'{i}.pt')[0]) for i in range(len(lengths))]
    assert set(owners) == {0, 1}                      # every file written exactly once, both ranks used
    from ppgs_amd import data, distributed
    shards = distributed.shard_lpt([data.flops(n // 160) for n in lengths], WORLD)
    assert [owners[i] for i in shards[0]] == [0] * len(shards[0])
    assert [owners[i] for i in shards[1]] == [1] * len(shards[1])


def worker4(rank, port, world, lengths, result_path):
    os.environ.update(
        MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
        WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import ppg_oracle
    from ppgs_amd import data, distributed, weights
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cpus = distributed.bind_cpus(rank, world)
    assert cpus == [] or sorted(os.sched_getaffinity(0)) == cpus
    state = weights.seeded_state_dict(seed=1234, num_layers=1)
    generator = torch.Generator().manual_seed(6)
    audios = [0.1 * torch.randn(1, n * 160, generator=generator) for n in lengths]
    shards = distributed.shard_lpt([data.flops(n) for n in lengths], world)

    def audio_of(index):
        assert index in shards[rank], (rank, index)
        return audios[index]

    def compute(padded, sample_lengths):
        return ppg_oracle.from_audio(state, padded)

    out = distributed.from_audios_sharded(audio_of, compute=compute, max_frames=400, frames=list(lengths))
    if rank == 0:
        torch.save({'out': out, 'empty': [r for r, shard in enumerate(shards) if not shard]}, result_path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_with_an_empty_shard(tmp_path):
    """World size 4 over three utterances: one rank gets nothing (it still takes part in the count exchange and
    sends no payload), the other three one utterance each; rank 0 reassembles all of them in input order."""
    world, lengths = 4, [70, 45, 110]
    result = tmp_path / 'result4.pt'
    mp.spawn(worker4, args=(free_port(), world, lengths, str(result)), nprocs=world, join=True)
    got = torch.load(result)
    assert len(got['empty']) == 1
    from oracle import ppg_oracle
    from ppgs_amd import weights
    state = weights.seeded_state_dict(seed=1234, num_layers=1)
    generator = torch.Generator().manual_seed(6)
    audios = [0.1 * torch.randn(1, n * 160, generator=generator) for n in lengths]
    for index, n in enumerate(lengths):
        ref = ppg_oracle.from_audio(state, audios[index][None])[0]
        assert got['out'][index].shape == (40, n)
        assert np.abs(got['out'][index].numpy() - ref[:, :n].numpy()).max() < 1e-6


# ---- bench.py's own N-rank path: rank_main for `--gpus 2 --workload c4`, end to end over gloo ----

class _StubEngine:
    """Stands in for ppgs_amd.engine.Engine (which has no CPU path): uniform posteriors of the right shape.  What the
    test walks is bench.py's bookkeeping around it, not arithmetic."""

    def __init__(self, state, device, precision):
        self.launches = 0

    def encode(self, mel, lengths):
        self.launches += 5
        return torch.full((mel.shape[0], 40, mel.shape[2]), 1.0 / 40)

    def profile(self, on, classes=None, stride=1):
        self.launches = 0

    def profile_read(self):
        return {'ffn': (0.05 * self.launches, self.launches)}

    def pipelines(self, tokens):
        return 1


def _bench_rank(rank, port, world, out_path):
    import argparse
    import contextlib
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), PPGS_BENCH_BACKEND='gloo')
    torch.set_num_threads(1)
    bench = importlib.import_module('bench')
    import ppgs_amd
    from ppgs_amd import engine as E
    # no GPU here: the device calls rank_main makes are no-ops, the tensors live on the CPU, the engine is the stub
    bench.DEVICE = 'cpu'
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: world
    torch.cuda.set_device = lambda device: None
    torch.cuda.synchronize = lambda *a: None
    ppgs_amd.data.row_budget = lambda max_frames, tile_rows=160, gpu=None: None
    E.Engine = _StubEngine
    ppgs_amd.preprocess.mel.from_audios = lambda audio: torch.zeros(audio.shape[0], 80, audio.shape[-1] // 160)
    args = argparse.Namespace(gpus=world, steps=1, warmup=1, workload='c4', precision='bf16', utterances=48,
                              env_config={}, env_ablation={}, no_cpu=True, no_alt=True)
    if rank == 0:
        with open(out_path, 'w') as f, contextlib.redirect_stdout(f):
            bench.rank_main(args)
    else:
        bench.rank_main(args)


def test_bench_rank_main_c4_two_ranks_over_gloo(tmp_path):
    """The driver's first multi-GPU run must not die in bookkeeping: bench.py's rank_main, `--gpus 2 --workload c4`,
    with a stub engine on CPU tensors over gloo -- LPT shards, per-rank batches, barriers, max-over-ranks, the gatherv to
    rank 0 and every field of the record (`gather_ms`, `gather_bytes`, the world-size field, `rank0_cpus`)."""
    import json
    out = tmp_path / 'line.json'
    mp.spawn(_bench_rank, args=(free_port(), 2, str(out)), nprocs=2, join=True)
    line = json.loads(out.read_text().strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['gloo_world_size'] == 2 and line['scaling'] == 'strong'
    assert line['config']['utterances'] == 48 and line['value'] > 0 and line['gather_ms'] >= 0
    generator = torch.Generator().manual_seed(1234)
    frames = torch.randint(50, 3001, (48,), generator=generator).tolist()
    assert line['gather_bytes'] == sum(frames) * 40 * 4
    assert line['roofline']['timed_launches'] > 0 and 'rank0_cpus' in line
