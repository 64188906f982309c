"""How long does the HOST take to enqueue one benchmark step when 8 rank processes share a box?

VERDICT r4 item 9: the N-rank path has only ever run with its ranks aliased onto one GPU's queues, which says
nothing about eight Python launch loops contending for host cores.  This measures exactly that part, without
needing eight GPUs: every rank process enqueues the C2-shaped launch sequence (mel frontend + two-pipeline encode of
its own batch: ~26 launches, two fork / join event pairs) WITHOUT waiting for the GPU, so the time per step is
the host's -- Python, ctypes, the engine's C++, the HIP runtime's packet writes -- until the queue would back up
(the steps per measurement are few enough that it does not).  The GPU work itself is tiny (2 x 160 frames per rank),
so eight ranks on the one GPU of the pool do not turn the measurement into a GPU measurement.

    python tools/host_contention.py [--ranks 1 8] [--steps 150]

Per rank count and CPU binding (distributed.bind_cpus: the rank's slice of the cores / of its GPU's NUMA node; off:
the scheduler's choice over all cores): host microseconds per step, per rank (min / median / max), beside the
0.69 ms the GPU needs for the real step -- the host has to stay well under that for the GPUs to stay fed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, bind, steps, queue, barrier):
    import torch
    import ppgs_amd
    from ppgs_amd import distributed, engine as E
    cpus = distributed.bind_cpus(rank, world, device=0) if bind else []
    torch.cuda.set_device(0)
    state = ppgs_amd.weights.seeded_state_dict(seed=1234)
    model = E.Engine(state, 0, 'bf16')
    audio = (0.1 * torch.randn(2, 1, 25600, generator=torch.Generator().manual_seed(rank))).cuda()
    lengths = [160, 160]

    def step():
        mel = ppgs_amd.preprocess.mel.from_audios(audio)
        return model.encode(mel, lengths)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    results = []
    for rep in range(5):
        barrier.wait()                       # all ranks enqueue at the same time
        start, cpu0 = time.perf_counter(), time.thread_time()
        for _ in range(steps):
            step()
        host, cpu = time.perf_counter() - start, time.thread_time() - cpu0
        torch.cuda.synchronize()
        total = time.perf_counter() - start
        results.append((1e6 * host / steps, 1e6 * total / steps, 1e6 * cpu / steps))
    results.sort()
    queue.put({'rank': rank, 'cpus': len(cpus), 'host_us_per_step': results[len(results) // 2][0],
               'with_gpu_us_per_step': results[len(results) // 2][1], 'cpu_us_per_step': results[len(results) // 2][2]})


def main():
    import torch.multiprocessing as mp
    parser = argparse.ArgumentParser()
    parser.add_argument('--ranks', type=int, nargs='+', default=[1, 8])
    parser.add_argument('--steps', type=int, default=40)
    args = parser.parse_args()
    ctx = mp.get_context('spawn')
    out = {'what': 'host time to ENQUEUE one step (mel frontend + encode of 2 x 160 frames: the same launch sequence as the benchmark step), '
                   'no synchronisation inside the timed loop; N rank processes on one box sharing GPU 0',
           'logical_cores': os.cpu_count(), 'steps_per_measurement': args.steps, 'runs': []}
    for world in args.ranks:
        for bind in (True, False):
            queue, barrier = ctx.Queue(), ctx.Barrier(world)
            procs = [ctx.Process(target=worker, args=(r, world, bind, args.steps, queue, barrier)) for r in range(world)]
            for p in procs:
                p.start()
            rows = [queue.get(timeout=600) for _ in procs]
            for p in procs:
                p.join()
            host = sorted(r['host_us_per_step'] for r in rows)
            run = {'ranks': world, 'bind_cpus': bind, 'cpus_per_rank': rows[0]['cpus'],
                   'host_us_per_step': {'min': host[0], 'median': host[len(host) // 2], 'max': host[-1]},
                   # CPU time of the launching thread (time.thread_time): what the enqueue COSTS the host, without the time
                   # it is blocked on a queue that the one shared GPU drains N times slower than N GPUs would
                   'cpu_us_per_step_median': sorted(r['cpu_us_per_step'] for r in rows)[len(rows) // 2],
                   'with_gpu_us_per_step_max': max(r['with_gpu_us_per_step'] for r in rows)}
            out['runs'].append(run)
            print(json.dumps(run), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
