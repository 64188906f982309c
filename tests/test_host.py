"""Host-side logic on CPU: packing scheduler, cost model, checkpoint reader,
audio ingest, CLI flags, LPT sharding."""
import math
import subprocess
import sys

import numpy as np
import pytest
import torch

import ppgs_amd
from ppgs_amd import data, distributed, load, weights


def test_reference_packing_mode_matches_reference_sampler(golden):
    g = golden('g8_packing')
    lens = g['lengths']
    for tag, max_frames in (('32000', 32000), ('inf', math.inf)):
        batches = data.pack_batches(lens, max_frames, mode='reference')
        assert [len(b) for b in batches] == list(g[f'batches_{tag}_sizes'])
        assert np.array_equal(np.concatenate(batches), g[f'batches_{tag}_flat'])


def test_sorted_packing_budget_and_efficiency(golden):
    lens = golden('g8_packing')['lengths']
    batches = data.pack_batches(lens, 32000)
    assert sorted(i for b in batches for i in b) == list(range(len(lens)))
    for b in batches:
        assert len(b) * lens[b].max() <= 32000
    reference = data.pack_batches(lens, 32000, mode='reference')
    assert data.padding_efficiency(lens, batches) > 0.9
    assert data.padding_efficiency(lens, batches) > data.padding_efficiency(lens, reference)
    # an item longer than the budget forms its own batch
    assert data.pack_batches([10, 5000, 20], 1000) == [[1], [2, 0]]
    assert data.pack_batches([], 1000) == []


def test_filter_lengths_warns_like_reference():
    with pytest.warns(UserWarning, match='exceeds max_frames'):
        keep = data.filter_lengths([10, 99, 50], 60, ['a', 'b', 'c'])
    assert keep == [0, 2]


def test_collate_zero_pads():
    audios = [torch.ones(1, 5), 2 * torch.ones(1, 3)]
    padded, lengths = data.collate(audios)
    assert padded.shape == (2, 1, 5) and lengths.tolist() == [5, 3]
    assert padded[1, 0].tolist() == [2, 2, 2, 0, 0]


def test_cost_model_matches_survey_numbers():
    assert data.chunk_lengths(1000) == [500, 500, 250]
    assert data.flops(1000) == 19648000000
    assert data.flops(100) == 13414400 * 100 + 5120 * 100 * 100
    assert data.flops(1000, input_channels=768, hidden=512) == \
        35594240 * 1250 + 10240 * (500 ** 2 * 2 + 250 ** 2)


def test_state_dict_layout_and_checkpoint_roundtrip(tmp_path):
    state = weights.seeded_state_dict(seed=3)
    assert sum(v.numel() for k, v in state.items() if k != 'position.encoding') == 6729256
    assert weights.geometry(state) == (80, 256, 5)
    bare, wrapped = tmp_path / 'bare.pt', tmp_path / 'wrapped.pt'
    torch.save(state, bare)
    torch.save({'model': state, 'step': 7}, wrapped)
    for path in (bare, wrapped):
        loaded = load.state_dict(str(path), 'mel')
        assert all(torch.equal(loaded[k], state[k]) for k in state)
    with pytest.raises(ValueError):
        load.state_dict(str(bare), 'w2v2fb')          # geometry mismatch
    with pytest.raises(ValueError):
        load.state_dict(str(bare), 'bottleneck')      # reference load.py:44-47
    broken = dict(state)
    del broken['output_layer.bias']
    with pytest.raises(KeyError):
        load.state_dict(broken)
    # same seed, same bits (fixtures regenerate weights from the seed)
    again = weights.seeded_state_dict(seed=3)
    assert all(torch.equal(again[k], state[k]) for k in state)


def test_positional_encoding_formula():
    pe = weights.positional_encoding(256)
    assert pe.shape == (5000, 1, 256)
    p, i = 37, 10
    freq = math.exp(-2 * i * math.log(10000.0) / 256)
    assert abs(pe[p, 0, 2 * i].item() - math.sin(p * freq)) < 1e-5
    assert abs(pe[p, 0, 2 * i + 1].item() - math.cos(p * freq)) < 1e-5


def test_audio_ingest(tmp_path):
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    x = (0.1 * rng.standard_normal(16000)).astype(np.float32)
    wavfile.write(tmp_path / 'f32.wav', 16000, x)
    wavfile.write(tmp_path / 'i16.wav', 16000, (x * 32768).astype(np.int16))
    assert load.info(tmp_path / 'f32.wav') == (16000, 16000)
    a = load.audio(tmp_path / 'f32.wav')
    assert a.shape == (1, 16000) and torch.equal(a[0], torch.from_numpy(x))
    b = load.audio(tmp_path / 'i16.wav')
    assert (a - b).abs().max() < 1 / 32768
    # resampling: identity at 16 kHz; anything else is the HIP kernel, which
    # must fail loudly (no CPU path) when no device is visible
    assert ppgs_amd.resample(a, 16000) is a
    if not torch.cuda.is_available():
        with pytest.raises(ppgs_amd.engine.PpgError):
            ppgs_amd.resample(a, 8000)
    assert data.frames_of(16000, 16000) == 100 and data.frames_of(8000, 8000) == 100


def test_cli_flags_mirror_reference():
    out = subprocess.run(
        [sys.executable, '-m', 'ppgs_amd', '--help'], capture_output=True,
        text=True, check=True).stdout
    for flag in ('--audio_files', '--output_files', '--representation',
                 '--checkpoint', '--num-workers', '--gpu', '--max-frames',
                 '--legacy-mode'):
        assert flag in out


def test_api_surface_matches_reference_names():
    for name in ('from_audio', 'from_features', 'from_file', 'from_file_to_file',
                 'from_files_to_files', 'from_dataloader', 'infer', 'resample'):
        assert callable(getattr(ppgs_amd, name))
    assert len(ppgs_amd.PHONEMES) == 40 and ppgs_amd.PHONEMES[-1] == '<silent>'
    assert ppgs_amd.PHONEMES[:3] == ['aa', 'ae', 'ah']
    import inspect
    sig = inspect.signature(ppgs_amd.from_files_to_files)
    assert list(sig.parameters) == [
        'audio_files', 'output_files', 'representation', 'checkpoint',
        'num_workers', 'gpu', 'max_frames', 'legacy_mode']
    sig = inspect.signature(ppgs_amd.from_features)
    assert list(sig.parameters) == [
        'features', 'lengths', 'representation', 'checkpoint', 'gpu',
        'softmax', 'legacy_mode']


def test_lpt_sharding_balances_cost():
    rng = np.random.default_rng(1)
    frames = rng.integers(50, 3001, size=500)
    costs = [data.flops(int(f)) for f in frames]
    shards = distributed.shard_lpt(costs, 8)
    assert sorted(i for s in shards for i in s) == list(range(500))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) / (sum(loads) / 8) < 1.02
    assert distributed.shard_lpt(costs, 8) == shards          # deterministic


def test_product_never_imports_reference_or_oracle():
    """ppgs_amd/ is the product: no module of it may import the reference package
    (`ppgs`) or the test-only oracle, in any branch."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.abspath(ppgs_amd.__file__))
    files = glob.glob(os.path.join(root, '**', '*.py'), recursive=True)
    assert len(files) > 10
    for path in files:
        text = open(path).read()
        hit = re.search(r'^\s*(?:import|from)\s+(ppgs|oracle)\b', text, re.M)
        assert hit is None, (path, hit.group(0))
    with pytest.raises(ValueError):
        os.environ.pop('PPGS_AMD_SIMILARITY_MATRIX', None)
        ppgs_amd.core.similarity_matrix()


def test_bench_entry_point_fails_loudly_without_the_gpus():
    """bench.py --gpus N must either run N ranks or fail: never report fewer."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, 'bench.py')
    if torch.cuda.is_available():
        pytest.skip('CPU-side check')
    for argv in (['--gpus', '2'], ['--gpus', '1']):
        run = subprocess.run([sys.executable, bench] + argv, capture_output=True, text=True)
        assert run.returncode != 0 and 'bench.py' in (run.stderr + run.stdout)
        assert '"metric"' not in run.stdout
    # launched as rank 0 of 2 by a launcher while --gpus says 1: refuse
    env = dict(os.environ, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0')
    run = subprocess.run([sys.executable, bench, '--gpus', '1'], capture_output=True, text=True, env=env)
    assert run.returncode != 0


def test_plan_rows_and_row_budgeted_packing():
    """data.plan_rows restates the planner's row count per item (ppg_plan_windows through the C ABI);
    pack_batches(max_rows=...) keeps every batch's plan within the budget and the frame invariant."""
    import random
    from ppgs_amd import engine
    rng = random.Random(3)
    for _ in range(200):
        frames = rng.choice([16, 100, 499, 500, 501, 850, 1000, 1234, 2999])
        lengths = [frames] + [rng.randint(0, frames) for _ in range(rng.randint(0, 5))]
        _, info = engine.plan_windows(len(lengths), frames, lengths)
        assert info.tokens == sum(data.plan_rows(length, frames) for length in lengths), lengths
    lengths = [rng.randint(50, 3000) for _ in range(2000)]
    free = data.pack_batches(lengths, 32000)
    bound = data.pack_batches(lengths, 32000, max_rows=40960)
    assert sorted(i for b in bound for i in b) == list(range(len(lengths)))
    over = 0
    for batches, limit in ((free, None), (bound, 40960)):
        for batch in batches:
            longest = max(lengths[i] for i in batch)
            assert len(batch) == 1 or len(batch) * longest <= 32000
            _, info = engine.plan_windows(len(batch), longest, [lengths[i] for i in batch])
            if limit:
                assert info.tokens <= limit
            else:
                over += info.tokens > 40960
    assert over > 0                      # the budget is what keeps them out, not the corpus
    assert len(bound) <= len(free) + len(free) // 20 + 1


def test_no_kernel_is_exposed_to_the_packed_fp32_mfma_hazard():
    """on gfx950 a v_pk_add/mul/fma_f32 with op_sel set on SOURCE 1 computes with the wrong operand half,
    intermittently, while another wave of its SIMD issues 16x16x32 MFMAs (tools/probes/pk_mfma_probe.hip; it is what
    made round 2's mel frontend wrong beside attention kernels).  hipcc writes that form by itself, so the BUILT
    library's ISA is audited: NO kernel may contain it, whatever its register allocation (a > 256-register kernel keeps
    its own waves apart, but another kernel's small MFMA wave fits in the 512 - vgprs registers it leaves)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no llvm-objdump')
    if not os.path.exists(os.path.join(root, 'ppgs_amd', 'libppgs_amd.so')):
        pytest.skip('library not built')
    run = subprocess.run([sys.executable, os.path.join(root, 'tools', 'pk_scan.py'), '--strict'],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-4000:] + run.stderr[-2000:]
    assert ' 0 exposed' in run.stdout
    # the audit sees the kernels that matter and knows the vulnerable form when it meets it
    assert 'frontend_kernel' in run.stdout and 'attn_mixed_kernel' in run.stdout and 'layer32_kernel' in run.stdout
    assert 'one wave per SIMD' not in run.stdout          # (round 4's exemption class is gone)
    sys.path.insert(0, os.path.join(root, 'tools'))
    import pk_scan
    assert pk_scan.vulnerable('v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]')
    assert pk_scan.vulnerable('v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,1,0] op_sel_hi:[1,1,1]')
    assert not pk_scan.vulnerable('v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]')
    assert not pk_scan.vulnerable('v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,1] op_sel_hi:[0,0,0]')
    assert not pk_scan.vulnerable('v[0:1], v[2:3], v[4:5]')


def test_bind_cpus_against_a_faked_two_socket_eight_gpu_tree(tmp_path):
    """The first real 8-GPU run must not be the first execution of the binding logic: a faked sysfs tree of the usual
    MI355X box -- 2 NUMA nodes x 96 cores (+ SMT siblings 192..383), 4 GPUs per node -- gives every rank a disjoint
    slice of its own GPU's node; a GPU without a numa_node file falls back to the equal split."""
    sysfs = tmp_path / 'sys'
    bdfs = []
    for gpu in range(8):
        bdf = f'0000:{0x05 + 0x10 * gpu:02x}:00.0'
        bdfs.append(bdf)
        device = sysfs / 'bus' / 'pci' / 'devices' / bdf
        device.mkdir(parents=True)
        (device / 'numa_node').write_text(f'{gpu // 4}\n')
    for node in range(2):
        directory = sysfs / 'devices' / 'system' / 'node' / f'node{node}'
        directory.mkdir(parents=True)
        (directory / 'cpulist').write_text(f'{96 * node}-{96 * node + 95},{192 + 96 * node}-{192 + 96 * node + 95}\n')
    slices = [distributed.bind_cpus(rank, 8, sysfs=str(sysfs), bdfs=bdfs, apply=False) for rank in range(8)]
    assert all(len(cpus) == 48 for cpus in slices)
    for rank, cpus in enumerate(slices):
        node = rank // 4
        assert all((96 * node <= c < 96 * node + 96) or (192 + 96 * node <= c < 192 + 96 * node + 96) for c in cpus), rank
    flat = [c for cpus in slices for c in cpus]
    assert len(flat) == len(set(flat)) == 384
    # 4 ranks on a 8-GPU node: the ranks still follow THEIR GPUs' nodes
    slices = [distributed.bind_cpus(rank, 4, sysfs=str(sysfs), bdfs=bdfs[2:6], apply=False) for rank in range(4)]
    assert [min(c) // 96 % 2 for c in slices] == [0, 0, 1, 1] and all(len(c) == 96 for c in slices)
    # the pure rule: unknown node -> equal split of what is allowed; too few cores -> hands off
    assert distributed.cpu_slice(1, 2, range(16), [-1, -1], {}) == list(range(8, 16))
    assert distributed.cpu_slice(0, 8, range(8), [0] * 8, {0: set(range(8))}) == []
    assert distributed.cpu_slice(3, 4, range(64), [0, 0, 1, 1], {0: set(range(32)), 1: set(range(32, 64))}) == list(range(48, 64))


def test_w2v2fb_engine_cache_is_bound_to_the_model_object(monkeypatch):
    """The HIP engines of the w2v2fb representation are cached per (device, model, precision).  The key holds id(model):
    the entry must keep the model alive and be checked against the object -- a freed model's id can be handed to the
    next model, and an engine found under it would hold the old weights (seen as an intermittent GPU test failure)."""
    import types
    import torch
    from ppgs_amd import engine
    from ppgs_amd.preprocess import w2v2fb
    built = []

    class Fake:
        def __init__(self, *args):
            built.append(args)
    monkeypatch.setattr(engine, 'W2v2FeatureEncoder', Fake)
    monkeypatch.setattr(engine, 'W2v2Body', Fake)
    w2v2fb.clear()
    device = torch.device('cuda', 0)
    make = lambda: types.SimpleNamespace(feature_extractor=types.SimpleNamespace(state_dict=lambda: {}))
    a, b = make(), make()
    ea, ba = w2v2fb.feature_encoder_for(device, a), w2v2fb.body_for(device, a)
    assert w2v2fb.feature_encoder_for(device, a) is ea and w2v2fb.body_for(device, a) is ba and len(built) == 2
    # an entry under b's key that belongs to another model (what id reuse produces) is not handed out
    from ppgs_amd import core
    w2v2fb._encoders[(str(device), id(b), w2v2fb.w2v2_precision())] = w2v2fb._encoders[(str(device), id(a), w2v2fb.w2v2_precision())]
    w2v2fb._bodies[(str(device), id(b), w2v2fb.w2v2_precision())] = w2v2fb._bodies[(str(device), id(a), w2v2fb.w2v2_precision())]
    assert w2v2fb.feature_encoder_for(device, b) is not ea and w2v2fb.body_for(device, b) is not ba and len(built) == 4
    # and the cache holds the models
    assert any(entry[0] is a for entry in w2v2fb._encoders.values())
    w2v2fb.clear()
    assert not w2v2fb._encoders and not w2v2fb._bodies and not w2v2fb._models


def test_w2v2fb_engines_follow_the_package_precision(monkeypatch):
    """The wav2vec2 engines have fp32 and 16-bit forms: PRECISION = 'fp16x2' (<= 1e-4 on fp16 hi + lo operands, PPG
    network only) builds them in fp32 -- their own <= 1e-4 form -- and every other precision is handed through; the
    engine caches are keyed on what was built."""
    import types
    import torch
    from ppgs_amd import core, engine
    from ppgs_amd.preprocess import w2v2fb
    built = []

    class Fake:
        def __init__(self, *args):
            built.append(args[-1])
    monkeypatch.setattr(engine, 'W2v2FeatureEncoder', Fake)
    monkeypatch.setattr(engine, 'W2v2Body', Fake)
    w2v2fb.clear()
    device = torch.device('cuda', 0)
    model = types.SimpleNamespace(feature_extractor=types.SimpleNamespace(state_dict=lambda: {}))
    for precision, expected in (('fp16x2', 'fp16x2'), ('fp32', 'fp32'), ('fp16', 'fp16'), ('bf16', 'bf16')):
        monkeypatch.setattr(core, 'PRECISION', precision)
        assert w2v2fb.w2v2_precision() == expected
        w2v2fb.feature_encoder_for(device, model)
        w2v2fb.body_for(device, model)
    assert built == ['fp16x2', 'fp16x2', 'fp32', 'fp32', 'fp16', 'fp16', 'bf16', 'bf16']
    # PPGS_AMD_W2V2_FP32=1: the wav2vec2 engines of the fp16x2 mode in fp32 (the route until round 5) -- the fp32 mode's pair
    monkeypatch.setenv('PPGS_AMD_W2V2_FP32', '1')
    monkeypatch.setattr(core, 'PRECISION', 'fp16x2')
    assert w2v2fb.w2v2_precision() == 'fp32'
    w2v2fb.feature_encoder_for(device, model)
    assert len(built) == 8
    w2v2fb.clear()


def test_hot_kernels_do_not_wait_for_store_acknowledgements():
    """hipcc's wait-count insertion across control flow puts `s_waitcnt vmcnt(0)` behind stores -- the
    wave then waits for the store's ACKNOWLEDGEMENT (300 cycles per store instruction in ppg_gemm32.hip's epilogue
    before it was restructured).  tools/wait_scan.py counts, per kernel of the built library, the vmcnt(0) waits that
    follow a store; the kernels of the benchmarked paths have a budget (what is left: phase ends, the irregular-tile
    path of the Q/K/V tail).  A compiler upgrade or an edit that re-introduces the pattern fails here."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    library = os.path.join(root, 'ppgs_amd', 'libppgs_amd.so')
    if not os.path.exists(library) or not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no built library / llvm-objdump')
    spec = importlib.util.spec_from_file_location('wait_scan', os.path.join(root, 'tools', 'wait_scan.py'))
    wait_scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wait_scan)
    kernels = wait_scan.scan(library)
    budget = {
        'layer32_kernel<PrecBF16, 256, true, 5>': 4, 'layer32_kernel<PrecBF16, 256, false, 5>': 0,
        'head32_kernel<PrecBF16, 5>': 13, 'attn_mixed_kernel<PrecBF16>': 3, 'outconv_kernel<PrecBF16>': 0,     # (attention: the tile loops' own end-of-iteration waits: two-stage, three-stage and its tile-0 stage)
        'gemm32_kernel<PrecBF16, 4, 0>': 0, 'gemm32_kernel<PrecBF16, 4, 1>': 0, 'gemm32_kernel<PrecBF16, 5, 3>': 0,
        'posconv_kernel<PrecBF16>': 0, 'w2v2_layernorm_kernel<PrecBF16, 768>': 0,
        'linear_kernel<PrecBF16, 1, 16, 1>': 1, 'ffn32x2_kernel<true, true>': 3,
    }
    for name, allowed in budget.items():
        assert name in kernels, name
        assert kernels[name]['vmcnt0_after_store'] <= allowed, (name, dict(kernels[name]))


def test_hand_placed_loads_stay_untouched_until_their_wait():
    """The head kernel requests its convolution weights with inline-asm loads at its top and awaits them by hand behind
    the gather -- code the compiler places, blind to the loads.  tools/asm_load_scan.py follows every hand-placed
    fragment load of the built head kernels in issue order and fails on an instruction that names a register still in
    flight (hipcc did exactly that when the tail's weights were requested ahead of the epilogue: the registers were
    copied into the accumulation file before the loads had landed)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    library = os.path.join(root, 'ppgs_amd', 'libppgs_amd.so')
    if not os.path.exists(library) or not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no built library / llvm-objdump')
    spec = importlib.util.spec_from_file_location('asm_load_scan', os.path.join(root, 'tools', 'asm_load_scan.py'))
    asm_load_scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(asm_load_scan)
    seen, tracked, bad = asm_load_scan.scan(library, 'head32_kernel')
    assert seen == 4 and tracked >= 4 * 100, (seen, tracked)
    assert not bad, bad[:5]
    # the attention kernels request their Q rows the same way (awaited with a counted vmcnt beside the tile DMAs); the
    # layer kernels (W2 into accumulation registers, hand-counted waits in the chunk loop and the tail), the fp16x2 layer
    # kernel and the wav2vec2 body's GEMM place theirs by hand as well
    for family, kernels, loads in (('attn_mixed_kernel', 4, 4), ('attn_kernel', 8, 4), ('layer32_kernel', 12, 100),
                                   ('ffn32x2_kernel', 3, 100), ('gemm32_kernel', 16, 100)):
        seen, tracked, bad = asm_load_scan.scan(library, family)
        assert seen == kernels and tracked >= loads * kernels, (family, seen, tracked)
        assert not bad, (family, bad[:5])


def test_asm_mfma_results_are_read_far_enough_behind_their_mfma():
    """the layer kernels' phase A issues its MFMAs from inline asm with the accumulator in
    architectural registers, and hipcc's hazard recognizer does not see an MFMA inside asm -- the wait states between the
    MFMA's write and the pack / ReLU that read it, and between an MFMA still reading its C operand (b1) and a VALU write
    of those registers, exist only as source placement.  tools/mfma_hazard_scan.py walks the feature-split kernels of
    the BUILT library and requires PASSES + 4 wait states in front of the first consumer (gfx940 guide: 8 passes 11) and
    PASSES - 1 in front of a VALU overwrite of a live C operand: a compiler upgrade or a flag that lets the scheduler
    hoist a consumer fails here, not as a silently stale h in the default bf16 / fp16 layer kernel."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    library = os.path.join(root, 'ppgs_amd', 'libppgs_amd.so')
    if not os.path.exists(library) or not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no built library / llvm-objdump')
    sys.path.insert(0, os.path.join(root, 'tools'))
    spec = importlib.util.spec_from_file_location('mfma_hazard_scan', os.path.join(root, 'tools', 'mfma_hazard_scan.py'))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    seen, audited, bad = scan.scan(library)
    assert seen >= 30 and audited >= 500, (seen, audited)        # (the asm MFMAs of the layer kernels are there to be audited)
    assert not bad, bad[:5]
    # the walk has teeth: with a window of 24 more wait states it names the closest consumers (the fp16x2 kernel's)
    scan.MARGIN = 28
    try:
        assert scan.scan(library)[2]
    finally:
        scan.MARGIN = 4


def test_bench_refuses_experiment_switches_and_reads_the_clock_probe(monkeypatch, tmp_path):
    """bench.py's line is self-defending: with a PPGS_AMD_* experiment switch in the environment it
    exits before touching the GPU (--allow-ablation: the switches are carried in the line, its `value` nulled);
    configuration switches (PPGS_AMD_STREAMS ...) are recorded, not refused.  And the two text parsers beside the line:
    the driver's sysfs clock file and the stand-alone dense-MFMA probe's output."""
    import argparse
    import importlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    for key in list(os.environ):
        if key.startswith('PPGS_AMD_'):
            monkeypatch.delenv(key)
    args = argparse.Namespace(allow_ablation=False)
    monkeypatch.setenv('PPGS_AMD_STREAMS', '1')
    assert bench.env_guard(args) == ({'PPGS_AMD_STREAMS': '1'}, {})
    monkeypatch.setenv('PPGS_AMD_LAYER32', '0')
    with pytest.raises(SystemExit):
        bench.env_guard(args)
    args.allow_ablation = True
    assert bench.env_guard(args)[1] == {'PPGS_AMD_LAYER32': '0'}
    # sysfs: the line with the star is the current level
    clock = tmp_path / 'pp_dpm_sclk'
    clock.write_text('0: 500Mhz \n1: 1873Mhz *\n2: 2400Mhz \n')
    sampler = bench.ClockSampler.__new__(bench.ClockSampler)
    sampler.path = str(clock)
    assert sampler._read() == 1873.0
    # the probe record committed with round 5
    with open(os.path.join(root, 'profiles', 'r5_mfma_clock_probe.txt')) as f:
        probe = bench.parse_mfma_probe(f.read())
    assert 1.5 < probe['dense']['clock_ghz_p50'] < 2.4 and 31.5 < probe['dense']['cycles_per_mfma'] < 33.0
    assert 0.6 < probe['dense']['tflops'] / 2500.0 < 0.9
    assert probe['with_6_valu_per_mfma']['tflops'] < probe['with_3_valu_per_mfma']['tflops'] < probe['dense']['tflops']


def test_bench_roofline_flops_follow_from_the_launches_that_compute_them():
    """`roofline.flops_per_launch` is the MEAN over a step's five layer launches of what each
    computes -- the last layer's launch has no Q/K/V tail, and with the head kernel making layer 0's Q/K/V only four of
    the five launches carry one.  flops_per_launch x launches per step can never exceed the step's algorithmic FLOPs."""
    import importlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    from ppgs_amd import data
    H, F, L = bench.HIDDEN, bench.FFN, bench.LAYERS
    steps = 5
    processed = bench.BATCH * sum(data.chunk_lengths(bench.FRAMES))
    step_flops = bench.BATCH * data.flops(bench.FRAMES)
    for pipelines in (1, 2):
        # the product's launch sequence: head kernel (class inconv), no gather / Q/K/V / out-projection launches
        kernels = {'gather': (0.0, 0), 'inconv': (1.0, steps * pipelines), 'qkv': (0.0, 0), 'outproj_ln': (0.0, 0),
                   'ffn': (1.0, L * steps * pipelines)}
        per_frame, op_fused, fused = bench.layer_flops_per_frame(kernels, steps, pipelines)
        assert op_fused and fused == L - 1
        assert per_frame == 4 * H * F + 2 * H * H + 6 * H * H * (L - 1) / L
        assert per_frame * processed * L <= step_flops
        # the unfused head (gather + input convolution + a stand-alone Q/K/V launch for layer 0): four tails as well
        kernels.update(gather=(1.0, steps * pipelines), qkv=(1.0, steps * pipelines))
        assert bench.layer_flops_per_frame(kernels, steps, pipelines)[2] == L - 1
        # a stand-alone Q/K/V launch per layer: no tails
        kernels['qkv'] = (1.0, L * steps * pipelines)
        assert bench.layer_flops_per_frame(kernels, steps, pipelines)[0] == 4 * H * F + 2 * H * H
    # every layer launch of a step, summed, stays inside the step
    assert (4 * H * F + 2 * H * H) * processed * L + 6 * H * H * processed * (L - 1) < step_flops
