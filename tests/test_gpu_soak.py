"""Containment soak: the kernels of this package beside OTHER kernels on the chip.

Round 2's 256-thread mel frontend computed a frame pair wrong about once per 300 pairs when an attention kernel of
another HIP stream shared its CU (PyTorch's own scaled_dot_product_attention is enough).  Since round 3 the product
runs kernels of its own two pipelines side by side by default, so every victim below is driven for thousands of
launches with `torch.nn.functional.scaled_dot_product_attention` hammering a third stream, and every result must be
bit-identical to the result of the same call on a quiet chip in the one-pipeline configuration:

  * ppg_encode as two pipelines (bf16 and fp16), >= 2000 C2 steps each; the fp16x2 mode at both geometries, 600 steps each,
  * the mel frontend (512-thread workgroups that own their CU),
  * the wav2vec2 body as two pipelines,
  * the batched KV-cached stream step (row-mapped launches).

The comparisons run on the device (one flag per step, summed) so that the host never drains the queues: victim and
aggressor stay co-resident for the whole run.
"""
import pytest
import torch

import ppgs_amd
from ppgs_amd import engine as E
from ppgs_amd import weights as W

pytestmark = pytest.mark.gpu

STEPS = 2000


class Aggressor:
    """scaled_dot_product_attention on a stream of its own, topped up as the victim loop goes."""

    def __init__(self, per_step=3):
        self.stream = torch.cuda.Stream()
        self.q = torch.randn(32, 2, 1000, 128, device='cuda', dtype=torch.bfloat16)
        self.per_step = per_step
        self.launched = 0

    def top_up(self):
        with torch.cuda.stream(self.stream):
            for _ in range(self.per_step):
                torch.nn.functional.scaled_dot_product_attention(self.q, self.q, self.q)
        self.launched += self.per_step


def soak(step, reference, steps, aggressor, flush_every=250):
    """`steps` calls of step() on the current stream beside the aggressor; number of results != reference."""
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    for index in range(steps):
        aggressor.top_up()
        out = step()
        bad += (out != reference).any()
        if index % flush_every == flush_every - 1:
            torch.cuda.synchronize()           # (bounds the queue depth; both streams are refilled right after)
    torch.cuda.synchronize()
    return int(bad)


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_soak_encode_two_pipelines_beside_sdpa(monkeypatch, precision):
    state = W.seeded_state_dict(seed=1234)
    gen = torch.Generator().manual_seed(77)
    feats = torch.randn(32, 80, 1000, generator=gen).half().cuda()
    lengths = [1000] * 32
    monkeypatch.setenv('PPGS_AMD_STREAMS', '1')
    reference = E.Engine(state, 0, precision).encode(feats, lengths).clone()
    monkeypatch.delenv('PPGS_AMD_STREAMS')
    engine = E.Engine(state, 0, precision)
    _, info = E.plan_windows(32, 1000, lengths, engine=engine)
    assert engine.pipelines(info.tokens) == 2
    torch.cuda.synchronize()
    aggressor = Aggressor()
    victim = torch.cuda.Stream()
    with torch.cuda.stream(victim):
        bad = soak(lambda: engine.encode(feats, lengths), reference, STEPS, aggressor)
    assert bad == 0, f'{bad} of {STEPS} two-pipeline {precision} steps beside SDPA differ from the quiet one-pipeline result'
    assert aggressor.launched >= STEPS


@pytest.mark.parametrize('geometry', ['hidden256', 'hidden512'])
def test_soak_fp16x2_encode_beside_sdpa(geometry):
    """The fp16x2 mode's kernels (hi + lo operand planes: ffn32x2 layer kernel, split-precision attention and linear
    kernels; at hidden 512 the 32-KiB-tile attention and the two-GEMM FFN) beside SDPA on another stream: every step
    bit-equal to the quiet run.  Its steps are 3 - 5 x the 16-bit ones, so the soak is 600 steps (>= 15 000 launches)."""
    steps = 600
    if geometry == 'hidden256':
        state = W.seeded_state_dict(seed=1234)
        feats = torch.randn(32, 80, 1000, generator=torch.Generator().manual_seed(78)).half().cuda()
        lengths = [1000] * 32
    else:
        state = W.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
        feats = torch.randn(16, 768, 1000, generator=torch.Generator().manual_seed(79)).half().cuda()
        lengths = [1000] * 16
    engine = E.Engine(state, 0, 'fp16x2')
    reference = engine.encode(feats, lengths).clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(reference).all())
    aggressor = Aggressor()
    victim = torch.cuda.Stream()
    with torch.cuda.stream(victim):
        bad = soak(lambda: engine.encode(feats, lengths), reference, steps, aggressor)
    assert bad == 0, f'{bad} of {steps} fp16x2 {geometry} steps beside SDPA differ from the quiet result'
    assert aggressor.launched >= steps


def test_soak_frontend_beside_sdpa():
    gen = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
    reference = ppgs_amd.preprocess.mel.from_audios(audio).clone()
    torch.cuda.synchronize()
    aggressor = Aggressor(per_step=1)
    victim = torch.cuda.Stream()
    with torch.cuda.stream(victim):
        bad = soak(lambda: ppgs_amd.preprocess.mel.from_audios(audio), reference, STEPS, aggressor)
    assert bad == 0, f'{bad} of {STEPS} frontend launches beside SDPA differ'


def test_frontend_beside_a_matrix_core_aggressor_on_its_own_simds():
    """The mechanism itself: a synthetic kernel that spins on v_mfma_f32_16x16x32_bf16 with 16 registers
    and NO LDS fits beside the frontend's workgroups on their SIMDs even though those own the CU's LDS -- the
    situation in which round 2's arithmetic (v_pk_*_f32 with op_sel on source 1) got 80 of 240 launches wrong."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'bin', 'libaggressors.so')
    if not os.path.exists(path):
        pytest.skip('tools/bin/libaggressors.so not built (tools/probes/build.sh; __graft_entry__.build() does it)')
    lib = ctypes.CDLL(path)
    lib.aggressor_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
    gen = torch.Generator().manual_seed(1234)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
    reference = ppgs_amd.preprocess.mel.from_audios(audio).clone()
    src, sink = torch.randn(1 << 20, device='cuda'), torch.zeros(256, device='cuda')
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    for rep in range(60):
        with torch.cuda.stream(a):
            for _ in range(2):      # mask 2 = v_mfma_f32_16x16x32_bf16 only; ~400 us per launch
                assert lib.aggressor_launch(2, 512, 1500, 0, src.data_ptr(), sink.data_ptr(), a.cuda_stream) == 0
        with torch.cuda.stream(b):
            for _ in range(6):
                bad += (ppgs_amd.preprocess.mel.from_audios(audio) != reference).any()
        torch.cuda.synchronize()
    assert int(bad) == 0, f'{int(bad)} of 360 frontend launches beside the MFMA aggressor differ'


def test_soak_whole_steps_on_two_caller_streams_beside_sdpa():
    """The file pipeline's shape: whole steps (frontend + two-pipeline encode) alternating on two caller streams, so that
    one step's frontend runs beside the other's encoder, with the external aggressor on top."""
    state = W.seeded_state_dict(seed=1234)
    engine = E.Engine(state, 0, 'bf16')
    gen = torch.Generator().manual_seed(5)
    audio = (0.1 * torch.randn(32, 1, 160000, generator=gen)).cuda()
    lengths = [1000] * 32
    reference = engine.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths).clone()
    torch.cuda.synchronize()
    aggressor = Aggressor(per_step=2)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    # one counter per caller stream, updated on that stream (a flag handed to another stream would race with the
    # caching allocator reusing its block)
    bad = [torch.zeros((), dtype=torch.int64, device='cuda') for _ in streams]
    total = 600
    for index in range(total):
        aggressor.top_up()
        with torch.cuda.stream(streams[index % 2]):
            out = engine.encode(ppgs_amd.preprocess.mel.from_audios(audio), lengths)
            bad[index % 2] += (out != reference).any()
        if index % 100 == 99:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    wrong = int(bad[0]) + int(bad[1])
    assert wrong == 0, f'{wrong} of {total} overlapped steps differ'


def test_soak_w2v2_body_two_pipelines_beside_sdpa(monkeypatch):
    import transformers
    transformers.utils.logging.set_verbosity_error()
    torch.manual_seed(5)
    hf = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config(num_hidden_layers=4)).eval()
    x = torch.randn(16, 499, 512, generator=torch.Generator().manual_seed(4)).cuda()
    valid = [499] * 16
    monkeypatch.setenv('PPGS_AMD_W2V2_STREAMS', '1')
    reference = E.W2v2Body(hf, 0, 'bf16')(x, valid).clone()
    monkeypatch.delenv('PPGS_AMD_W2V2_STREAMS')
    body = E.W2v2Body(hf, 0, 'bf16')
    torch.cuda.synchronize()
    aggressor = Aggressor(per_step=4)
    victim = torch.cuda.Stream()
    steps = 500
    with torch.cuda.stream(victim):
        bad = soak(lambda: body(x, valid), reference, steps, aggressor, flush_every=100)
    assert bad == 0, f'{bad} of {steps} two-pipeline body forwards beside SDPA differ'


def test_soak_batched_stream_steps_beside_sdpa():
    """64 KV-cached causal streams advanced 16 frames per step: the whole 30-step sequence repeated beside the
    aggressor equals the quiet run, output by output."""
    state = W.seeded_state_dict(seed=1234)
    engine = E.Engine(state, 0, 'bf16', True)
    gen = torch.Generator().manual_seed(12)
    batch, frames, hop = 64, 480, 16
    feats = torch.randn(batch, 80, frames, generator=gen).half().cuda()

    def sequence(aggressor=None):
        stream = engine.batched_stream(batch, frames)
        outs = []
        for step in range(frames // hop):
            if aggressor is not None:
                aggressor.top_up()
            last = step == frames // hop - 1
            out = stream.push(feats[:, :, step * hop:(step + 1) * hop], flush=last)
            outs.append(torch.cat([piece for piece in out], dim=1))
        return outs

    reference = sequence()
    torch.cuda.synchronize()
    aggressor = Aggressor(per_step=1)
    victim = torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    reps = 70                                        # 70 x 30 = 2100 steps
    with torch.cuda.stream(victim):
        for rep in range(reps):
            for out, ref in zip(sequence(aggressor), reference):
                bad += (out != ref).any()
            torch.cuda.synchronize()
    assert int(bad) == 0, f'{int(bad)} of {reps * (frames // hop)} stream steps beside SDPA differ'


class MatrixAggressor:
    """tools/probes/aggressors.hip mask 2: a kernel that spins on v_mfma_f32_16x16x32_bf16 with 16 registers and NO LDS --
    it fits beside a 380- or a 478-register workgroup on its SIMDs (512 - vgprs registers are left) whatever LDS that
    workgroup holds; launched on a stream of its own and topped up as the victim loop goes."""

    def __init__(self, iters=1500):
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'bin', 'libaggressors.so')
        if not os.path.exists(path):
            pytest.skip('tools/bin/libaggressors.so not built (tools/probes/build.sh; __graft_entry__.build() does it)')
        self.lib = ctypes.CDLL(path)
        self.lib.aggressor_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
        self.src, self.sink = torch.randn(1 << 20, device='cuda'), torch.zeros(256, device='cuda')
        self.stream = torch.cuda.Stream()
        self.iters = iters
        self.launched = 0

    def top_up(self):
        assert self.lib.aggressor_launch(2, 512, self.iters, 0, self.src.data_ptr(), self.sink.data_ptr(),
                                         self.stream.cuda_stream) == 0
        self.launched += 1


def test_soak_ffn32x2_beside_the_matrix_core_aggressor():
    """`ffn32x2_kernel` (the <= 1e-4 mode's layer kernel) allocates 380 / 420 registers and carried 48
    v_pk_add_f32 with op_sel on source 1 -- 92 - 132 registers per SIMD lane were free for another kernel's MFMA wave,
    and the audit exempted it as "one wave per SIMD".  The form is gone from the library (tools/pk_scan.py: zero in ANY
    kernel); this is the soak: fp16x2, 32 x 1000, beside the 16-register LDS-free 16x16x32 aggressor, bit-equal."""
    steps = 400
    state = W.seeded_state_dict(seed=1234)
    feats = torch.randn(32, 80, 1000, generator=torch.Generator().manual_seed(81)).half().cuda()
    lengths = [1000] * 32
    engine = E.Engine(state, 0, 'fp16x2')
    reference = engine.encode(feats, lengths).clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(reference).all())
    aggressor = MatrixAggressor(iters=4000)          # (~1 ms per launch: covers an fp16x2 step of ~1.9 ms with two)
    victim = torch.cuda.Stream()

    class Twice:
        def top_up(self):
            aggressor.top_up()
            aggressor.top_up()
    with torch.cuda.stream(victim):
        bad = soak(lambda: engine.encode(feats, lengths), reference, steps, Twice(), flush_every=50)
    assert bad == 0, f'{bad} of {steps} fp16x2 steps beside the MFMA aggressor differ from the quiet result'


@pytest.mark.parametrize('precision', ['bf16', 'fp16'])
def test_soak_subtile_layer_kernel_beside_the_matrix_core_aggressor(precision):
    """The sub-tile `layer32_kernel<..., 2>` (configs[4]: 64 x 160 causal frames, and every small batch): 478 registers,
    a partial-CU LDS footprint, 32 vulnerable instructions until round 5.  2000 steps beside the aggressor, bit-equal."""
    steps = 2000
    state = W.seeded_state_dict(seed=1234)
    feats = torch.randn(64, 80, 160, generator=torch.Generator().manual_seed(82)).half().cuda()
    lengths = [160] * 64
    engine = E.Engine(state, 0, precision, True)
    reference = engine.encode(feats, lengths).clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(reference).all())
    aggressor = MatrixAggressor(iters=1200)          # (~0.3 ms per launch against a ~0.28 ms step)
    victim = torch.cuda.Stream()
    with torch.cuda.stream(victim):
        bad = soak(lambda: engine.encode(feats, lengths), reference, steps, aggressor)
    assert bad == 0, f'{bad} of {steps} sub-tile causal {precision} steps beside the MFMA aggressor differ'
