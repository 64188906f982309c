"""configs[2]: w2v2fb representation, batch = 16 x 1000 frames (16 x 160000 samples), one MI355X.

    python tools/bench_c3.py [precision]          # prints one JSON record

Three legs, each timed on its own with the inputs resident in HBM:
  * `ppg_network`   w2v2fb-shaped features (16, 768, 1000) fp16 -> hidden-512 PPG network (the engine);
  * `feature_encoder_hip` / `feature_encoder_pytorch`   the wav2vec2 convolutional feature encoder on
    16 x 160080 samples: ppg_w2v2_features against HF's module on PyTorch-ROCm (same seeded weights);
  * `w2v2_transformer_pytorch`   feature projection + 12-layer transformer of the HF model (PyTorch-ROCm).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppgs_amd                                           # noqa: E402
from ppgs_amd import engine as E                          # noqa: E402

PEAK = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3, 'fp16x2': 2500.0}   # (fp16x2: algorithmic FLOPs against the fp16 peak; it issues three MFMAs per product)


def timed(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - start) / steps


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    # the wav2vec2 engines run in the same mode (fp16x2: hi + lo operand pairs since round 5; PPGS_AMD_W2V2_FP32=1: in fp32,
    # the route until then)
    w2v2_precision = 'fp32' if precision == 'fp16x2' and os.environ.get('PPGS_AMD_W2V2_FP32', '0') == '1' else precision
    state = ppgs_amd.weights.seeded_state_dict(seed=55, input_channels=768, hidden_channels=512)
    model = E.Engine(state, 0, precision)
    feats = torch.randn(16, 768, 1000).half().cuda()
    lengths = [1000] * 16
    ms = timed(lambda: model.encode(feats, lengths))
    model.profile(True)
    for _ in range(5):
        model.encode(feats, lengths)
    torch.cuda.synchronize()
    kernels = {n: v[0] / 5 for n, v in model.profile_read().items()}
    launches = {n: v[1] / 5 for n, v in model.profile_read().items()}
    model.profile(False)
    # 35 594 240 FLOP per processed frame + 10 240 Tc^2 per window (H = 512, heads 2)
    flops = 16 * (35_594_240 * 1250 + 10_240 * (500 ** 2 * 2 + 250 ** 2))
    layer_flops = (4 * 512 * 2048 + 2 * 512 * 512 + 6 * 512 * 512 * 4 / 5) * 16 * 1250
    layer_ms = kernels['ffn'] / max(launches['ffn'], 1)
    if precision == 'fp16x2':        # the FFN is two GEMM launches here, out-projection + LayerNorm a third: no one 'layer launch'
        layer_ms = (kernels['ffn'] + kernels.get('outproj_ln', 0.0) + kernels.get('qkv', 0.0)) / 5
    record = {
        'config': 'configs[2]: w2v2fb representation, batch = 16 x 1000 frames, 1 MI355X', 'dtype': precision, 'w2v2_dtype': w2v2_precision,
        'ppg_network': {
            'ms_per_step': ms, 'frames_per_s': 16000 / ms * 1e3, 'end_to_end_tflops': flops / ms / 1e9,
            'kernel_ms_per_step': kernels,
            'roofline': {'kernel': ('linear_kernel<PrecX2> launches of a layer (Q/K/V, out-projection + LayerNorm, linear-1 + ReLU, linear-2 + LayerNorm; token-split)' if precision == 'fp16x2' else
                                    'layer32_kernel<hidden 512> (feature-split, 96-token workgroups; token-split ffn_kernel in fp32 mode)'), 'bound': 'mfma',
                         'achieved': layer_flops / layer_ms / 1e9, 'peak': PEAK[precision], 'unit': 'TFLOP/s',
                         'frac': layer_flops / layer_ms / 1e9 / PEAK[precision], 'mean_launch_ms': layer_ms}},
    }
    try:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        from oracle import make_golden_w2v2 as M
        hf = M.seeded_model().cuda()
        audio = (0.1 * torch.randn(16, 160080)).cuda()
        encoder = E.W2v2FeatureEncoder(hf.feature_extractor.state_dict(), 0, w2v2_precision)
        conv_flops = 0
        t = 160080
        for layer, (k, s) in enumerate(zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2))):
            t = (t - k) // s + 1
            conv_flops += 2 * t * 512 * k * (1 if layer == 0 else 512)
        conv_flops *= 16
        hip_ms = timed(lambda: encoder(audio))
        # PPGS_BENCH_C3_NATIVE_ONLY=1 (tests/prof_configs.sh): no PyTorch-ROCm comparison legs, so that a rocprofv3 trace
        # of this command holds this package's kernels only
        native_only = bool(os.environ.get('PPGS_BENCH_C3_NATIVE_ONLY'))
        with torch.no_grad():
            extract = encoder(audio) if native_only else hf.feature_extractor(audio).transpose(1, 2)
            torch_ms = body_ms = body16_ms = float('nan')
            if not native_only:
                torch_ms = timed(lambda: hf.feature_extractor(audio), steps=5, warmup=2)
                body_ms = timed(lambda: hf.encoder(hf.feature_projection(extract)[0]), steps=5, warmup=2)
                with torch.autocast('cuda', dtype=torch.float16):
                    body16_ms = timed(lambda: hf.encoder(hf.feature_projection(extract)[0]), steps=5, warmup=2)
        body = E.W2v2Body(hf, 0, w2v2_precision)
        frames = extract.shape[1]
        body_flops = 16 * frames * (2 * 512 * 768 + 2 * 48 * 128 * 768 + 12 * (8 * 768 * 768 + 4 * 768 * 3072 + 4 * frames * 768))
        body_hip_ms = timed(lambda: body(extract, [frames] * 16))
        record['w2v2_transformer_hip'] = {'ms': body_hip_ms, 'tflops': body_flops / body_hip_ms / 1e9, 'flops': body_flops,
                                          'what': 'feature projection + positional convolution + 12 layers (engine.W2v2Body: one launch per GEMM on ppg_gemm32.hip, ppg_posconv.hip)'}
        record['feature_encoder_hip'] = {'ms': hip_ms, 'tflops': conv_flops / hip_ms / 1e9, 'flops': conv_flops}
        record['feature_encoder_pytorch_fp32'] = {'ms': torch_ms, 'tflops': conv_flops / torch_ms / 1e9}
        record['w2v2_transformer_pytorch_fp32'] = {'ms': body_ms}
        record['w2v2_transformer_pytorch_fp16_autocast'] = {'ms': body16_ms}
        record['end_to_end_ms'] = {'all_hip': hip_ms + body_hip_ms + ms, 'native_encoder_fp16_body': hip_ms + body16_ms + ms, 'native_encoder_fp32_body': hip_ms + body_ms + ms,
                                   'all_pytorch_fp32_w2v2': torch_ms + body_ms + ms}
    except Exception as error:                                    # transformers missing: kernel path only
        record['feature_encoder'] = f'skipped: {error}'
    print(json.dumps(record))


if __name__ == '__main__':
    main()
