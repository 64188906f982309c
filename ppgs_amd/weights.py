"""Checkpoint layout of the PPG network and a seeded synthetic checkpoint.

The key list and shapes are those of the reference ``Transformer``'s
``state_dict`` (reference ppgs/model/transformer.py:15-43, built from
``torch.nn.TransformerEncoderLayer``; enumerated in SURVEY.md 8(b)):

    position.encoding                              (5000, 1, H)
    input_layer.weight / .bias                     (H, Cin, 5) / (H,)
    model.layers.{l}.self_attn.in_proj_weight      (3H, H)    rows [q; k; v]
    model.layers.{l}.self_attn.in_proj_bias        (3H,)
    model.layers.{l}.self_attn.out_proj.weight     (H, H)   / .bias (H,)
    model.layers.{l}.linear1.weight / .bias        (F, H)   / (F,)
    model.layers.{l}.linear2.weight / .bias        (H, F)   / (H,)
    model.layers.{l}.norm1.weight / .bias          (H,)
    model.layers.{l}.norm2.weight / .bias          (H,)
    output_layer.weight / .bias                    (40, H, 5) / (40,)

There is no trained checkpoint offline (the reference downloads
``mel-800k.pt`` from the HF hub, ppgs/load.py:59-63), so benchmarks and parity
tests use :func:`seeded_state_dict`: same architecture, deterministic values.
"""
import math

import torch

from . import config


def positional_encoding(channels, max_len=config.MAX_POSITIONS):
    """Sinusoidal table, (max_len, 1, channels) fp32.

    Same construction as reference ppgs/model/transformer.py:93-100 (the
    table is a registered buffer and therefore part of the checkpoint).
    """
    index = torch.arange(max_len).unsqueeze(1)
    frequency = torch.exp(
        torch.arange(0, channels, 2) * (-math.log(10000.0) / channels))
    encoding = torch.zeros(max_len, 1, channels)
    encoding[:, 0, 0::2] = torch.sin(index * frequency)
    encoding[:, 0, 1::2] = torch.cos(index * frequency)
    return encoding


def state_dict_shapes(
    input_channels=config.INPUT_CHANNELS,
    hidden_channels=config.HIDDEN_CHANNELS,
    num_layers=config.NUM_HIDDEN_LAYERS,
    output_channels=config.OUTPUT_CHANNELS,
    kernel_size=config.KERNEL_SIZE,
    ffn_channels=config.FFN_CHANNELS,
    max_len=config.MAX_POSITIONS,
):
    """Ordered {key: shape} of the reference checkpoint."""
    H, F = hidden_channels, ffn_channels
    shapes = {
        'position.encoding': (max_len, 1, H),
        'input_layer.weight': (H, input_channels, kernel_size),
        'input_layer.bias': (H,),
    }
    for l in range(num_layers):
        p = f'model.layers.{l}.'
        shapes[p + 'self_attn.in_proj_weight'] = (3 * H, H)
        shapes[p + 'self_attn.in_proj_bias'] = (3 * H,)
        shapes[p + 'self_attn.out_proj.weight'] = (H, H)
        shapes[p + 'self_attn.out_proj.bias'] = (H,)
        shapes[p + 'linear1.weight'] = (F, H)
        shapes[p + 'linear1.bias'] = (F,)
        shapes[p + 'linear2.weight'] = (H, F)
        shapes[p + 'linear2.bias'] = (H,)
        shapes[p + 'norm1.weight'] = (H,)
        shapes[p + 'norm1.bias'] = (H,)
        shapes[p + 'norm2.weight'] = (H,)
        shapes[p + 'norm2.bias'] = (H,)
    shapes['output_layer.weight'] = (output_channels, H, kernel_size)
    shapes['output_layer.bias'] = (output_channels,)
    return shapes


def seeded_state_dict(
    seed=config.RANDOM_SEED,
    input_channels=config.INPUT_CHANNELS,
    hidden_channels=config.HIDDEN_CHANNELS,
    num_layers=config.NUM_HIDDEN_LAYERS,
    sharpen=1.0,
):
    """Deterministic synthetic checkpoint in the reference layout.

    Matrices are U(-b, b) with b = sqrt(3 / fan_in) (unit-gain), biases
    U(-0.1, 0.1), LayerNorm affine = 1 + 0.1 N(0,1) / 0.1 N(0,1) so that no
    parameter is at a value (0 or 1) that would hide an indexing bug.
    ``sharpen`` scales every matrix: >1 gives peakier posteriors so that a
    1e-4 tolerance on probabilities is discriminating.
    """
    generator = torch.Generator(device='cpu')
    generator.manual_seed(seed)
    shapes = state_dict_shapes(
        input_channels=input_channels,
        hidden_channels=hidden_channels,
        num_layers=num_layers)
    state = {}
    for key, shape in shapes.items():
        if key == 'position.encoding':
            state[key] = positional_encoding(hidden_channels, shape[0])
        elif 'norm' in key and key.endswith('weight'):
            state[key] = 1. + .1 * torch.randn(shape, generator=generator)
        elif 'norm' in key:
            state[key] = .1 * torch.randn(shape, generator=generator)
        elif key.endswith('bias'):
            state[key] = .2 * torch.rand(shape, generator=generator) - .1
        else:
            fan_in = math.prod(shape[1:])
            bound = sharpen * math.sqrt(3. / fan_in)
            state[key] = bound * (
                2. * torch.rand(shape, generator=generator) - 1.)
    return state


def geometry(state):
    """Infer (input_channels, hidden_channels, num_layers) from a state dict."""
    hidden, cin, _ = state['input_layer.weight'].shape
    layers = 0
    while f'model.layers.{layers}.linear1.weight' in state:
        layers += 1
    return cin, hidden, layers
