"""ppg_attn64.hip alone: the attention kernel of the encoder's whole-batch launches (head dimension 128, 16-bit modes)
against softmax(q k^T + mask) v computed by torch in fp32 from the SAME 16-bit operands.

The engine-level parity tests see attention through five layers of everything else; this one packs random Q | K rows
and V^T in the engine's layouts (ppg_device.h: V^T rows in pair_row order, the columns of every 32-token group in
position order 8 g + 4 e + r <- token 16 e + 4 g + r), launches the kernel through the test-only entry
tools/probes/attn64_probe.hip (tools/bin/libattn64_probe.so, built by __graft_entry__.build()) and compares the
attention output itself, window by window -- ragged windows of 1 .. 500 keys, padding inside windows, the causal
mask, logits far above the first tile's maximum (the re-base path), both output orders.
The reference being replaced: F.multi_head_attention_forward as called by torch's TransformerEncoderLayer
(ppgs/model/transformer.py:74-81 of the reference checkout), scores in log2 units (the engine folds log2(e) / sqrt(d)
into W_q).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, HEADS, DH = 256, 2, 128
QTILE = 256


def probe():
    path = os.path.join(ROOT, 'tools', 'bin', 'libattn64_probe.so')
    if not os.path.exists(path):
        pytest.skip('tools/bin/libattn64_probe.so not built (tools/probes/build.sh; __graft_entry__.build() does it)')
    lib = ctypes.CDLL(path)
    lib.attn64_probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p]
    return lib


def pair_row(j):
    s = j & 31
    return (j & ~31) + 8 * ((s & 15) >> 2) + 4 * (s >> 4) + (s & 3)


def position(k):           # column of token k inside the transposed-V buffer (relative to the window's first column)
    t = k & 31
    e, g, r = t >> 4, (t >> 2) & 3, t & 3
    return (k & ~31) + 8 * g + 4 * e + r


def ao32_byte(m, n, hidden=256):
    toks, ks_n, tb_n = 160, hidden // 16, 5
    tile, r = divmod(m, toks)
    tb, tok = r >> 5, r & 31
    return ((((tile * tb_n + tb) * ks_n + (n >> 4)) * 64 + ((n >> 3) & 1) * 32 + tok) * 16)


def run_case(windows, dtype, causal, scale=1.0, tiled=False, seed=0, spike=None):
    """windows: [(frames, valid)].  -> (max abs error over all valid query rows, max |reference|)"""
    lib = probe()
    gen = torch.Generator().manual_seed(seed)
    precision = 1 if dtype == torch.bfloat16 else 2
    tok_off, vt_off, M, Mvt = [], [], 0, 0
    for frames, _ in windows:
        tok_off.append(M)
        vt_off.append(Mvt)
        M += (frames + 15) // 16 * 16
        Mvt += (frames + 31) // 32 * 32
    Mt = (M + 159) // 160 * 160
    qk = torch.zeros(M + 64, 2 * H, dtype=dtype)
    vt = torch.zeros(H, Mvt + 64, dtype=dtype)
    rho = torch.tensor([pair_row(j) for j in range(H)])
    items, refs = [], []
    for wi, (frames, valid) in enumerate(windows):
        f16, f32 = (frames + 15) // 16 * 16, (frames + 31) // 32 * 32
        q = (scale * 0.35 * torch.randn(f16, H, generator=gen)).to(dtype)
        k = torch.randn(f16, H, generator=gen).to(dtype)
        v = torch.randn(f32, H, generator=gen).to(dtype)
        if spike is not None:                # one key far above the others, past the first tile (the re-base path)
            kq, amp = spike
            if kq < valid:
                k[kq] = (amp * q[min(kq, frames - 1)].float() / max(float(q[min(kq, frames - 1)].float().norm()), 1e-3)).to(dtype)
        qk[tok_off[wi]:tok_off[wi] + f16, :H] = q
        qk[tok_off[wi]:tok_off[wi] + f16, H:] = k
        pos = torch.tensor([position(t) for t in range(f32)])
        block = torch.empty(H, f32, dtype=dtype)
        block[:, pos] = v[:, rho].T
        vt[:, vt_off[wi]:vt_off[wi] + f32] = block
        for q0 in range(0, frames, QTILE):
            items.append([wi, q0, tok_off[wi], vt_off[wi], frames, valid, 0, 0])
        # reference (fp32, from the 16-bit operands)
        out = torch.zeros(frames, H, dtype=torch.float64)
        q64, k64, v64 = q.double(), k.double(), v.double()
        for h in range(HEADS):
            sl = slice(h * DH, (h + 1) * DH)
            s = q64[:frames, sl].contiguous() @ k64[:valid, sl].contiguous().T.contiguous()
            if causal:
                qi = torch.arange(frames)[:, None]
                s = s.masked_fill(torch.arange(valid)[None, :] > qi, float('-inf'))
            p = torch.softmax(s * float(np.log(2.0)), dim=1)
            out[:, sl] = p @ v64[:valid, sl].contiguous()
        refs.append(out.float())
    items.sort(key=lambda it: -it[5])
    d_items = torch.tensor(items, dtype=torch.int32).cuda()
    d_qk, d_vt = qk.cuda(), vt.cuda()
    ao = torch.full((Mt * H,), float('nan'), dtype=dtype).cuda()
    torch.cuda.synchronize()                 # (the uploads above are complete before the raw launch)
    rc = lib.attn64_probe_launch(precision, d_qk.data_ptr(), 2 * H * 2, d_vt.data_ptr(), (Mvt + 64) * 2, ao.data_ptr(), H,
                                 int(causal), d_items.data_ptr(), len(items), HEADS, M, int(tiled), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    got = ao.cpu()
    if tiled:
        index = torch.tensor([[ao32_byte(m, n) // 2 for n in range(0, H, 8)] for m in range(M)])
        rows = got[(index[:, :, None] + torch.arange(8)[None, None, :]).reshape(M, H)]
    else:
        rows = got[:M * H].reshape(M, H)
    worst, biggest = 0.0, 0.0
    for wi, (frames, valid) in enumerate(windows):
        mine = rows[tok_off[wi]:tok_off[wi] + frames].float()
        rows_ok = valid if causal else frames            # (causal rows past `valid` see keys the mask removed only partly: skip)
        if causal:
            rows_ok = frames
        assert bool(torch.isfinite(mine).all()), (wi, frames, valid)
        worst = max(worst, float((mine[:rows_ok] - refs[wi][:rows_ok]).abs().max()))
        biggest = max(biggest, float(refs[wi].abs().max()))
    return worst, biggest


# 16-bit P and 16-bit output: relative to outputs of magnitude ~1..3
TOL = {torch.float16: 4e-3, torch.bfloat16: 3e-2}


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('causal', [False, True])
def test_attn64_ragged_windows_vs_torch(dtype, causal):
    windows = [(500, 500), (500, 437), (250, 250), (256, 256), (257, 257), (64, 64), (65, 64), (30, 30), (1, 1), (16, 7),
               (129, 129), (320, 300), (192, 192), (448, 448)]
    err, mag = run_case(windows, dtype, causal)
    assert mag > 0.5
    assert err < TOL[dtype], (err, mag)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_attn64_tiled_output_order(dtype):
    err, _ = run_case([(500, 500), (250, 201), (96, 96)], dtype, False, tiled=True, seed=3)
    assert err < TOL[dtype], err


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('causal', [False, True])
def test_attn64_rebase_paths(monkeypatch, dtype, causal):
    """Logits far above the first tile's maximum (a spiked key in a later tile: p passes the ceiling, the shift re-bases),
    and the always-re-base policy of the tests (PPGS_AMD_ATTN_REBASE=always)."""
    windows = [(500, 500), (300, 260), (130, 130)]
    err, _ = run_case(windows, dtype, causal, scale=4.0, seed=5, spike=(100, 60.0))
    assert err < 2 * TOL[dtype], err
    monkeypatch.setenv('PPGS_AMD_ATTN_REBASE', 'always')
    err, _ = run_case(windows, dtype, causal, seed=6)
    assert err < TOL[dtype], err
