"""Golden vectors captured THROUGH THE REFERENCE'S GENUINE GLUE (SURVEY.md 8(c),
fixture G7): ``ppgs.from_audio`` / ``ppgs.from_features`` / ``ppgs.infer`` of
/root/reference/ppgs/core.py:22-128,551-596 and ``ppgs.preprocess.from_audio``
of ppgs/preprocess/core.py:194-216, loaded unmodified from where they lie and run
with stubs for the THIRD-PARTY packages only (torchaudio: empty; torchutil:
``inference.context`` = eval + inference_mode + autocast(switch), ``notify`` =
identity; librosa.filters.mel: HF's implementation, as in make_golden.py) and a
``ppgs.load.model`` shim that returns the reference ``Transformer`` holding the
seeded weights.

    python oracle/make_golden_entry.py     # writes tests/golden/g7_glue.npz, g6_sharp_stats.npz

Runs only in the build container (needs /root/reference).

g7_glue.npz
  audio (1,1,16000), config C1
  ppg_shipped   ppgs.from_audio(audio, 16000, gpu=None) AS SHIPPED: CPU autocast
                (bf16) in preprocess and in infer -> (1,40,100) bfloat16, stored
                as fp32 (informational: how far the reference's own shipped
                arithmetic is from its fp32 arithmetic)
  ppg_fp32      the fp32 route through the same glue:
                ppgs.from_features(mel.from_audios(audio, n).float(), frames)
                with the autocast switch OFF (graded, 1e-4)
  ppg_sharp_*   the same two with the sharpened checkpoint
  logits_fp32   from_features(..., softmax=False)
  batch_*       from_features on a ragged (3,80,160) batch through the glue
g6_sharp_stats.npz
  C2-size (32 x 1000) statistics of the SHARPENED checkpoint through the fp32
  route (per-item argmax track of 4 items, mean/max, first/last 64 frames), and
  the same through the shipped bf16 autocast (argmax agreement of the
  reference's own reduced-precision arithmetic with its fp32 arithmetic).
"""
import contextlib
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as G          # noqa: E402
from ppgs_amd import weights as W            # noqa: E402

AUTOCAST = {'enabled': True}


def import_glue():
    ppgs = G.import_reference()
    sys.modules['torchaudio'] = types.ModuleType('torchaudio')
    torchutil = types.ModuleType('torchutil')
    torchutil.notify = lambda *a, **k: (lambda f: f)
    torchutil.iterator = lambda iterable, *a, **k: iterable
    torchutil.inference = types.ModuleType('torchutil.inference')

    @contextlib.contextmanager
    def context(model):
        # torchutil.inference.context: eval mode + inference_mode + autocast
        model.eval()
        with torch.inference_mode(), torch.autocast(
                next(model.parameters()).device.type, enabled=AUTOCAST['enabled']):
            yield
    torchutil.inference.context = context
    sys.modules['torchutil'] = torchutil
    sys.modules['torchutil.inference'] = torchutil.inference
    ppgs.PHONEMES = [str(i) for i in range(40)]
    core = G._load('ppgs.core', os.path.join(G.REF, 'ppgs', 'core.py'))
    for name in ('from_audio', 'from_features', 'infer', 'resample'):
        setattr(ppgs, name, getattr(core, name))
    pre = G._load('ppgs.preprocess.core', os.path.join(G.REF, 'ppgs', 'preprocess', 'core.py'))
    ppgs.preprocess.from_audio = pre.from_audio
    ppgs.preprocess.save_masked = pre.save_masked
    ppgs.load = types.ModuleType('ppgs.load')
    sys.modules['ppgs.load'] = ppgs.load
    return ppgs, core


def main():
    ppgs, core = import_glue()
    states = {
        'seeded': W.seeded_state_dict(seed=1234),
        'sharp': W.seeded_state_dict(seed=4321, sharpen=2.0)}
    models = {k: G.reference_model(ppgs, v) for k, v in states.items()}

    def use(name):
        ppgs.load.model = lambda checkpoint=None, representation=None: models[name]
        for attr in ('models', 'model', 'checkpoint', 'representation', 'device_type'):
            if hasattr(core.infer, attr):
                delattr(core.infer, attr)

    mel = ppgs.preprocess.mel
    audio = 0.1 * G.randn(71, 1, 1, 16000)
    out = dict(audio=audio)
    for name, tag in (('seeded', ''), ('sharp', '_sharp')):
        use(name)
        AUTOCAST['enabled'] = True
        shipped = core.from_audio(audio, 16000, gpu=None)
        assert shipped.dtype == torch.bfloat16 and shipped.shape == (1, 40, 100)
        AUTOCAST['enabled'] = False
        feats = mel.from_audios(audio, 16000).float()
        fp32 = core.from_features(feats, torch.tensor([100]), gpu=None)
        assert fp32.dtype == torch.float32
        out[f'ppg_shipped{tag}'] = shipped.float()
        out[f'ppg_fp32{tag}'] = fp32
        print(name, 'shipped vs fp32 max-abs', float((shipped.float() - fp32).abs().max()))
    use('seeded')
    AUTOCAST['enabled'] = False
    feats = mel.from_audios(audio, 16000).float()
    out['logits_fp32'] = core.from_features(feats, torch.tensor([100]), gpu=None, softmax=False)
    batch = G.randn(72, 3, 80, 160).half()
    lengths = torch.tensor([160, 121, 16])
    out['batch_features'] = batch
    out['batch_lengths'] = lengths
    out['batch_ppg_fp32'] = core.from_features(batch.float(), lengths.clone(), gpu=None)
    AUTOCAST['enabled'] = True
    out['batch_ppg_shipped'] = core.from_features(batch, lengths.clone(), gpu=None).float()
    G.save('g7_glue', **out)

    # C2-size statistics, sharpened checkpoint
    use('sharp')
    audio = 0.1 * G.randn(1234, 32, 1, 160000)
    feats16 = mel.from_audios(audio, 160000)
    lengths = torch.full((32,), 1000, dtype=torch.long)
    AUTOCAST['enabled'] = False
    ppg = core.from_features(feats16.float(), lengths.clone(), gpu=None)
    AUTOCAST['enabled'] = True
    ppg_bf16 = core.from_features(feats16, lengths.clone(), gpu=None).float()
    top2 = ppg.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    agree = (ppg.argmax(1) == ppg_bf16.argmax(1))
    print('sharp C2: reference bf16-autocast vs fp32 max-abs', float((ppg - ppg_bf16).abs().max()),
          'argmax agreement', float(agree.float().mean()),
          'frames with margin < 0.02:', float((margin < 0.02).float().mean()))
    G.save('g6_sharp_stats',
           ppg_mean=ppg.mean(dim=-1), ppg_max=ppg.amax(dim=-1),
           argmax=ppg.argmax(1).to(torch.int8),            # (32, 1000) phoneme track
           margin=margin.half(),                           # top-1 minus top-2 posterior
           shipped_argmax_agreement=agree.float().mean(),
           shipped_max_abs=(ppg - ppg_bf16).abs().max(),
           ppg_item0_first64=ppg[0, :, :64], ppg_item31_last64=ppg[31, :, -64:])


if __name__ == '__main__':
    main()
