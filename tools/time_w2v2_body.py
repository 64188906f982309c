"""Wall time of the HIP wav2vec2 body at configs[2] size (16 x 499 frames): python tools/time_w2v2_body.py [precision]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppgs_amd import engine as E    # noqa: E402
import transformers                 # noqa: E402
transformers.utils.logging.set_verbosity_error()
torch.manual_seed(5)
model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config()).eval().cuda()
body = E.W2v2Body(model, 0, sys.argv[1] if len(sys.argv) > 1 else 'bf16')
B = int(os.environ.get("BODY_BATCH", 16))
x = torch.randn(B, 499, 512).cuda()
for _ in range(5):
    body(x, [499] * B)
torch.cuda.synchronize()
start = time.perf_counter()
for _ in range(20):
    body(x, [499] * B)
torch.cuda.synchronize()
print(f'{(time.perf_counter() - start) / 20 * 1e3:.3f} ms')
