import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppgs_amd import engine as E
import transformers
transformers.utils.logging.set_verbosity_error()
torch.manual_seed(5)
model = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config()).eval().cuda()
body = E.W2v2Body(model, 0, 'bf16')
x = torch.randn(16, 499, 512).cuda()
for _ in range(8):
    body(x, [499] * 16)
torch.cuda.synchronize()
