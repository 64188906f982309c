"""Constants of the PPG inference path.

Values follow the reference configuration modules (reference
ppgs/config/defaults.py:20-32,127-161,170,202 and ppgs/config/static.py:22);
they are plain module constants here -- the engine has no yapecs layer.
"""
import math

# Audio / frontend (reference ppgs/config/defaults.py:20-32)
HOPSIZE = 160
NUM_FFT = 1024
NUM_MELS = 80
SAMPLE_RATE = 16000
WINDOW_SIZE = 1024
NUM_BINS = NUM_FFT // 2 + 1

# Model (reference ppgs/config/defaults.py:127-161)
ATTENTION_HEADS = 2
IS_CAUSAL = False
HIDDEN_CHANNELS = 256
INPUT_CHANNELS = 80
KERNEL_SIZE = 5
NUM_HIDDEN_LAYERS = 5
OUTPUT_CHANNELS = 40
CHUNK_OVERLAP = 50
CHUNK_LENGTH = 500
FFN_CHANNELS = 2048          # torch.nn.TransformerEncoderLayer default
LAYER_NORM_EPS = 1e-5        # torch.nn.TransformerEncoderLayer default
MAX_POSITIONS = 5000         # reference ppgs/model/transformer.py:24

# Batching (reference ppgs/config/defaults.py:170,202; static.py:22)
BUCKETS = 1
RANDOM_SEED = 1234
MAX_INFERENCE_FRAMES = math.inf

REPRESENTATION = 'mel'
BEST_REPRESENTATION = 'mel'

# Per-representation model geometry (reference ppgs/config/w2v2fb.py:7-10,
# ppgs/load.py:36-42)
MODEL_GEOMETRY = {
    'mel': dict(input_channels=80, hidden_channels=256),
    'w2v2fb': dict(input_channels=768, hidden_channels=512),
}

# Exponent of the similarity matrix in ppgs.distance (reference config/defaults.py:214)
SIMILARITY_EXPONENT = 1.2
