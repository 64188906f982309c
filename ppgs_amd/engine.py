"""ctypes binding of the C ABI in include/ppgs_amd.h (libppgs_amd.so).

PyTorch-ROCm supplies device memory (tensors), the current HIP stream and the
device index; all arithmetic happens in the HIP library.  There is no CPU or
eager-PyTorch fallback: if the library is missing or no GPU is visible the
calls raise.
"""
import ctypes
import os
import threading

import torch

from . import config, weights

PPG_MAX_LAYERS = 16
PRECISIONS = {'fp32': 0, 'bf16': 1, 'fp16': 2, 'fp16x2': 3}
KERNEL_CLASSES = (
    'gather', 'inconv', 'qkv', 'attention', 'outproj_ln', 'ffn',
    'outconv_softmax', 'frontend')

# PPGS_AMD_LIB: an alternative build of the same library (kernel experiments)
_LIB_PATH = os.environ.get('PPGS_AMD_LIB') or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), 'libppgs_amd.so')
_lib = None
_lib_lock = threading.Lock()

_FP = ctypes.POINTER(ctypes.c_float)


class PpgConfig(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in (
        'input_channels', 'hidden_channels', 'num_layers', 'ffn_channels',
        'output_channels', 'kernel_size', 'heads', 'is_causal',
        'max_positions', 'chunk_length', 'chunk_overlap', 'precision')]


class PpgWeights(ctypes.Structure):
    _fields_ = (
        [('position_encoding', _FP), ('input_weight', _FP), ('input_bias', _FP)] +
        [(name, _FP * PPG_MAX_LAYERS) for name in (
            'in_proj_weight', 'in_proj_bias', 'out_proj_weight',
            'out_proj_bias', 'linear1_weight', 'linear1_bias',
            'linear2_weight', 'linear2_bias', 'norm1_weight', 'norm1_bias',
            'norm2_weight', 'norm2_bias')] +
        [('output_weight', _FP), ('output_bias', _FP)])


class PpgWindow(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in (
        'item', 'chunked', 'start', 'frames', 'valid', 'keep_lo', 'keep_hi',
        'out_frame', 'tok_off', 'vt_off', 'pad0', 'pad1')]


class PpgW2v2LayerWeights(ctypes.Structure):
    _fields_ = [(name, ctypes.POINTER(ctypes.c_float)) for name in (
        'q_weight', 'q_bias', 'k_weight', 'k_bias', 'v_weight', 'v_bias', 'out_weight', 'out_bias',
        'norm1_weight', 'norm1_bias', 'ffn1_weight', 'ffn1_bias', 'ffn2_weight', 'ffn2_bias',
        'norm2_weight', 'norm2_bias')]


class PpgW2v2BodyWeights(ctypes.Structure):
    _fields_ = [
        ('hidden', ctypes.c_int32), ('heads', ctypes.c_int32), ('ffn', ctypes.c_int32),
        ('num_layers', ctypes.c_int32), ('conv_kernel', ctypes.c_int32), ('conv_groups', ctypes.c_int32),
        ('layer_norm_eps', ctypes.c_float), ('pad0', ctypes.c_int32)] + [
        (name, ctypes.POINTER(ctypes.c_float)) for name in (
            'proj_norm_weight', 'proj_norm_bias', 'proj_weight', 'proj_bias',
            'pos_conv_weight', 'pos_conv_bias', 'enc_norm_weight', 'enc_norm_bias')] + [
        ('layers', PpgW2v2LayerWeights * 24)]


class PpgAttentionItem(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in (
        'window', 'q0', 'queries', 'frames', 'valid', 'narrow')]


class PpgPlanInfo(ctypes.Structure):
    _fields_ = [
        ('num_windows', ctypes.c_int32), ('skipped_windows', ctypes.c_int32),
        ('tokens', ctypes.c_int32), ('vt_tokens', ctypes.c_int32),
        ('processed_frames', ctypes.c_int64),
        ('attention_pairs', ctypes.c_int64),
        ('workspace_bytes', ctypes.c_size_t)]


# every symbol include/ppgs_amd.h declares: name -> (restype, argtypes)
_I64P = ctypes.POINTER(ctypes.c_int64)
SYMBOLS = {
    'ppg_last_error': (ctypes.c_char_p, []),
    'ppg_abi_version': (ctypes.c_int, []),
    'ppg_engine_create': (ctypes.c_int, [
        ctypes.POINTER(PpgConfig), ctypes.POINTER(PpgWeights), ctypes.c_int,
        ctypes.POINTER(ctypes.c_void_p)]),
    'ppg_engine_destroy': (None, [ctypes.c_void_p]),
    'ppg_plan_windows': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _I64P, ctypes.c_int,
        ctypes.POINTER(PpgWindow), ctypes.c_int, ctypes.POINTER(PpgPlanInfo)]),
    'ppg_plan_attention_items': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _I64P, ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(PpgAttentionItem), ctypes.c_int]),
    'ppg_workspace_bytes': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _I64P, ctypes.c_int,
        ctypes.POINTER(ctypes.c_size_t)]),
    'ppg_encode': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, _I64P, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'ppg_frontend': (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'ppg_resample_length': (ctypes.c_int64, [
        ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'ppg_resample': (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'ppg_distance': (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'ppg_sparsify': (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    'ppg_grid_sample': (ctypes.c_int, [
        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'ppg_w2v2_body_create': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'ppg_w2v2_body_destroy': (None, [ctypes.c_void_p]),
    'ppg_w2v2_body_workspace_bytes': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    'ppg_w2v2_body_forward': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, _I64P, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'ppg_stream_create': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'ppg_stream_destroy': (None, [ctypes.c_void_p]),
    'ppg_stream_rows': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
        ctypes.POINTER(ctypes.c_int)]),
    'ppg_stream_posteriors': (ctypes.c_void_p, [ctypes.c_void_p]),
    'ppg_stream_push': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    'ppg_engine_nonfinite': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'ppg_engine_pipelines': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'ppg_stream_create_batch': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'ppg_stream_batch': (ctypes.c_int, [ctypes.c_void_p]),
    'ppg_stream_push_batch': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
        ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    'ppg_w2v2_create': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'ppg_w2v2_destroy': (None, [ctypes.c_void_p]),
    'ppg_w2v2_frames': (ctypes.c_int64, [ctypes.c_int64]),
    'ppg_w2v2_workspace_bytes': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t)]),
    'ppg_w2v2_features': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'ppg_engine_profile': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'ppg_engine_profile_read': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
        _I64P]),
    'ppg_engine_profile_reset': (ctypes.c_int, [ctypes.c_void_p]),
    'ppg_engine_profile_stride': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'ppg_frontend_profile': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'ppg_frontend_profile_read': (ctypes.c_int, [
        ctypes.c_int, ctypes.POINTER(ctypes.c_double), _I64P]),
    'ppg_io_last_error': (ctypes.c_char_p, []),
    'ppg_wav_info': (ctypes.c_int, [
        ctypes.c_char_p, _I64P, ctypes.POINTER(ctypes.c_int32),
        ctypes.POINTER(ctypes.c_int32)]),
    'ppg_wav_read_batch': (ctypes.c_int, [
        ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int64, _I64P,
        ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    'ppg_pt_write_batch': (ctypes.c_int, [
        ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_int64, _I64P, ctypes.c_int]),
}


def library_sha16(path=None):
    """First 16 hex digits of the sha256 of the built library (a clean `make` reproduces it byte for byte): what binds a
    committed rocprofv3 counter summary (profiles/r*_pmc_summary*.txt, header `# lib_sha256_16=`) to a build."""
    import hashlib
    with open(path or _LIB_PATH, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def library():
    """Load libppgs_amd.so (built in-tree by __graft_entry__.build / make)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RuntimeError(
                    f'{_LIB_PATH} is missing: build it with '
                    '`python -c "import __graft_entry__ as g; g.build()"` or '
                    '`make -C ppgs_amd/csrc`. The PPG engine has no fallback '
                    'path.')
            lib = ctypes.CDLL(_LIB_PATH)
            for name, (restype, argtypes) in SYMBOLS.items():
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


class PpgError(RuntimeError):
    pass


_ERRORS = {-1: ValueError, -2: PpgError, -3: PpgError, -4: ValueError}


def _check(code):
    if code < 0:
        message = library().ppg_last_error().decode()
        raise _ERRORS.get(code, PpgError)(f'ppgs_amd: {message} (code {code})')
    return code


def _lengths_array(lengths, batch):
    if isinstance(lengths, torch.Tensor):
        lengths = lengths.detach().cpu().reshape(-1).tolist()
    elif isinstance(lengths, int):
        lengths = [lengths]
    lengths = [int(v) for v in lengths]
    if len(lengths) != batch:
        raise ValueError(
            f'lengths has {len(lengths)} entries for a batch of {batch}')
    return (ctypes.c_int64 * batch)(*lengths)


def plan_windows(batch, frames, lengths, legacy_mode=False, engine=None):
    """Host-only chunk plan (works without a GPU): (windows, info)."""
    lib = library()
    arr = _lengths_array(lengths, batch)
    info = PpgPlanInfo()
    handle = engine._handle if engine is not None else None
    count = _check(lib.ppg_plan_windows(
        handle, batch, frames, arr, int(legacy_mode), None, 0,
        ctypes.byref(info)))
    windows = (PpgWindow * max(count, 1))()
    _check(lib.ppg_plan_windows(
        handle, batch, frames, arr, int(legacy_mode), windows, count,
        ctypes.byref(info)))
    return list(windows)[:count], info


def plan_attention_items(batch, frames, lengths, legacy_mode=False, heads=config.ATTENTION_HEADS, engine=None):
    """Host-only: the attention work items of a batch in launch order (ppg_plan_attention_items)."""
    lib = library()
    arr = _lengths_array(lengths, batch)
    handle = engine._handle if engine is not None else None
    count = _check(lib.ppg_plan_attention_items(handle, batch, frames, arr, int(legacy_mode), heads, None, 0))
    items = (PpgAttentionItem * max(count, 1))()
    _check(lib.ppg_plan_attention_items(handle, batch, frames, arr, int(legacy_mode), heads, items, count))
    return list(items)[:count]


class Engine:
    """One loaded PPG network on one GPU (mirrors the per-model cache of
    reference ppgs.infer, ppgs/core.py:565-583)."""

    def __init__(self, state, device=0, precision='bf16', is_causal=False,
                 heads=config.ATTENTION_HEADS):
        if not torch.cuda.is_available():
            raise PpgError(
                'ppgs_amd: no HIP device visible; the engine has no CPU path')
        lib = library()
        cin, hidden, layers = weights.geometry(state)
        self.input_channels = cin
        self.hidden_channels = hidden
        self.output_channels = state['output_layer.weight'].shape[0]
        self.precision = precision
        self.device = torch.device('cuda', device)
        cfg = PpgConfig(
            input_channels=cin, hidden_channels=hidden, num_layers=layers,
            ffn_channels=(
                state['model.layers.0.linear1.weight'].shape[0]
                if layers else config.FFN_CHANNELS),
            output_channels=self.output_channels,
            kernel_size=state['input_layer.weight'].shape[2], heads=heads,
            is_causal=int(is_causal),
            max_positions=state['position.encoding'].shape[0],
            chunk_length=config.CHUNK_LENGTH,
            chunk_overlap=config.CHUNK_OVERLAP,
            precision=PRECISIONS[precision])
        keep = []

        def ptr(key):
            tensor = state[key].detach().to('cpu', torch.float32).contiguous()
            keep.append(tensor)
            return ctypes.cast(tensor.data_ptr(), _FP)
        wts = PpgWeights()
        wts.position_encoding = ptr('position.encoding')
        wts.input_weight = ptr('input_layer.weight')
        wts.input_bias = ptr('input_layer.bias')
        wts.output_weight = ptr('output_layer.weight')
        wts.output_bias = ptr('output_layer.bias')
        names = {
            'in_proj_weight': 'self_attn.in_proj_weight',
            'in_proj_bias': 'self_attn.in_proj_bias',
            'out_proj_weight': 'self_attn.out_proj.weight',
            'out_proj_bias': 'self_attn.out_proj.bias',
            'linear1_weight': 'linear1.weight', 'linear1_bias': 'linear1.bias',
            'linear2_weight': 'linear2.weight', 'linear2_bias': 'linear2.bias',
            'norm1_weight': 'norm1.weight', 'norm1_bias': 'norm1.bias',
            'norm2_weight': 'norm2.weight', 'norm2_bias': 'norm2.bias'}
        for field, key in names.items():
            array = getattr(wts, field)
            for l in range(layers):
                array[l] = ptr(f'model.layers.{l}.{key}')
        handle = ctypes.c_void_p()
        _check(lib.ppg_engine_create(
            ctypes.byref(cfg), ctypes.byref(wts), device, ctypes.byref(handle)))
        self._handle = handle
        self._lib = lib
        self._workspaces = {}         # one scratch buffer per HIP stream in use

    def __del__(self):
        handle = getattr(self, '_handle', None)
        if handle:
            self._lib.ppg_engine_destroy(handle)
            self._handle = None

    def workspace_bytes(self, batch, frames, lengths, legacy_mode=False):
        size = ctypes.c_size_t()
        _check(self._lib.ppg_workspace_bytes(
            self._handle, batch, frames, _lengths_array(lengths, batch),
            int(legacy_mode), ctypes.byref(size)))
        return size.value

    def stream(self, max_frames, dtype=torch.float16):
        """A KV-cached causal stream over one utterance of up to `max_frames` frames
        (see Stream); the engine must be causal."""
        return Stream(self, max_frames, dtype)

    def pipelines(self, tokens):
        """HIP streams a batch of `tokens` token rows (PlanInfo.tokens) is split over (1 or 2)."""
        return int(self._lib.ppg_engine_pipelines(self._handle, int(tokens)))

    def nonfinite(self, clear=True):
        """True when a launch of this engine produced a non-finite logit for a valid frame since the
        flag was last cleared (synchronises with the device): fp16 operands past 65504, or
        non-finite input features."""
        flag = ctypes.c_int()
        with torch.cuda.device(self.device):
            _check(self._lib.ppg_engine_nonfinite(self._handle, int(clear), ctypes.byref(flag)))
        return bool(flag.value)

    def check_finite(self):
        """Raise PpgError if nonfinite() (and clear the flag)."""
        if self.nonfinite(clear=True):
            raise PpgError(
                f'ppgs_amd: non-finite logits ({self.precision} operands): an activation left the operand '
                "format's range (fp16: |x| < 65504) or the input features are not finite; "
                'PPGS_AMD_PRECISION=bf16 or fp32 has the range of fp32')

    def batched_stream(self, batch, max_frames, dtype=torch.float16):
        """KV-cached causal streams over `batch` utterances advanced together (see
        BatchedStream); the engine must be causal."""
        return BatchedStream(self, batch, max_frames, dtype)

    def long_stream(self, dtype=torch.float16):
        """A KV-cached causal stream over an utterance of any length > 500 frames
        (see LongStream); the engine must be causal."""
        return LongStream(self, dtype)

    def encode(self, features, lengths, softmax=True, legacy_mode=False,
               workspace=None):
        """features (B, Cin, T) fp16/fp32 on this engine's GPU -> (B, 40, T)
        fp32 on the same GPU (posteriors, or logits if softmax=False).
        `workspace`: a caller-owned uint8 scratch tensor of at least
        workspace_bytes(); default: one grow-only buffer per HIP stream."""
        if features.dim() != 3 or features.shape[1] != self.input_channels:
            raise ValueError(
                f'features must be (batch, {self.input_channels}, frames), '
                f'got {tuple(features.shape)}')
        features = features.to(self.device)
        if features.dtype == torch.float16:
            dtype = 0
        else:
            features = features.to(torch.float32)
            dtype = 1
        features = features.contiguous()
        batch, _, frames = features.shape
        arr = _lengths_array(lengths, batch)
        size = ctypes.c_size_t()
        _check(self._lib.ppg_workspace_bytes(
            self._handle, batch, frames, arr, int(legacy_mode),
            ctypes.byref(size)))
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            # encodes issued on different streams may overlap: each stream
            # gets its own scratch buffer (grow-only, reused call to call)
            if workspace is not None:
                if workspace.numel() < size.value or workspace.device != self.device:
                    raise ValueError(
                        f'workspace of {workspace.numel()} bytes, {size.value} needed')
            else:
                workspace = self._workspaces.get(stream)
                if workspace is None or workspace.numel() < size.value:
                    # (zeros: nothing reads a byte it has not written -- test_poisoned_workspace -- but a grow-only
                    # buffer allocated once may as well start defined)
                    workspace = torch.zeros(
                        max(size.value, 256), dtype=torch.uint8, device=self.device)
                    # a buffer allocated under stream capture lives in the graph's
                    # private pool: never cache it for later eager calls
                    if not torch.cuda.is_current_stream_capturing():
                        self._workspaces[stream] = workspace
            out = torch.empty(
                (batch, self.output_channels, frames), dtype=torch.float32,
                device=self.device)
            _check(self._lib.ppg_encode(
                self._handle, features.data_ptr(), dtype, arr, batch, frames,
                int(softmax), int(legacy_mode), out.data_ptr(),
                workspace.data_ptr(), workspace.numel(), stream))
        return out

    # -- HIP-graph replay for launch-bound shapes ---------------------------
    def graphed(self, batch, frames, lengths=None, softmax=True,
                legacy_mode=False, feature_dtype=torch.float16):
        """A replayable encode for one fixed (batch, frames, lengths) shape:
        the ~30 launches of ppg_encode captured once into a HIP graph (the
        plan is cached and nothing inside allocates or synchronises), e.g.
        the streaming configuration -- 64 causal 160-frame chunks per step --
        where launch latency, not kernel time, sets the step rate.

        Returns ``run(features) -> posteriors``; the returned tensor is a
        static buffer overwritten by the next call.  ``run.static_input`` is
        the buffer the graph reads: a producer that writes it in place calls
        ``run()`` without an argument and saves the copy.
        """
        lengths = [frames] * batch if lengths is None else lengths
        static_in = torch.zeros(
            (batch, self.input_channels, frames), dtype=feature_dtype,
            device=self.device)
        # the graph gets a scratch buffer of its own (two graphs of one engine replayed on
        # different streams must not share one), allocated before the capture starts
        scratch = torch.empty(
            max(self.workspace_bytes(batch, frames, lengths, legacy_mode), 256),
            dtype=torch.uint8, device=self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):                       # the window plan is built, uploaded and cached here
                self.encode(static_in, lengths, softmax, legacy_mode, workspace=scratch)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            # (under capture ppg_encode pins the cached plan: the graph holds its device pointers)
            static_out = self.encode(static_in, lengths, softmax, legacy_mode, workspace=scratch)

        def run(features=None):
            # features=None: the caller's producer (e.g. the mel frontend) wrote run.static_input in place
            if features is not None:
                static_in.copy_(features, non_blocking=True)
            graph.replay()
            return static_out
        run.graph = graph
        run.scratch = scratch
        run.static_input = static_in
        run.static_output = static_out
        return run

    # -- per-kernel HIP-event timing (bench.py roofline leg) ----------------
    def profile(self, enable=True, classes=None, stride=1):
        """Time launches with HIP events: every kernel class, or only the
        named ones (each timed launch adds two event records to the stream);
        stride n times every n-th launch of a class only."""
        _check(self._lib.ppg_engine_profile_stride(self._handle, int(stride)))
        mask = 0
        if enable:
            mask = -1 if classes is None else sum(
                1 << KERNEL_CLASSES.index(name) for name in classes)
        _check(self._lib.ppg_engine_profile(self._handle, mask))
        _check(self._lib.ppg_engine_profile_reset(self._handle))

    def profile_read(self):
        """{kernel class: (total ms, launches)} since profile(True)."""
        result = {}
        for index, name in enumerate(KERNEL_CLASSES[:-1]):
            total, count = ctypes.c_double(), ctypes.c_int64()
            _check(self._lib.ppg_engine_profile_read(
                self._handle, index, ctypes.byref(total), ctypes.byref(count)))
            result[name] = (total.value, count.value)
        return result


def frontend(audio, spectrogram=False, mel=True):
    """audio (B, 1, N) or (B, N) fp32 on a GPU -> fp16 spectrogram (B,513,T)
    and/or log-mel (B,80,T) on the same GPU, T = N // 160."""
    if not audio.is_cuda:
        raise PpgError('ppgs_amd: frontend input must live on a HIP device')
    if audio.dim() == 3:
        if audio.shape[1] != 1:
            raise ValueError(f'audio must be (batch, 1, samples), got {tuple(audio.shape)}')
        audio = audio[:, 0]
    audio = audio.to(torch.float32).contiguous()
    batch, samples = audio.shape
    frames = samples // config.HOPSIZE
    spec_out = mel_out = None
    if spectrogram:
        spec_out = torch.empty(
            (batch, config.NUM_BINS, frames), dtype=torch.float16,
            device=audio.device)
    if mel:
        mel_out = torch.empty(
            (batch, config.NUM_MELS, frames), dtype=torch.float16,
            device=audio.device)
    lib = library()
    with torch.cuda.device(audio.device):
        stream = torch.cuda.current_stream().cuda_stream
        _check(lib.ppg_frontend(
            audio.device.index, audio.data_ptr(), batch, samples,
            spec_out.data_ptr() if spectrogram else None,
            mel_out.data_ptr() if mel else None, stream))
    return spec_out, mel_out


def resample(audio, sample_rate, target_rate=config.SAMPLE_RATE):
    """(..., samples) fp32 on a GPU at sample_rate -> (..., samples') at
    target_rate, torchaudio.transforms.Resample defaults (ppg_resample)."""
    if not audio.is_cuda:
        raise PpgError('ppgs_amd: the native resampler works on a HIP device tensor')
    shape = audio.shape
    flat = audio.reshape(-1, shape[-1]).to(torch.float32).contiguous()
    lib = library()
    length = lib.ppg_resample_length(flat.shape[1], int(sample_rate), int(target_rate))
    out = torch.empty((flat.shape[0], length), dtype=torch.float32, device=audio.device)
    with torch.cuda.device(audio.device):
        _check(lib.ppg_resample(
            audio.device.index, flat.data_ptr(), flat.shape[0], flat.shape[1],
            int(sample_rate), int(target_rate), out.data_ptr(),
            torch.cuda.current_stream().cuda_stream))
    return out.reshape(shape[:-1] + (length,))


def distance_frames(ppg_x, ppg_y, mix=None):
    """Per-frame similarity-weighted Jensen-Shannon term (ppg_distance):
    (40, frames) x 2 on a GPU [+ (40, 40) mixing matrix] -> (frames,)."""
    if not (ppg_x.is_cuda and ppg_y.is_cuda):
        raise PpgError('ppgs_amd: the post-ops work on HIP device tensors')
    x = ppg_x.to(torch.float32).contiguous()
    y = ppg_y.to(torch.float32).contiguous()
    if x.shape != y.shape or x.dim() != 2 or x.shape[0] != 40:
        raise ValueError(f'PPGs must both be (40, frames), got {tuple(x.shape)} and {tuple(y.shape)}')
    if mix is not None:
        mix = mix.to(device=x.device, dtype=torch.float32).contiguous()
    out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(library().ppg_distance(
            x.device.index, x.data_ptr(), y.data_ptr(), x.shape[1],
            mix.data_ptr() if mix is not None else None, out.data_ptr(),
            torch.cuda.current_stream().cuda_stream))
    return out


def sparsify(ppg, method, threshold):
    """(batch, 40, frames) on a GPU -> same shape (ppg_sparsify); method 0
    constant / 1 percentile / 2 topk, one threshold."""
    if not ppg.is_cuda:
        raise PpgError('ppgs_amd: the post-ops work on HIP device tensors')
    x = ppg.to(torch.float32).contiguous()
    if x.dim() != 3 or x.shape[1] != 40:
        raise ValueError(f'PPG must be (batch, 40, frames), got {tuple(x.shape)}')
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(library().ppg_sparsify(
            x.device.index, x.data_ptr(), x.shape[0], x.shape[2], int(method),
            float(threshold), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out


def grid_sample(ppg, grid):
    """(..., frames) on a GPU sampled at the fractional frame indices `grid`
    (length,) -> (..., length) (ppg_grid_sample)."""
    if not ppg.is_cuda:
        raise PpgError('ppgs_amd: the post-ops work on HIP device tensors')
    if grid.dim() != 1:
        raise ValueError(f'grid must be one-dimensional, got {tuple(grid.shape)}')
    if ppg.dim() < 1 or ppg.shape[-1] < 1:
        raise ValueError(f'PPG must be (..., frames >= 1), got {tuple(ppg.shape)}')
    x = ppg.to(torch.float32).contiguous()
    g = grid.to(device=x.device, dtype=torch.float32).contiguous()
    out = torch.empty(x.shape[:-1] + (g.shape[0],), dtype=torch.float32, device=x.device)
    rows = x.numel() // x.shape[-1]
    if rows and g.shape[0]:
        with torch.cuda.device(x.device):
            _check(library().ppg_grid_sample(
                x.device.index, x.data_ptr(), rows, x.shape[-1], g.data_ptr(),
                g.shape[0], out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out      # fp32 also for half-precision PPGs: the reference's float grid promotes them


class PpgW2v2Weights(ctypes.Structure):
    _fields_ = [('conv_weight', _FP * 7), ('norm_weight', _FP), ('norm_bias', _FP)]


class Stream:
    """Streaming causal inference over ONE utterance (include/ppgs_amd.h: ppg_stream_*).

    The reference has no streaming mode: its causal configuration runs every chunk as an
    independent forward (ppgs/config/causal_transformer.py:18).  A Stream produces what the
    causal forward of the whole utterance (one window, <= 500 frames) produces, a chunk at
    a time, keeping K / V^T and the residual rows of everything seen on the device.  Both
    5-tap convolutions look two frames ahead, so push() returns the posteriors that became
    final: frames < received - 4 (all of them after flush=True)."""

    def __init__(self, engine, max_frames, dtype=torch.float16):
        if dtype not in (torch.float16, torch.float32):
            raise ValueError('stream features are fp16 or fp32')
        self.engine = engine
        self.dtype = dtype
        self._lib = engine._lib
        handle = ctypes.c_void_p()
        with torch.cuda.device(engine.device):
            _check(self._lib.ppg_stream_create(
                engine._handle, int(max_frames), 0 if dtype == torch.float16 else 1,
                ctypes.byref(handle)))
        self._handle = handle
        rows = ctypes.c_int()
        _check(self._lib.ppg_stream_rows(self._handle, ctypes.byref(rows), None, None))
        self.rows = rows.value
        self.max_frames = int(max_frames)

    def __del__(self):
        handle, self._handle = getattr(self, '_handle', None), None
        if handle:
            self._lib.ppg_stream_destroy(handle)

    @property
    def received(self):
        value = ctypes.c_int()
        _check(self._lib.ppg_stream_rows(self._handle, None, ctypes.byref(value), None))
        return value.value

    def push(self, chunk, flush=False, softmax=True):
        """chunk (input_channels, n) on the engine's GPU (n may be 0 with flush=True) ->
        (40, k) fp32: the posteriors (logits if softmax=False) of the k frames that became
        final with this chunk, in order."""
        engine = self.engine
        if chunk is None:
            chunk = torch.empty(engine.input_channels, 0, dtype=self.dtype, device=engine.device)
        if chunk.dim() != 2 or chunk.shape[0] != engine.input_channels:
            raise ValueError(f'chunk must be ({engine.input_channels}, frames), got {tuple(chunk.shape)}')
        chunk = chunk.to(device=engine.device, dtype=self.dtype).contiguous()
        first, count = ctypes.c_int(), ctypes.c_int()
        with torch.cuda.device(engine.device):
            stream = torch.cuda.current_stream().cuda_stream
            _check(self._lib.ppg_stream_push(
                self._handle, ctypes.c_void_p(chunk.data_ptr()), chunk.shape[1], int(flush),
                int(softmax), ctypes.byref(first), ctypes.byref(count),
                ctypes.c_void_p(stream)))
            out = torch.empty(engine.output_channels, count.value, dtype=torch.float32,
                              device=engine.device)
            if count.value:
                # the stream owns (output_channels, rows) fp32; copy the newly final columns out
                # (ordered behind the step on the current stream)
                address = self._lib.ppg_stream_posteriors(self._handle)
                source = _device_view(address, (engine.output_channels, self.rows), engine.device)
                out.copy_(source[:, first.value:first.value + count.value])
        return out


class BatchedStream:
    """KV-cached streaming causal inference for `batch` utterances advanced together (ppg_stream_create_batch /
    ppg_stream_push_batch): configs[4] of the reference -- causal_transformer, streaming chunks, batch = 64 -- with
    the state the reference does not keep.  Item b is an utterance of its own (one window, <= 500 frames) with its
    own frontier: a push hands every item its own number of new frames (ragged, not aligned to anything, 0 = the
    item sits this step out) and ONE launch sequence advances them all.  Each item equals the causal forward of
    its whole utterance, like :class:`Stream`."""

    def __init__(self, engine, batch, max_frames, dtype=torch.float16):
        if dtype not in (torch.float16, torch.float32):
            raise ValueError('stream features are fp16 or fp32')
        self.engine, self.dtype, self.batch = engine, dtype, int(batch)
        self._lib = engine._lib
        handle = ctypes.c_void_p()
        with torch.cuda.device(engine.device):
            _check(self._lib.ppg_stream_create_batch(
                engine._handle, self.batch, int(max_frames), 0 if dtype == torch.float16 else 1,
                ctypes.byref(handle)))
        self._handle = handle
        rows = ctypes.c_int()
        _check(self._lib.ppg_stream_rows(self._handle, ctypes.byref(rows), None, None))
        self.rows = rows.value
        self.max_frames = int(max_frames)
        self.received = [0] * self.batch
        address = self._lib.ppg_stream_posteriors(self._handle)
        self._posteriors = _device_view(
            address, (self.batch, engine.output_channels, self.rows), engine.device)

    def __del__(self):
        handle, self._handle = getattr(self, '_handle', None), None
        if handle:
            self._lib.ppg_stream_destroy(handle)

    def push(self, chunks, counts=None, flush=False, softmax=True):
        """chunks (batch, input_channels, n_max) on the engine's GPU (or None with only flushes), counts[b] <= n_max
        new frames of item b (default: n_max for every item), flush: bool or one bool per item -> a list of
        (40, k_b) fp32 tensors: the posteriors of item b's frames that became final with this step, in order."""
        engine = self.engine
        if chunks is None:
            chunks = torch.empty(self.batch, engine.input_channels, 0, dtype=self.dtype, device=engine.device)
        if chunks.dim() != 3 or chunks.shape[0] != self.batch or chunks.shape[1] != engine.input_channels:
            raise ValueError(f'chunks must be ({self.batch}, {engine.input_channels}, frames), got {tuple(chunks.shape)}')
        chunks = chunks.to(device=engine.device, dtype=self.dtype).contiguous()
        nmax = chunks.shape[2]
        counts = [nmax] * self.batch if counts is None else [int(n) for n in counts]
        flushes = [bool(flush)] * self.batch if isinstance(flush, bool) else [bool(f) for f in flush]
        if len(counts) != self.batch or len(flushes) != self.batch:
            raise ValueError('counts / flush need one entry per item')
        array = ctypes.c_int * self.batch
        first, final = array(), array()
        with torch.cuda.device(engine.device):
            stream = torch.cuda.current_stream().cuda_stream
            _check(self._lib.ppg_stream_push_batch(
                self._handle, ctypes.c_void_p(chunks.data_ptr()), nmax, array(*counts),
                array(*[int(f) for f in flushes]), int(softmax), first, final, ctypes.c_void_p(stream)))
            # ONE copy (ordered behind the step on the current stream; the stream object owns the source): the column
            # range all newly final frames lie in, of every item; the results are views of it
            spans = [(first[b], first[b] + final[b]) for b in range(self.batch) if final[b] > 0]
            lo = min((a for a, _ in spans), default=0)
            hi = max((b for _, b in spans), default=0)
            block = self._posteriors[:, :, lo:hi].clone()
            out = [block[b, :, first[b] - lo:first[b] - lo + final[b]] if final[b] > 0 else block[b, :, :0]
                   for b in range(self.batch)]
        for b in range(self.batch):
            self.received[b] += counts[b]
        return out


class LongStream:
    """Streaming causal inference over ONE utterance longer than a window.

    Reference Transformer.forward (ppgs/model/transformer.py:49-64) runs sequences of more than
    500 frames as independent windows of 500 rows every 400 frames over the left-replicate-padded
    sequence (row tt of window i = frame max(400 i + tt - 50, 0)), keeps rows [50, 450) of each and
    concatenates them.  A LongStream feeds every frame to the one or two windows it belongs to --
    each a KV-cached Stream -- and emits the kept rows in frame order: the causal forward of the
    whole utterance as the reference computes it for T > 500, incrementally, with the windows'
    own look-ahead of 4 frames.  (For T <= 500 the reference uses ONE unpadded window, which is
    what Stream reproduces; the regime cannot be switched once frames have been emitted.)"""

    def __init__(self, engine, dtype=torch.float16):
        self.engine, self.dtype = engine, dtype
        self.chunk, self.overlap = config.CHUNK_LENGTH, config.CHUNK_OVERLAP
        self.stride = self.chunk - 2 * self.overlap
        self.received = 0
        self._windows = {}          # index -> [Stream, rows emitted so far]
        self._done = -1             # windows <= this index have emitted all their kept rows (they finish in order)
        self._first = None

    def _window(self, index):
        if index not in self._windows:
            self._windows[index] = [Stream(self.engine, self.chunk, self.dtype), 0]
        return self._windows[index]

    def _emit(self, index, out, total=None):
        """rows of window `index` that became final -> the kept ones (as frames, in order)"""
        entry = self._windows[index]
        first = entry[1]
        entry[1] += out.shape[1]
        rows = torch.arange(first, entry[1], device=out.device)
        keep = (rows >= self.overlap) & (rows < self.chunk - self.overlap)
        if total is not None:
            keep &= (self.stride * index + rows - self.overlap) < total
        return out[:, keep]

    def push(self, chunk, flush=False, softmax=True):
        """chunk (input_channels, n) -> (40, k): the posteriors of the k frames that became final."""
        engine = self.engine
        if chunk is None:
            chunk = torch.empty(engine.input_channels, 0, dtype=self.dtype, device=engine.device)
        chunk = chunk.to(device=engine.device, dtype=self.dtype)
        n = chunk.shape[1]
        if self._first is None and n:
            self._first = chunk[:, :1]
        pieces = []
        lo, hi = self.received, self.received + n
        if n:
            first_window = max((lo + self.overlap - (self.chunk - 1) + self.stride - 1) // self.stride, 0)
            last_window = (hi - 1 + self.overlap) // self.stride
            for index in range(max(first_window, self._done + 1), last_window + 1):
                # (a finished window's last `overlap` rows keep arriving: nobody keeps them, and feeding them
                # would re-create its Stream -- workspace, K/V caches, a device synchronisation -- for nothing)
                # window rows tt = f - stride * index + overlap in [0, chunk)
                f0 = max(lo, self.stride * index - self.overlap)
                f1 = min(hi, self.stride * index - self.overlap + self.chunk)
                if f1 <= f0:
                    continue
                rows = chunk[:, f0 - lo:f1 - lo]
                if index == 0 and f0 == 0:        # the replicate padding on the left of the first window
                    rows = torch.cat([self._first.expand(-1, self.overlap), rows], dim=1)
                stream = self._window(index)[0]
                pieces.append(self._emit(index, stream.push(rows, softmax=softmax)))
        self.received = hi
        if flush:
            total = self.received
            for index in sorted(self._windows):
                stream = self._windows[index][0]
                if self.stride * index < total:   # a window whose kept rows start past the end does not exist
                    pieces.append(self._emit(index, stream.push(None, flush=True, softmax=softmax), total))
            self._windows.clear()
            self._done = (self.received + self.overlap - 1) // self.stride if self.received else -1
        else:
            # windows that have emitted all their kept rows are done
            for index in [i for i, entry in self._windows.items() if entry[1] >= self.chunk - self.overlap]:
                del self._windows[index]
                self._done = max(self._done, index)
        if not pieces:
            return torch.empty(engine.output_channels, 0, dtype=torch.float32, device=engine.device)
        return torch.cat(pieces, dim=1)


def _device_view(address, shape, device):
    """fp32 tensor view of device memory the library owns (no copy)."""
    import numpy as np

    class _Holder:
        pass
    holder = _Holder()
    count = int(np.prod(shape))
    holder.__cuda_array_interface__ = {
        'shape': (count,), 'typestr': '<f4', 'data': (int(address), False), 'version': 2}
    return torch.as_tensor(holder, device=device).view(*shape)


class W2v2FeatureEncoder:
    """The wav2vec 2.0 convolutional feature encoder on the HIP engine
    (ppg_w2v2_*): HF ``Wav2Vec2Model.feature_extractor`` -- seven strided
    convolutions, GroupNorm on the first, exact GELU -- from its state dict
    (keys ``conv_layers.{l}.conv.weight``, ``conv_layers.0.layer_norm.*``)."""

    def __init__(self, state, device=0, precision='fp16'):
        if not torch.cuda.is_available():
            raise PpgError(
                'ppgs_amd: no HIP device visible; the engine has no CPU path')
        lib = library()
        keep = []

        def ptr(key):
            tensor = state[key].detach().to('cpu', torch.float32).contiguous()
            keep.append(tensor)
            return ctypes.cast(tensor.data_ptr(), _FP)
        wts = PpgW2v2Weights()
        for layer in range(7):
            wts.conv_weight[layer] = ptr(f'conv_layers.{layer}.conv.weight')
        wts.norm_weight = ptr('conv_layers.0.layer_norm.weight')
        wts.norm_bias = ptr('conv_layers.0.layer_norm.bias')
        expected = [(512, 1, 10)] + [(512, 512, 3)] * 4 + [(512, 512, 2)] * 2
        shapes = [tuple(state[f'conv_layers.{i}.conv.weight'].shape) for i in range(7)]
        if shapes != expected:
            raise ValueError(f'not a wav2vec2-base feature encoder: conv weights {shapes}')
        handle = ctypes.c_void_p()
        _check(lib.ppg_w2v2_create(
            ctypes.byref(wts), PRECISIONS[precision], device, ctypes.byref(handle)))
        self._handle, self._lib = handle, lib
        self.device = torch.device('cuda', device)
        self.precision = precision
        self._workspaces = {}

    def __del__(self):
        handle = getattr(self, '_handle', None)
        if handle:
            self._lib.ppg_w2v2_destroy(handle)
            self._handle = None

    def frames(self, samples):
        return int(self._lib.ppg_w2v2_frames(int(samples)))

    def __call__(self, audio):
        """audio (batch, samples) fp32 on this GPU -> (batch, frames, 512) fp32
        (HF ``extract_features`` before the feature projection)."""
        if audio.dim() != 2:
            raise ValueError(f'audio must be (batch, samples), got {tuple(audio.shape)}')
        audio = audio.to(self.device, torch.float32).contiguous()
        batch, samples = audio.shape
        frames = self.frames(samples)
        if frames < 1:
            raise ValueError(f'{samples} samples are too few for the conv stack')
        size = ctypes.c_size_t()
        _check(self._lib.ppg_w2v2_workspace_bytes(self._handle, batch, samples, ctypes.byref(size)))
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            workspace = self._workspaces.get(stream)
            if workspace is None or workspace.numel() < size.value:
                workspace = torch.empty(size.value, dtype=torch.uint8, device=self.device)
                self._workspaces[stream] = workspace
            out = torch.empty((batch, frames, 512), dtype=torch.float32, device=self.device)
            _check(self._lib.ppg_w2v2_features(
                self._handle, audio.data_ptr(), batch, samples, out.data_ptr(),
                workspace.data_ptr(), workspace.numel(), stream))
        return out


def _weight_normed(conv, dim=2):
    """The effective weight of a weight-normalised convolution (HF's positional convolution:
    weight_norm(name='weight', dim=2)), computed from g and v -- with the legacy
    torch.nn.utils.weight_norm `.weight` is a plain attribute that only the forward pre-hook
    refreshes, so after from_pretrained it can be stale."""
    parametrizations = getattr(conv, 'parametrizations', None)
    if parametrizations is not None and 'weight' in parametrizations:
        g, v = parametrizations.weight.original0, parametrizations.weight.original1
    elif hasattr(conv, 'weight_g') and hasattr(conv, 'weight_v'):
        g, v = conv.weight_g, conv.weight_v
    else:
        return conv.weight
    return torch._weight_norm(v.detach().float(), g.detach().float(), dim)


class W2v2Body:
    """The wav2vec 2.0 transformer body on the HIP engine (ppg_w2v2_body_*): HF
    ``Wav2Vec2Model``'s ``feature_projection`` + ``encoder`` (post-norm layers, grouped
    positional convolution), built from the HF module's parameters."""

    def __init__(self, model, device=0, precision='fp16'):
        if not torch.cuda.is_available():
            raise PpgError('ppgs_amd: no HIP device visible; the engine has no CPU path')
        lib = library()
        cfg = model.config
        if getattr(cfg, 'do_stable_layer_norm', False):
            raise ValueError('the HIP wav2vec2 body implements the post-norm encoder (do_stable_layer_norm = False)')
        keep = []

        def ptr(tensor):
            tensor = tensor.detach().to('cpu', torch.float32).contiguous()
            keep.append(tensor)
            return ctypes.cast(tensor.data_ptr(), _FP)
        wts = PpgW2v2BodyWeights()
        wts.hidden, wts.heads, wts.ffn = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        wts.num_layers = len(model.encoder.layers)
        wts.conv_kernel, wts.conv_groups = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
        wts.layer_norm_eps = cfg.layer_norm_eps
        projection, encoder = model.feature_projection, model.encoder
        wts.proj_norm_weight, wts.proj_norm_bias = ptr(projection.layer_norm.weight), ptr(projection.layer_norm.bias)
        wts.proj_weight, wts.proj_bias = ptr(projection.projection.weight), ptr(projection.projection.bias)
        conv = encoder.pos_conv_embed.conv
        wts.pos_conv_weight, wts.pos_conv_bias = ptr(_weight_normed(conv)), ptr(conv.bias)
        wts.enc_norm_weight, wts.enc_norm_bias = ptr(encoder.layer_norm.weight), ptr(encoder.layer_norm.bias)
        for index, layer in enumerate(encoder.layers):
            lw, attn = wts.layers[index], layer.attention
            lw.q_weight, lw.q_bias = ptr(attn.q_proj.weight), ptr(attn.q_proj.bias)
            lw.k_weight, lw.k_bias = ptr(attn.k_proj.weight), ptr(attn.k_proj.bias)
            lw.v_weight, lw.v_bias = ptr(attn.v_proj.weight), ptr(attn.v_proj.bias)
            lw.out_weight, lw.out_bias = ptr(attn.out_proj.weight), ptr(attn.out_proj.bias)
            lw.norm1_weight, lw.norm1_bias = ptr(layer.layer_norm.weight), ptr(layer.layer_norm.bias)
            lw.ffn1_weight = ptr(layer.feed_forward.intermediate_dense.weight)
            lw.ffn1_bias = ptr(layer.feed_forward.intermediate_dense.bias)
            lw.ffn2_weight = ptr(layer.feed_forward.output_dense.weight)
            lw.ffn2_bias = ptr(layer.feed_forward.output_dense.bias)
            lw.norm2_weight, lw.norm2_bias = ptr(layer.final_layer_norm.weight), ptr(layer.final_layer_norm.bias)
        handle = ctypes.c_void_p()
        _check(lib.ppg_w2v2_body_create(
            ctypes.byref(wts), PRECISIONS[precision], device, ctypes.byref(handle)))
        self._handle, self._lib = handle, lib
        self.device = torch.device('cuda', device)
        self.precision = precision
        self.hidden = cfg.hidden_size
        self._workspaces = {}

    def __del__(self):
        handle = getattr(self, '_handle', None)
        if handle:
            self._lib.ppg_w2v2_body_destroy(handle)
            self._handle = None

    def __call__(self, features, valid_frames):
        """features (batch, frames, 512) fp32 on this GPU (HF ``extract_features``), valid frames
        per item (HF's frame-level attention mask as lengths) -> ``last_hidden_state``
        (batch, frames, hidden) fp32."""
        if features.dim() != 3 or features.shape[2] != 512:
            raise ValueError(f'features must be (batch, frames, 512), got {tuple(features.shape)}')
        features = features.to(self.device, torch.float32).contiguous()
        batch, frames, _ = features.shape
        arr = _lengths_array(valid_frames, batch)
        size = ctypes.c_size_t()
        _check(self._lib.ppg_w2v2_body_workspace_bytes(self._handle, batch, frames, ctypes.byref(size)))
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            workspace = self._workspaces.get(stream)
            if workspace is None or workspace.numel() < size.value:
                workspace = torch.empty(size.value, dtype=torch.uint8, device=self.device)
                self._workspaces[stream] = workspace
            out = torch.empty((batch, frames, self.hidden), dtype=torch.float32, device=self.device)
            _check(self._lib.ppg_w2v2_body_forward(
                self._handle, features.data_ptr(), arr, batch, frames, out.data_ptr(),
                workspace.data_ptr(), workspace.numel(), stream))
        return out


def frontend_profile(device, enable=True):
    _check(library().ppg_frontend_profile(device, int(enable)))


def frontend_profile_read(device):
    total, count = ctypes.c_double(), ctypes.c_int64()
    _check(library().ppg_frontend_profile_read(
        device, ctypes.byref(total), ctypes.byref(count)))
    return total.value, count.value


###############################################################################
# Native file ingest / output (host only)
###############################################################################


def _paths_array(paths):
    encoded = [os.fsencode(os.fspath(p)) for p in paths]
    return (ctypes.c_char_p * len(encoded))(*encoded)


def _check_io(code):
    if code < 0:
        raise ValueError('ppgs_amd: ' + library().ppg_io_last_error().decode())


def wav_info(path):
    """(samples per channel, sample rate, channels) from the RIFF header."""
    samples, rate, channels = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
    _check_io(library().ppg_wav_info(
        os.fsencode(os.fspath(path)), ctypes.byref(samples), ctypes.byref(rate),
        ctypes.byref(channels)))
    return samples.value, rate.value, channels.value


def wav_read_batch(paths, max_samples, threads=4, pin_memory=None):
    """Decode WAV files into one zero-padded (B, 1, max_samples) fp32 tensor
    (pinned when a GPU is present) -> (tensor, samples (B,), rates list)."""
    count = len(paths)
    if pin_memory is None:
        pin_memory = torch.cuda.is_available()
    batch = torch.empty((count, 1, max_samples), dtype=torch.float32,
                        pin_memory=pin_memory)
    samples = (ctypes.c_int64 * count)()
    rates = (ctypes.c_int32 * count)()
    _check_io(library().ppg_wav_read_batch(
        _paths_array(paths), count, batch.data_ptr(), max_samples, max_samples,
        samples, rates, int(threads)))
    return batch, torch.tensor(list(samples), dtype=torch.long), list(rates)


def pt_write_batch(paths, tensor, lengths, threads=4):
    """Write tensor[i, :, :lengths[i]] (fp32, CPU, (B, rows, T) contiguous) as
    torch.load-able .pt files."""
    if tensor.is_cuda or tensor.dtype != torch.float32 or tensor.dim() != 3:
        raise ValueError('pt_write_batch expects a CPU fp32 (B, rows, T) tensor')
    tensor = tensor.contiguous()
    count, rows, frames = tensor.shape
    cols = (ctypes.c_int64 * count)(*[int(v) for v in lengths])
    _check_io(library().ppg_pt_write_batch(
        _paths_array(paths), count, tensor.data_ptr(), rows * frames, rows,
        frames, cols, int(threads)))
