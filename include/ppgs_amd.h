/*
 * ppgs_amd.h -- C ABI of the MI355X-native PPG inference engine.
 *
 * The reference (interactiveaudiolab/ppgs) has no FFI layer: its boundary is
 * Python.  This header is the C-ABI seam a drop-in sits
 * behind; each entry point names the reference function it replaces
 * (paths relative to the reference checkout).  All device pointers are raw
 * HIP device addresses on the engine's device, `stream` is a hipStream_t
 * passed as void*.  Every function returns 0 on success or a negative
 * PPG_E* code; ppg_last_error() returns a thread-local message.
 * Nothing here computes on the CPU: without a HIP device the compute entry
 * points fail with PPG_EDEVICE.
 */
#ifndef PPGS_AMD_H
#define PPGS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPG_ABI_VERSION 1
#define PPG_MAX_LAYERS 16

enum {
    PPG_OK = 0,
    PPG_EINVAL = -1,     /* bad argument (shape, length, null pointer)        */
    PPG_EDEVICE = -2,    /* HIP runtime error / no device                     */
    PPG_EWORKSPACE = -3, /* workspace too small                               */
    PPG_ELENGTH = -4,    /* legacy_mode length limit, PE table overflow
                            (reference ppgs/model/transformer.py:46-48,103)    */
};

/* Arithmetic the encoder GEMMs run in. */
enum {
    PPG_PRECISION_FP32 = 0, /* f32-input MFMA, parity mode (<=1e-4 vs oracle)  */
    PPG_PRECISION_BF16 = 1, /* bf16 MFMA, fp32 accumulate, throughput mode     */
    PPG_PRECISION_FP16 = 2, /* fp16 MFMA operands (11-bit significand), fp32 accumulate:
                               same MFMA rate as bf16, ~8x smaller operand rounding;
                               operands must stay below 65504 in magnitude (what the
                               reference's CUDA autocast assumes, ppgs/core.py:586)  */
    PPG_PRECISION_FP16X2 = 3, /* fp32 values as two fp16 halves (hi + lo), a product as three fp16 MFMAs with
                               fp32 accumulation: fp32-grade operands (22 significand bits; <= 1e-4 vs the
                               reference's fp32 forward -- the autocast-off route of ppgs/core.py:586-594)
                               at a third of the fp16 MFMA rate, 5x the f32-input MFMA rate.  Hidden 256 with
                               head dimension 128 (mel-sized models) and hidden 512 with head dimension 256
                               (the w2v2fb network; no KV-cached stream there); the wav2vec2 engines
                               (ppg_w2v2_create / ppg_w2v2_body_create) take it too; magnitudes below 65504 as fp16   */
};

/* dtype tags for feature tensors handed to ppg_encode */
enum { PPG_DTYPE_F16 = 0, PPG_DTYPE_F32 = 1 };

/* Kernel classes for ppg_engine_profile_read */
enum {
    PPG_K_GATHER = 0,
    PPG_K_INCONV = 1,
    PPG_K_QKV = 2,
    PPG_K_ATTENTION = 3,
    PPG_K_OUTPROJ_LN = 4,
    PPG_K_FFN = 5,
    PPG_K_OUTCONV_SOFTMAX = 6,
    PPG_K_FRONTEND = 7,
    PPG_K_COUNT = 8,
};

/*
 * Model geometry: constructor arguments of the reference network,
 * ppgs/model/transformer.py:15-43 (+ torch TransformerEncoderLayer defaults:
 * ffn 2048, ReLU, post-norm, eps 1e-5) and the chunking constants
 * ppgs/config/defaults.py:158-161.
 */
typedef struct PpgConfig {
    int32_t input_channels;  /* 80 (mel) / 768 (w2v2fb)                       */
    int32_t hidden_channels; /* 256 / 512; multiple of 64                     */
    int32_t num_layers;      /* 5                                             */
    int32_t ffn_channels;    /* 2048                                          */
    int32_t output_channels; /* 40                                            */
    int32_t kernel_size;     /* 5                                             */
    int32_t heads;           /* 2; hidden/heads must be 128 or 256            */
    int32_t is_causal;       /* config/causal_transformer.py:18               */
    int32_t max_positions;   /* 5000 rows in position.encoding                */
    int32_t chunk_length;    /* 500                                           */
    int32_t chunk_overlap;   /* 50                                            */
    int32_t precision;       /* PPG_PRECISION_*                               */
} PpgConfig;

/*
 * Host pointers to fp32 arrays in the reference checkpoint layout
 * (state_dict of ppgs.model.Transformer, ppgs/load.py:74-79; key list in
 * SURVEY.md 8(b)).  Copied to the device (and re-packed) by
 * ppg_engine_create; the caller may free them afterwards.
 */
typedef struct PpgWeights {
    const float* position_encoding;           /* (max_positions, H)           */
    const float* input_weight;                /* (H, Cin, 5)                  */
    const float* input_bias;                  /* (H)                          */
    const float* in_proj_weight[PPG_MAX_LAYERS];  /* (3H, H) rows [q;k;v]     */
    const float* in_proj_bias[PPG_MAX_LAYERS];    /* (3H)                     */
    const float* out_proj_weight[PPG_MAX_LAYERS]; /* (H, H)                   */
    const float* out_proj_bias[PPG_MAX_LAYERS];   /* (H)                      */
    const float* linear1_weight[PPG_MAX_LAYERS];  /* (F, H)                   */
    const float* linear1_bias[PPG_MAX_LAYERS];    /* (F)                      */
    const float* linear2_weight[PPG_MAX_LAYERS];  /* (H, F)                   */
    const float* linear2_bias[PPG_MAX_LAYERS];    /* (H)                      */
    const float* norm1_weight[PPG_MAX_LAYERS];    /* (H)                      */
    const float* norm1_bias[PPG_MAX_LAYERS];
    const float* norm2_weight[PPG_MAX_LAYERS];
    const float* norm2_bias[PPG_MAX_LAYERS];
    const float* output_weight;               /* (40, H, 5)                   */
    const float* output_bias;                 /* (40)                         */
} PpgWeights;

/*
 * One window of the chunk plan: an independent <=chunk_length-frame forward
 * of reference Transformer.forward's recursion (transformer.py:49-64).
 */
typedef struct PpgWindow {
    int32_t item;     /* batch row                                            */
    int32_t chunked;  /* 1: source frame of window column t is
                            max(start + t - overlap, 0) (left replicate pad);
                         0: source frame = t                                   */
    int32_t start;    /* window start in the replicate-padded sequence         */
    int32_t frames;   /* Tc, window length (<= chunk_length)                   */
    int32_t valid;    /* per-item valid frames inside the window (mask length) */
    int32_t keep_lo;  /* window columns [keep_lo, keep_hi) go to the output    */
    int32_t keep_hi;
    int32_t out_frame;/* output frame of window column keep_lo                 */
    int32_t tok_off;  /* first row of the window in the token-major buffers    */
    int32_t vt_off;   /* first column of the window in the transposed-V buffer */
    int32_t pad0, pad1;
} PpgWindow;

typedef struct PpgPlanInfo {
    int32_t num_windows;     /* windows that are computed (valid > 0)          */
    int32_t skipped_windows; /* windows whose item is exhausted (all-masked)   */
    int32_t tokens;          /* rows of the token-major buffers (padded)       */
    int32_t vt_tokens;       /* columns of the transposed-V buffer (padded)    */
    int64_t processed_frames;/* sum of window lengths (unpadded)               */
    int64_t attention_pairs; /* sum of Tc^2 over computed windows              */
    size_t workspace_bytes;  /* what ppg_encode needs                          */
} PpgPlanInfo;

typedef struct PpgEngine PpgEngine;

const char* ppg_last_error(void);
int ppg_abi_version(void);

/*
 * Engine = one loaded model on one device; replaces the per-(representation,
 * checkpoint) cache of ppgs.infer (ppgs/core.py:565-583) and ppgs.load.model
 * (ppgs/load.py:33-81, the state_dict -> module step).
 */
int ppg_engine_create(const PpgConfig* config, const PpgWeights* weights,
                      int device, PpgEngine** engine);
void ppg_engine_destroy(PpgEngine* engine);

/*
 * Chunk planner, host only (usable without a device): the window list of
 * reference Transformer.forward (transformer.py:49-64) for a (B, C, T) batch
 * with per-item lengths.  `legacy_mode` = one window over the whole sequence
 * (transformer.py:46-48).  Writes up to max_windows entries (all windows,
 * skipped ones included with tok_off = -1) and returns the total count, or a
 * negative error.  `engine` may be NULL: then chunk 500 / overlap 50.
 */
int ppg_plan_windows(const PpgEngine* engine, int batch, int frames,
                     const int64_t* lengths_host, int legacy_mode,
                     PpgWindow* windows, int max_windows, PpgPlanInfo* info);

/*
 * The attention launch order of a batch, host only: one item per (window,
 * query tile), in the order the workgroups are dispatched -- longest windows
 * first, the tiles of one window `8 / gcd(8, heads)` items apart so that all
 * of them (for one head) run on one XCD's L2, half-width tiles for the short
 * windows of a batch (ppg_engine.hip: split_groups).  `engine` may be NULL:
 * then chunk 500 / overlap 50, head dimension 128, `heads` as given, one
 * launch group.  Writes up to max_items entries and returns the total count,
 * or a negative error.
 */
typedef struct PpgAttentionItem {
    int32_t window;          /* index into the COMPUTED windows (tok_off >= 0) in plan order */
    int32_t q0;              /* first query row of the tile, window-relative                   */
    int32_t queries;         /* tile width: 128 or 64 (head dimension 128), 64 (256)           */
    int32_t frames;          /* rows of the window                                             */
    int32_t valid;           /* keys that count (the rest is padding)                          */
    int32_t narrow;          /* 1: half-width tile                                             */
} PpgAttentionItem;
int ppg_plan_attention_items(const PpgEngine* engine, int batch, int frames,
                             const int64_t* lengths_host, int legacy_mode, int heads,
                             PpgAttentionItem* items, int max_items);

/* Workspace (device bytes) ppg_encode needs for this batch. */
int ppg_workspace_bytes(const PpgEngine* engine, int batch, int frames,
                        const int64_t* lengths_host, int legacy_mode,
                        size_t* bytes);

/*
 * The forward pass: replaces ppgs.from_features -> ppgs.infer ->
 * Transformer.forward -> softmax(dim=1) (ppgs/core.py:72-128, 551-596;
 * ppgs/model/transformer.py:45-81).
 *   features : device, (batch, input_channels, frames), fp16 or fp32
 *   lengths  : HOST int64[batch] valid frames per row (1 <= len <= frames)
 *   out      : device fp32 (batch, output_channels, frames); posteriors if
 *              softmax != 0, else logits.  Frames >= length hold the
 *              reference's values there (zero logits -> uniform 1/40).
 *   workspace: device scratch of >= ppg_workspace_bytes, 256-B aligned
 */
int ppg_encode(PpgEngine* engine, const void* features, int feature_dtype,
               const int64_t* lengths_host, int batch, int frames,
               int softmax, int legacy_mode, float* out,
               void* workspace, size_t workspace_bytes, void* stream);

/*
 * Frontend: replaces ppgs.preprocess.spectrogram.from_audios
 * (ppgs/preprocess/spectrogram.py:14-50) and ppgs.preprocess.mel.from_audios
 * (ppgs/preprocess/mel.py:14-19, 56-76).
 *   audio: device fp32 (batch, samples) rows already zero-extended to the
 *          batch's sample count (reference ppgs/data/collate.py:20-27)
 *   spec : device fp16 (batch, 513, samples/160) or NULL
 *   mel  : device fp16 (batch, 80, samples/160) or NULL
 * samples must be > 432 (reflect padding) -- same limit as torch's
 * reflection pad in the reference; batch * 513 * (samples / 160) must stay
 * below 2^32 (the kernel indexes its outputs with 32 bits; PPG_EINVAL beyond).
 */
int ppg_frontend(int device, const float* audio, int batch, int samples,
                 void* spec, void* mel, void* stream);

/*
 * Sample-rate conversion on the device: replaces ppgs.resample
 * (ppgs/core.py:599-608 = torchaudio.transforms.Resample with its defaults:
 * Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99), which the
 * reference applies to loaded audio that is not at 16 kHz (ppgs/load.py:29-30).
 *   audio: device fp32 (batch, samples) at orig_rate
 *   out  : device fp32 (batch, ppg_resample_length(samples, orig_rate, new_rate))
 */
int64_t ppg_resample_length(int64_t samples, int orig_rate, int new_rate);
int ppg_resample(int device, const float* audio, int batch, int64_t samples,
                 int orig_rate, int new_rate, float* out, void* stream);

/*
 * wav2vec 2.0 transformer body on the HIP engine: what HF
 * `Wav2Vec2Model.forward` runs after the convolutional feature encoder --
 * feature_projection (LayerNorm 512 + Linear 512 -> hidden), the encoder's
 * grouped positional convolution (k = 128, 16 groups, GELU) + residual +
 * LayerNorm, and `num_layers` post-norm layers (self-attention with a key
 * padding mask, FFN with exact GELU) -- transformers/models/wav2vec2/
 * modeling_wav2vec2.py: Wav2Vec2FeatureProjection, Wav2Vec2PositionalConvEmbedding,
 * Wav2Vec2Encoder (do_stable_layer_norm = False), Wav2Vec2EncoderLayer.
 * Weights are host fp32 in torch layouts; `pos_conv_weight` is the EFFECTIVE
 * convolution weight (weight norm applied): (hidden, hidden / groups, kernel).
 *   features     : device fp32 (batch, frames, 512) = ppg_w2v2_features' output
 *   valid_frames : HOST int64[batch]: frames of each item that are real (HF's
 *                  frame-level attention mask); rows past it are zeroed after
 *                  the projection and masked as attention keys
 *   out          : device fp32 (batch, frames, hidden) = last_hidden_state
 */
#define PPG_W2V2_MAX_LAYERS 24
typedef struct PpgW2v2LayerWeights {
    const float* q_weight; const float* q_bias;            /* (hidden, hidden), (hidden)  */
    const float* k_weight; const float* k_bias;
    const float* v_weight; const float* v_bias;
    const float* out_weight; const float* out_bias;
    const float* norm1_weight; const float* norm1_bias;    /* layer_norm                  */
    const float* ffn1_weight; const float* ffn1_bias;      /* intermediate_dense (ffn, hidden) */
    const float* ffn2_weight; const float* ffn2_bias;      /* output_dense (hidden, ffn)  */
    const float* norm2_weight; const float* norm2_bias;    /* final_layer_norm            */
} PpgW2v2LayerWeights;
typedef struct PpgW2v2BodyWeights {
    int32_t hidden;          /* 768  */
    int32_t heads;           /* 12 (hidden / heads = 64) */
    int32_t ffn;             /* 3072 */
    int32_t num_layers;
    int32_t conv_kernel;     /* 128  */
    int32_t conv_groups;     /* 16   */
    float layer_norm_eps;    /* 1e-5 */
    int32_t pad0;
    const float* proj_norm_weight; const float* proj_norm_bias;   /* (512)                */
    const float* proj_weight; const float* proj_bias;             /* (hidden, 512)        */
    const float* pos_conv_weight; const float* pos_conv_bias;     /* see above            */
    const float* enc_norm_weight; const float* enc_norm_bias;     /* (hidden)             */
    PpgW2v2LayerWeights layers[PPG_W2V2_MAX_LAYERS];
} PpgW2v2BodyWeights;
typedef struct PpgW2v2Body PpgW2v2Body;
int ppg_w2v2_body_create(const PpgW2v2BodyWeights* weights, int precision, int device, PpgW2v2Body** out);
void ppg_w2v2_body_destroy(PpgW2v2Body* body);
int ppg_w2v2_body_workspace_bytes(const PpgW2v2Body* body, int batch, int frames, size_t* bytes);
int ppg_w2v2_body_forward(PpgW2v2Body* body, const float* features, const int64_t* valid_frames_host,
                          int batch, int frames, float* out, void* workspace, size_t workspace_bytes,
                          void* hip_stream);

/*
 * Streaming causal mode.  The reference has no streaming
 * state (ppgs/config/causal_transformer.py:18 only switches the causal mask on;
 * every call is an independent forward), so the contract is defined here: a
 * stream reproduces the causal forward of ONE utterance of up to `max_frames`
 * (<= the chunk length: one window) frames, emitted incrementally while the
 * K / V^T rows and the residual rows of everything seen stay on the device.
 * Both 5-tap convolutions look 2 frames ahead, so after F frames the posteriors
 * of frames < F - 4 are final; `flush` (end of utterance) finalises the rest.
 * ppg_stream_push copies `n` new frames ((input_channels, n), device, the dtype
 * given at creation) and computes what became final; the posteriors live in
 * the stream's own buffer ppg_stream_posteriors(): (output_channels, rows)
 * fp32, `rows` from ppg_stream_rows, column t = frame t; columns
 * [*first_final, *first_final + *num_final) are the newly final ones.
 * The engine must have been created with is_causal = 1.
 */
typedef struct PpgStream PpgStream;
int ppg_stream_create(PpgEngine* engine, int max_frames, int feature_dtype, PpgStream** stream);
void ppg_stream_destroy(PpgStream* stream);
int ppg_stream_rows(const PpgStream* stream, int* rows, int* received_frames, int* final_frames);
const float* ppg_stream_posteriors(const PpgStream* stream);
int ppg_stream_push(PpgStream* stream, const void* chunk_device, int n_frames, int flush, int softmax,
                    int* first_final, int* num_final, void* hip_stream);
/*
 * The same for `batch` utterances advanced together (configs[4] is "streaming chunks, batch = 64",
 * ppgs/config/causal_transformer.py:18): one stream object, item b an utterance of its own with its own frontier.
 * ppg_stream_push_batch takes chunk (batch, input_channels, n_max) on the device and, per item, how many of its
 * n_max columns are new frames (counts_host[b] in [0, n_max], ragged; 0 and no flush = the item sits this step out)
 * and whether the item ends (flush_host, may be null); ONE launch sequence advances all items.  Posteriors:
 * (batch, output_channels, rows) fp32 at ppg_stream_posteriors(); first_final / num_final are per item.
 * An item equals the causal forward of its own utterance (as the one-utterance stream does).
 */
int ppg_stream_create_batch(PpgEngine* engine, int batch, int max_frames, int feature_dtype, PpgStream** stream);
int ppg_stream_batch(const PpgStream* stream);
int ppg_stream_push_batch(PpgStream* stream, const void* chunk_device, int n_max, const int* counts_host,
                          const int* flush_host, int softmax, int* first_final, int* num_final, void* hip_stream);

/*
 * wav2vec 2.0 feature encoder of the 'w2v2fb' representation (reference
 * ppgs/preprocess/w2v2fb/core.py:66 calls HF transformers
 * Wav2Vec2Model.feature_extractor -- Wav2Vec2FeatureEncoder of
 * models/wav2vec2/modeling_wav2vec2.py: Conv1d(1,512,k10,s5) + GroupNorm(512,512)
 * + GELU, then six Conv1d(512,512,k{3,3,3,3,2,2},s2) + GELU, no biases).
 *   weights : host fp32 arrays in torch layout: conv_weight[l] (512, Cin, k),
 *             norm_weight / norm_bias (512) of layer 0's GroupNorm
 *   audio   : device fp32 (batch, samples), already padded as the caller wants
 *   out     : device fp32 (batch, ppg_w2v2_frames(samples), 512) = HF's
 *             extract_features before the feature projection
 * The layers run as MFMA GEMMs in the precision given at creation (any PPG_PRECISION_*; FP16X2: layers 1..6 on fp16
 * hi + lo operand pairs, <= 1e-4 against HF's fp32 output).
 */
typedef struct PpgW2v2Weights {
    const float* conv_weight[7];
    const float* norm_weight;
    const float* norm_bias;
} PpgW2v2Weights;
typedef struct PpgW2v2 PpgW2v2;
int ppg_w2v2_create(const PpgW2v2Weights* weights, int precision, int device, PpgW2v2** out);
void ppg_w2v2_destroy(PpgW2v2* model);
int64_t ppg_w2v2_frames(int64_t samples);
int ppg_w2v2_workspace_bytes(const PpgW2v2* model, int batch, int64_t samples, size_t* bytes);
int ppg_w2v2_features(PpgW2v2* model, const float* audio, int batch, int64_t samples, float* out,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * PPG post-ops on the device (per-frame arithmetic over the 40 phonemes).
 *
 * ppg_distance: replaces the body of ppgs.distance (ppgs/core.py:399-472).
 *   ppg_x, ppg_y: device fp32 (40, frames);  mix: device fp32 (40, 40) =
 *   similarity.T ** exponent (the reference's normalize=True), or NULL
 *   (normalize=False);  jsd: device fp32 (frames) -- the reference's
 *   reduction='none' result; 'mean' / 'sum' are one reduction over it.
 * ppg_sparsify: replaces ppgs.sparsify (ppgs/core.py:510-543) for one
 *   threshold.  ppg, out: device fp32 (batch, 40, frames);  method 0 =
 *   'constant', 1 = 'percentile' (threshold = quantile in [0, 1]), 2 = 'topk'
 *   (threshold = k).
 * ppg_grid_sample: replaces ppgs.edit.grid.sample (ppgs/edit/grid.py:13-45),
 *   time-stretching by fractional frame indices.  ppg: device fp32 (rows,
 *   frames), rows = every leading dimension flattened;  grid: device fp32
 *   (length) indices into the frames;  out: device fp32 (rows, length) =
 *   linear interpolation between the two neighbouring frames, the last frame
 *   repeated once past the end.
 */
int ppg_distance(int device, const float* ppg_x, const float* ppg_y, int frames,
                 const float* mix, float* jsd, void* stream);
int ppg_sparsify(int device, const float* ppg, int batch, int frames, int method,
                 float threshold, float* out, void* stream);
int ppg_grid_sample(int device, const float* ppg, int rows, int frames,
                    const float* grid, int length, float* out, void* stream);

/*
 * Per-kernel-class timing with HIP events on the launch stream (used by
 * bench.py's roofline leg).  `classes` is a bitmask of (1 << PPG_K_*), 0 =
 * off, -1 = every class (each timed launch costs two event records on the
 * stream).  Enable, run, then read: total milliseconds and launch count
 * accumulated since the last reset.  Reading synchronises the recorded events.
 * ppg_engine_profile_stride(n): time only every n-th launch of each enabled
 * class (n = 1: all); `launches` then counts the timed ones.
 */
/*
 * Sticky non-finite flag.  The 16-bit operand modes assume activations inside the operand format's range (fp16:
 * |x| < 65504 -- what the reference's own CUDA autocast assumes of its checkpoints); an overflow upstream (or
 * non-finite input features) ends as NaN logits.  The output kernels set a device flag whenever a VALID frame's logit
 * is not finite; ppg_engine_nonfinite copies it to *flag (synchronising with the device) and, with clear != 0,
 * resets it.  The Python layer raises on it wherever it synchronises anyway (file pipelines, from_audio with
 * PPGS_AMD_CHECK_FINITE=1) so that NaN posteriors never leave silently.
 */
int ppg_engine_nonfinite(PpgEngine* engine, int clear, int* flag);
/*
 * Pipelines a batch whose window plan has `tokens` token rows (PpgPlanInfo.tokens) is split into: the windows of a
 * batch are independent, so a batch of at least 128 rows per CU runs as two half-batches on two HIP streams of the
 * engine (forked from and joined into the caller's stream; the results equal one pipeline's: bit for bit for a uniform batch in the
 * 16-bit modes, within the operand format's rounding otherwise -- the planner ranks a pipeline's windows for the attention tile
 * width, the fp32 mode's kernels sum hidden chunks from a tile-dependent start).
 * PPGS_AMD_STREAMS=1 at engine creation turns the split off.  No reference counterpart (one CUDA stream).
 */
int ppg_engine_pipelines(const PpgEngine* engine, int tokens);
int ppg_engine_profile(PpgEngine* engine, int classes);
int ppg_engine_profile_read(PpgEngine* engine, int kernel_class,
                            double* total_ms, int64_t* launches);
int ppg_engine_profile_reset(PpgEngine* engine);
int ppg_engine_profile_stride(PpgEngine* engine, int stride);
int ppg_frontend_profile(int device, int enable);
int ppg_frontend_profile_read(int device, double* total_ms, int64_t* launches);

/*
 * Host-side ingest / output stage of the file pipeline (no GPU involved).
 *
 * ppg_wav_read_batch: decode `count` RIFF/WAVE files (PCM 8/16/24/32-bit,
 * IEEE float 32/64, first channel) into rows of one fp32 batch buffer
 * dst[count][row_stride], zero-padded behind each file's samples, `threads`
 * at a time -- replaces torchaudio.load (ppgs/load.py:17-30) + the zero-pad
 * Collate (ppgs/data/collate.py:20-27) of the reference's DataLoader workers.
 * samples_out / rates_out receive each file's sample count and sample rate
 * (files not at 16 kHz must be resampled by the caller).
 *
 * ppg_pt_write_batch: write item i as a torch.load-able ".pt" holding the
 * contiguous fp32 tensor src[i][0:rows][0:cols[i]] -- replaces the spawn Pool
 * of save_masked / torch.save workers (ppgs/preprocess/core.py:219-221,
 * ppgs/core.py:358-378).
 */
const char* ppg_io_last_error(void);
int ppg_wav_info(const char* path, int64_t* samples, int32_t* sample_rate,
                 int32_t* channels);
int ppg_wav_read_batch(const char* const* paths, int count, float* dst,
                       int64_t row_stride, int64_t max_samples,
                       int64_t* samples_out, int32_t* rates_out, int threads);
int ppg_pt_write_batch(const char* const* paths, int count, const float* src,
                       int64_t item_stride, int rows, int64_t row_stride,
                       const int64_t* cols, int threads);

#ifdef __cplusplus
}
#endif
#endif /* PPGS_AMD_H */
