/* mi355asr.h -- C ABI of libmi355asr.so: the MI355X (gfx950) Conformer-CTC hot path of
 * Z-yq/TensorflowASR (encoder + CTCDecoder + CTC greedy decode) behind plain pointers and sizes.
 *
 * The reference has no FFI for this path: it is Keras `Model` objects called from Python
 * (test_asr.py:186-219) and, in deployment, ONNX sessions called from C++
 * (Inference/CppInference/onnx/src/core/asr_session.cpp:77-123).  Each entry point below names the
 * reference interface it stands in for.  INTEGRATION.md shows the ctypes stub (Python reference)
 * and the C++ `ASR::Session` patch a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MI355ASR_E* code on failure;
 *     mi355asr_last_error() returns a thread-local message for the last failure on this thread
 *   - `*_dev` pointers are DEVICE pointers owned by the caller (e.g. torch-ROCm tensors);
 *     host pointers are marked `_host`
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is enqueued
 *     asynchronously on it; nothing synchronises the device
 *   - a handle is not re-entrant: one in-flight call per handle (the reference's C++ Session has the
 *     same contract, asr_session.h keeps mutable buffers); distinct handles are independent
 *   - values and accumulation are fp32 everywhere, matching the reference's dtype.  Products are formed by the fp32 matrix
 *     instruction (v_mfma_f32_16x16x4_f32, an exact fp32 FMA chain) or, in the large dmodel-144 kernels, on the 16-bit
 *     matrix pipe from SPLIT fp32 operands: three bf16 terms each, exact (six v_mfma_f32_16x16x32_bf16 per product group,
 *     the dropped term pairs below 2^-24 of a product), or -- where an operand bound is known: the handle's own weights,
 *     its LayerNorm outputs, its frontend's dB range, a row maximum measured in the kernel -- two fp16 terms of the operand
 *     times a power of two (hi + lo, round-to-nearest: 2^-22 relative; three v_mfma_f32_16x16x32_f16 per product group).
 *     Both sit at the same measured distance from the fp64 oracle as the fp32 instruction; stage
 *     calls on caller-supplied tensors always take the exact three-term or fp32 kernels.
 *     gemm_dtype = 1 is the separate, lossy bf16 mode (operands rounded to bf16).
 */
#ifndef MI355ASR_H
#define MI355ASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355ASR_OK 0
#define MI355ASR_EINVAL -1   /* bad argument / unsupported configuration */
#define MI355ASR_ESTATE -2   /* call order (weights not finalised, ...) */
#define MI355ASR_EWEIGHT -3  /* unknown / missing / mis-shaped weight */
#define MI355ASR_EWORKSPACE -4
#define MI355ASR_EHIP -5     /* HIP runtime error */

typedef struct mi355asr_model mi355asr_model;

/* Constructor arguments of ConformerEncoder / StreamingConformerEncoder / CTCDecoder
 * (asr/models/conformer_blocks.py:278-294, 386-395, 568-572; values from the YAML files under asr/configs/). */
typedef struct {
  int32_t dmodel;            /* model_config.dmodel                       144 | 256            */
  int32_t num_blocks;        /* model_config.num_blocks                   13  | 4              */
  int32_t head_size;         /* model_config.head_size                    36  | 64   (tuned); 12, 16, 24, 32, 48, 72, 128 run on a general kernel */
  int32_t num_heads;         /* model_config.num_heads                    4                    */
  int32_t kernel_size;       /* model_config.kernel_size                  32  | 5    (tuned); any other size 1 .. 1024 runs on a general kernel   */
  float   fc_factor;         /* model_config.fc_factor                    0.5                  */
  int32_t reduction_factor;  /* model_config.reduction_factor             4          (2, 4, 6 or 8; the tuned kernels are the ones for 4)          */
  int32_t n_mels;            /* speech_config.num_feature_bins            80                   */
  int32_t sample_rate;       /* speech_config.sample_rate                 16000                */
  int32_t stride_ms;         /* speech_config.stride_ms                   10                   */
  int32_t n_dft;             /* hard-coded 1024 in conformer_blocks.py:312                     */
  int32_t chunk_size;        /* StreamingConformerEncoder.add_chunk_size (samples); 0 = offline */
  int32_t has_encoder;       /* 1: handle owns a ConformerEncoder; 0: CTCDecoder-only handle  */
  int32_t num_classes;       /* CTCDecoder num_classes (blank = num_classes-1); 0 = no CTC head */
  int32_t ctc_num_blocks;    /* model_config.ctcdecoder_num_blocks        1                    */
  int32_t ctc_kernel_size;   /* model_config.ctcdecoder_kernel_size       32                   */
  float   ctc_fc_factor;     /* model_config.ctcdecoder_fc_factor         0.5                  */
  int32_t gemm_dtype;        /* 0: fp32 values, fp32-accurate products everywhere (see the header comment; default)
                              * 1: bf16 MFMA for the dense layers -- bf16 GEMM inputs, fp32 accumulation, fp32
                              *    LayerNorm / softmax / activations / frontend (BASELINE config 3)               */
  int32_t mel_layer_type;    /* speech_config.mel_layer_type: 0 = 'Melspectrogram' (default), 1 = 'leaf' (LEAF frontend,
                              *    leaf_audio/frontend.py: Gabor filters + Gaussian pooling + PCEN + instance norm;
                              *    needs n_mels 80, stride_ms 10 at 16 kHz), 2 = 'Spectrogram' (any other value in the
                              *    reference, conformer_blocks.py:318-323: the n_dft/2+1 = 513 dB bins without the mel
                              *    matrix; n_mels is ignored and no freq2mel tensor exists)                        */
  int32_t add_wav_info;      /* speech_config.add_wav_info: 1 adds WavePickModel(waveform) (asr/models/wav_model.py:108-146)
                              *    to the subsampled features (conformer_blocks.py:344-348); needs L % hop_size == 0   */
} mi355asr_config;

const char* mi355asr_last_error(void);
const char* mi355asr_version(void);

/* replaces: ConformerEncoder(...)/CTCDecoder(...) construction, test_asr.py:28-75 */
int mi355asr_create(const mi355asr_config* cfg, mi355asr_model** out);
int mi355asr_destroy(mi355asr_model* m);

/* replaces: model.load_weights(path[, by_name=True]), test_asr.py:95-114.  `name` is the Keras-layout
 * tensor name (see DESIGN.md "weight names"), data is host fp32 in the Keras layout of that tensor.
 * May be called again after finalisation to overwrite a tensor (then finalise again). */
int mi355asr_load_weight(mi355asr_model* m, const char* name, const float* data_host, int32_t rank,
                         const int64_t* dims);
/* the same for checkpoints that store other element types (`load_weights(handle, name, host_ptr, dtype, rank,
 * dims)`): the tensor is converted to fp32 on the host.  bf16 / fp16 are the raw 16-bit patterns. */
#define MI355ASR_DT_F32 0
#define MI355ASR_DT_F16 1
#define MI355ASR_DT_BF16 2
#define MI355ASR_DT_F64 3
int mi355asr_load_weight_typed(mi355asr_model* m, const char* name, const void* data_host, int32_t dtype, int32_t rank,
                               const int64_t* dims);
/* number of tensors the configuration expects / name of the i-th one (so loaders can iterate) */
int mi355asr_num_weights(const mi355asr_model* m);
const char* mi355asr_weight_name(const mi355asr_model* m, int32_t i);
/* replaces: iterating model.weights / model.summary() after _build() (test_asr.py:85-93): Keras-layout shape of the
 * i-th tensor, *rank and dims[0..*rank); a non-Python caller sizes its buffers with it (examples/asr_session.cpp). */
int mi355asr_weight_shape(const mi355asr_model* m, int32_t i, int32_t* rank, int64_t* dims, int32_t max_rank);
/* replaces: model._build() (test_asr.py:85-87): checks every tensor is present, packs the matrices into
 * MFMA fragment order, folds BatchNorm into (scale, shift), uploads to HBM. */
int mi355asr_finalize_weights(mi355asr_model* m, void* stream);
/* optional, before mi355asr_finalize_weights: the most rows (batch x encoder frames) one call will bring.  Handles of
 * dmodel 256 / 512 pack every dense layer a second time as a split-bf16 slab ring for batches of >= 1500 rows (1.5 x the
 * dense weights' bytes, and their packing time); a handle that will only see single utterances or streaming chunks says
 * so here and keeps the per-wave kernels (and, at dmodel 256, the fused chains).  rows < 0 = unknown (the default: pack). */
int mi355asr_set_expected_rows(mi355asr_model* m, int64_t rows);
/* which STFT kernel mi355asr_finalize_weights selected: 1 = 32x32 Cooley-Tukey on the matrix cores (the loaded
 * mel_layer/{real,imag}_kernels are window[n]*exp(-2*pi*i*k*n/1024), as backend.py:27-69 builds them), 0 = dense DFT
 * GEMM with the kernels as loaded (a checkpoint changed them), -1 = no frontend / not finalized. */
int mi355asr_stft_mode(const mi355asr_model* m);

/* shape helpers: mel frames F = ceil(L/hop), encoder frames T = ceil(ceil(F/2)/2) per block of L samples */
int mi355asr_out_frames(const mi355asr_model* m, int32_t L, int32_t* mel_frames, int32_t* enc_frames);
/* bytes of caller-provided device scratch needed by any forward call on [B, L] */
int mi355asr_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t L, size_t* bytes);
/* same for the calls that start from encoder frames (ctc_forward, conformer_block): [B, T, dmodel] */
int mi355asr_ctc_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t T, size_t* bytes);

/* replaces: encoder(wav, training=False) / encoder.inference(wav)  (conformer_blocks.py:343-378, 574-594);
 *           ASR::Session::EncoderInference (asr_session.cpp:77-98), ONNX "inputs" -> "Identity:0".
 * wav_dev f32 [B, L] (the reference's trailing channel dim of 1 dropped) -> enc_out_dev f32 [B, T_total, dmodel]
 * with T_total = (L/chunk_size)*T(chunk_size) when chunk_size>0 else T(L). */
int mi355asr_encoder_forward(mi355asr_model* m, const float* wav_dev, int32_t B, int32_t L, float* enc_out_dev,
                             void* ws_dev, size_t ws_bytes, void* stream);

/* replaces: ctc_model(enc, training=False) (conformer_blocks.py:419-424); ASR::Session::CTCInference
 * (asr_session.cpp:100-123), ONNX "inputs" [B,T,d] -> "Identity:0" [B,T,V].
 * logits_dev (f32 [B,T,V]) and frame_argmax_dev (i32 [B,T]) may each be NULL.  The argmax is the one
 * tf.keras.backend.ctc_decode / ctc_greedy_decoder.h:9-20 take per frame (first maximum wins). */
int mi355asr_ctc_forward(mi355asr_model* m, const float* enc_dev, int32_t B, int32_t T, float* logits_dev,
                         int32_t* frame_argmax_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* replaces: tf.keras.backend.ctc_decode(probs, input_length)[0][0] greedy (test_asr.py:196-200) and
 * ctc_greedy_decoder(probs, blank_id, vocab) (ctc_greedy_decoder.h:5-44): merge repeated, drop blank,
 * dense output padded with -1.  in_len_dev may be NULL (= T for every utterance). Model-independent. */
int mi355asr_ctc_greedy(const int32_t* frame_argmax_dev, const int32_t* in_len_dev, int32_t B, int32_t T,
                        int32_t blank, int32_t* ids_dev, int32_t* out_len_dev, void* stream);

/* replaces: ctc_beam_search_decoder_batch(probs_split, vocabulary, beam_size, num_processes, cutoff_prob,
 * cutoff_top_n, ext_scorer = nullptr) of externals/ctc_decoders (ctc_beam_search_decoder.cpp:18-187, 426-459;
 * SWIG entry decoders.i) -- the scorer-less CTC prefix beam search.  Blank = class V-1, vocabulary = classes
 * 0..V-2 (as the reference assigns blank_id = vocabulary.size()).  Token ids are returned instead of the
 * concatenated vocabulary strings.  Outputs are HOST buffers: ids i32 [B, beam, max_len] (-1 padded, hypotheses
 * best first), lens i32 [B, beam], scores f32 [B, beam] (log prob), n_hyp i32 [B] (number of valid hypotheses).
 * Reference quirk kept: with cutoff_prob == 1.0 no class is pruned, whatever cutoff_top_n says
 * (decoder_utils.cpp:18-31).
 *   _host : probs_host f32 [B, T, V] on the host; everything runs on the CPU threads (model-independent helper,
 *           also what the unit tests pin against the reference's own decoder).
 *   device: x_dev f32 [B, T, V] logits (is_logits != 0: softmax is fused into the selection kernel) or
 *           probabilities; the per-frame top-cutoff_top_n selection runs on the GPU, the prefix search on
 *           prefix search on the device as well (beam_device.hip: one workgroup per utterance, the beam in LDS; beam_size
 *           <= 128, cutoff_top_n <= 40, ws_dev of mi355asr_ctc_prefix_beam_workspace_bytes) or, failing those, on
 *           `num_threads` host threads.  Needs cutoff_prob < 1 and cutoff_top_n <= 128 (otherwise use _host).
 *           Synchronises `stream` (the results are host arrays).
 *           ws_dev: at least B*T*cutoff_top_n*8 bytes (host search); mi355asr_ctc_prefix_beam_workspace_bytes for the
 *           device search. */
int mi355asr_ctc_prefix_beam_host(const float* probs_host, const int32_t* in_len_host, int32_t B, int32_t T, int32_t V,
                                  int32_t beam_size, double cutoff_prob, int32_t cutoff_top_n, int32_t num_threads,
                                  int32_t max_len, int32_t* ids_host, int32_t* lens_host, float* scores_host,
                                  int32_t* n_hyp_host);
int mi355asr_ctc_prefix_beam(const float* x_dev, int32_t is_logits, const int32_t* in_len_host, int32_t B, int32_t T,
                             int32_t V, int32_t beam_size, double cutoff_prob, int32_t cutoff_top_n,
                             int32_t num_threads, int32_t max_len, int32_t* ids_host, int32_t* lens_host,
                             float* scores_host, int32_t* n_hyp_host, void* ws_dev, size_t ws_bytes, void* stream);

/* The score arithmetic of the device prefix search, exposed for verification.  The reference's scores are defined by the
 * C library it is compiled against: log_sum_exp<float> = logf(expf(x - m) + expf(y - m)) + m (zip:ctc_decoders/
 * decoder_utils.h:41-49) and (float)log((double)p + FLT_MIN) (ctc_beam_search_decoder.cpp:57-59); the device search
 * evaluates glibc's own algorithms (csrc/refmath.h).  kind 0: out f32[i] = expf(in[i]), in [-17.5, 0];
 * 1: out f32[i] = logf(in[i]), in [1, 2]; 2: out f64[i] = log((double)in[i] + FLT_MIN), in [0, 1];
 * 3: out f32[i] = log_sum_exp(in[i], in[n + i]) (in holds 2 n values).  Device pointers. */
int mi355asr_beam_math_eval(int32_t kind, const float* in_dev, void* out_dev, int32_t n, void* stream);

/* Stateful prefix beam search for streaming recognition.
 * replaces: class BeamDecoder (zip:ctc_decoders/ctc_beam_search_decoder.h: BeamDecoder(vocabulary, beam_size,
 * cutoff_prob, cutoff_top_n, ext_scorer = nullptr), .decode(probs_seq), .reset(); .cpp:217-405).  decode() consumes
 * T more frames, continuing from the prefix trie the previous calls left, and returns the current beam (best
 * first) -- feeding an utterance in pieces gives the result of feeding it whole.  As in the reference class, the
 * vocabulary INCLUDES the blank as its last entry: probs rows have V = num_classes entries, blank = V-1 (:238-240).
 * Host-side (the trie search is branchy integer work); probs_host f32 [T, V]; outputs as in
 * mi355asr_ctc_prefix_beam_host for one utterance: ids i32 [beam, max_len], lens i32 [beam], scores f32 [beam]. */
int mi355asr_ctc_prefix_beam_workspace_bytes(int32_t B, int32_t T, int32_t cutoff_top_n, int32_t beam_size, int32_t max_len,
                                             size_t* bytes);

typedef struct mi355asr_beam mi355asr_beam;
int mi355asr_beam_create(int32_t V, int32_t beam_size, double cutoff_prob, int32_t cutoff_top_n, mi355asr_beam** out);
int mi355asr_beam_decode(mi355asr_beam* d, const float* probs_host, int32_t T, int32_t max_len, int32_t* ids_host,
                         int32_t* lens_host, float* scores_host, int32_t* n_hyp_host);
int mi355asr_beam_reset(mi355asr_beam* d);
int mi355asr_beam_destroy(mi355asr_beam* d);

/* encoder + CTCDecoder + greedy in one call: wav [B,L] -> ids i32 [B,T_total] (-1 padded), out_len i32 [B].
 * This is the timed region of bench.py (offline_stt steps 3-5, test_asr.py:191-198). */
int mi355asr_recognize(mi355asr_model* m, const float* wav_dev, int32_t B, int32_t L, const int32_t* in_len_dev,
                       int32_t* ids_dev, int32_t* out_len_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Stage-level entry points (same kernels the calls above run; exposed so that the parity tests can
 * localise a mismatch to one reference layer):
 *   melspectrogram   Melspectrogram.call (time_frequency.py:173-189): wav [B,L] -> mel [B,F,n_mels]
 *   conv_subsampling ConvSubsampling.call (conformer_blocks.py:90-96): mel [B,F,n_mels] -> [B,T,d]
 *   conformer_block  ConformerBlock.call (conformer_blocks.py:259-265) of block `index` of the encoder
 *                    (stack 0) or the CTCDecoder (stack 1): x [B,T,d] -> y [B,T,d] */
int mi355asr_melspectrogram(mi355asr_model* m, const float* wav_dev, int32_t B, int32_t L, float* mel_dev,
                            void* ws_dev, size_t ws_bytes, void* stream);
int mi355asr_conv_subsampling(mi355asr_model* m, const float* mel_dev, int32_t B, int32_t F, float* out_dev,
                              void* ws_dev, size_t ws_bytes, void* stream);
int mi355asr_conformer_block(mi355asr_model* m, int32_t stack, int32_t index, const float* x_dev, int32_t B,
                             int32_t T, float* y_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ---- ChunkConformer (asr/models/chunk_conformer_blocks.py:775-822, offline `predict`) ---------------------------
 * front (valid-padded Melspectrogram + left-padded VALID ConvSubsampling, :400-445, :23-70) -> ChunkConformerEncoder
 * (band attention [i-win_front, i+win_back], causal depthwise conv, :142-398, :462-560) -> phone_picker
 * (ChunkCTCDecoder :571-637) -> feature_pick (keep frames whose phone argmax is not the blank, :913-999) ->
 * ContextHelper (:679-770) -> text ChunkCTCDecoder -> logits [B, T_pick, num_classes].
 * Values from asr/configs/chunk_conformerS.yml.  All sub-models share dmodel / heads / kernel size here. */
typedef struct {
  int32_t dmodel, head_size, num_heads, kernel_size;      /* 144, 36, 4, 32                                 */
  float   fc_factor;                                      /* 0.5                                            */
  int32_t n_mels, sample_rate, stride_ms, n_dft;          /* 80, 16000, 10, 1024                            */
  int32_t reduction_factor;                               /* 4                                              */
  int32_t enc_num_blocks, enc_win_front, enc_win_back;    /* 15, 36, 0                                      */
  int32_t picker_num_classes, picker_num_blocks, picker_win_front, picker_win_back;   /* phone+1, 1, 36, 0 */
  int32_t helper_num_blocks, helper_win_front, helper_win_back;                       /* 2, 36, 0          */
  int32_t decoder_num_classes, decoder_num_blocks, decoder_win_front, decoder_win_back; /* txt+1, 1, 36, 8 */
} mi355asr_chunk_config;

/* optional DEVICE outputs of mi355asr_chunk_predict (NULL = not wanted).  Frame-major fp32 unless noted;
 * T = encoder frames of the utterances, Tp = max over the batch of picked frames (returned on the host). */
typedef struct {
  float*   front_out;       /* [B, T, d]                    ChunkConformerFront.call                      */
  float*   enc_out;         /* [B, T, d]                    ChunkConformerEncoder.call                    */
  float*   picker_logits;   /* [B, T, picker_num_classes]   phone_picker(...)[0]                          */
  float*   picker_hidden;   /* [B, T, d]                    phone_picker(...)[1]                          */
  float*   picked;          /* [B, >=Tp, d] (capacity B*T*d) feature_pick(...)[0], zero padded            */
  float*   helper_out;      /* [B, >=Tp, d] (capacity B*T*d) helper(picked)                               */
  float*   text_logits;     /* [B, Tp, decoder_num_classes] (capacity B*T*V) ChunkConformer.predict       */
  int32_t* text_argmax;     /* i32 [B, Tp] (capacity B*T)   per-frame argmax of text_logits               */
} mi355asr_chunk_outputs;

/* replaces: ChunkConformer(config, phone, txt) construction (test_chunk_asr.py:40-55).  The handle takes
 * weights through mi355asr_load_weight / mi355asr_finalize_weights like the other models (names: DESIGN.md). */
int mi355asr_chunk_create(const mi355asr_chunk_config* cfg, mi355asr_model** out);
int mi355asr_chunk_out_frames(const mi355asr_model* m, int32_t L, int32_t* mel_frames, int32_t* enc_frames);
int mi355asr_chunk_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t L, size_t* bytes);
/* replaces: ChunkConformer.predict(x) (chunk_conformer_blocks.py:815-822).  wav_dev f32 [B, L].
 * n_picked_host i32 [B] and t_pick_host i32 [1] are HOST outputs (the picked-frame counts size the second half of
 * the network, so the call synchronises `stream` once in the middle, as the reference's dynamic shapes do). */
int mi355asr_chunk_predict(mi355asr_model* m, const float* wav_dev, int32_t B, int32_t L,
                           const mi355asr_chunk_outputs* outs, int32_t* n_picked_host, int32_t* t_pick_host,
                           void* ws_dev, size_t ws_bytes, void* stream);

/* per-frame argmax of given logits or probabilities, f32 [M, V] -> i32 [M], first maximum wins: the decision step of
 * tf.keras.backend.ctc_decode(greedy) when the logits come from outside a head kernel (the streaming ChunkConformer path
 * concatenates logits of several calls, test_chunk_asr.py:84-96); feed the result to mi355asr_ctc_greedy. */
int mi355asr_frame_argmax(const float* x_dev, int32_t M, int32_t V, int32_t* out_dev, void* stream);

/* replaces: ChunkConformer.feature_pick(encoder_hidden_states, ctc_outs[, max_T]) (chunk_conformer_blocks.py:913-999), the
 * "length regulator" between the phone picker and the text decoder of the streaming path (test_chunk_asr.py:72-75): keep
 * the frames whose phone argmax is not the blank (class V - 1), compacted per utterance, zero padded to the batch maximum.
 * Handle-free, two calls because the batch maximum sizes the outputs (a dynamic shape in the reference):
 *   _count:  ctc_dev f32 [B, T, V] -> idx_dev i32 [B, T] (kept frame indices), cnt_dev i32 [B], counts_host i32 [B];
 *            synchronises the stream once to return the counts
 *   _gather: hidden_dev f32 [B, T, d], ctc_dev -> feat_out_dev f32 [B, Tp, d], ctc_out_dev f32 [B, Tp, V] (or NULL)
 *            for any Tp >= max(counts) (max_T of the reference) */
int mi355asr_feature_pick_count(const float* ctc_dev, int32_t B, int32_t T, int32_t V, int32_t* idx_dev, int32_t* cnt_dev,
                                int32_t* counts_host, void* stream);
int mi355asr_feature_pick_gather(const float* hidden_dev, const float* ctc_dev, const int32_t* idx_dev,
                                 const int32_t* cnt_dev, int32_t B, int32_t T, int32_t d, int32_t V, int32_t Tp,
                                 float* feat_out_dev, float* ctc_out_dev, void* stream);

/* ---- ChunkConformer streaming: one stream, explicit caches ------------------------------------------
 * replaces the pieces of ChunkConformer.picker_stream_predict / decoder_stream_predict (chunk_conformer_blocks.py:
 * 824-866, ONNX form :868-898).  The caller owns every cache tensor and does the slicing the reference does in
 * Python (valid / unvalid split by win_back, caches cut to the last win_front / kernel_size rows, dec_inp carry);
 * tensorflowasr_amd.models.ChunkConformer.{init_picker_caches, picker_stream_predict, init_decoder_caches,
 * decoder_stream_predict, feature_pick} is that code.  All tensors are device f32, row-major, batch 1.
 *
 * front_stream: ChunkConformerFront.stream_call (:447-458) + ConvSubsampling.stream_call (:72-91).
 *   wav_dev [Lw] = [front_wav_cache ; new samples]; sub_cache_dev [S, n_mels]
 *   -> new_sub_dev [S + nf, n_mels] = [sub cache ; last nf mel frames], front_out_dev [t_out, d]
 *   (nf = min(mel frames of the buffer, chunk_num), t_out = min(frames after the two stride-2 convs, chunk_num /
 *   reduction_factor): mi355asr_chunk_front_stream_shape).
 * stack_stream: ChunkConformerEncoder (stack 0) / phone picker ChunkCTCDecoder (1) / ContextHelper (2) / text
 *   ChunkCTCDecoder (3) .stream_call (:530-560, 641-672, 750-770) up to (not including) the valid/unvalid slicing:
 *   x_dev [T, d] (picker / decoder: dec_inp rows followed by the new rows; `project` is applied inside),
 *   mha_cache_dev [num_blocks, Cm, d], cnn_cache_dev [num_blocks, Cc, d]
 *   -> hidden_dev [T, d] (block-stack output), logits_dev [T, num_classes] / argmax_dev i32 [T] (stacks 1, 3; NULL =
 *   not wanted), new_mha_dev [num_blocks, Cm + T, d], new_cnn_dev [num_blocks, Cc + T, d] = [cache ; module input]
 *   per block, untrimmed.  Band attention is evaluated with the queries as the last T rows of the Cm + T keys. */
int mi355asr_chunk_front_stream_shape(const mi355asr_model* m, int32_t Lw, int32_t S, int32_t chunk_num, int32_t* nf,
                                      int32_t* t_out);
/* max_rows = largest (cache rows + T) of any stack_stream call; Lw, S, chunk_num as for front_stream */
int mi355asr_chunk_stream_workspace_bytes(const mi355asr_model* m, int32_t max_rows, int32_t Lw, int32_t S,
                                          int32_t chunk_num, size_t* bytes);
int mi355asr_chunk_front_stream(mi355asr_model* m, const float* wav_dev, int32_t Lw, const float* sub_cache_dev,
                                int32_t S, int32_t chunk_num, float* front_out_dev, float* new_sub_dev, void* ws_dev,
                                size_t ws_bytes, void* stream);
int mi355asr_chunk_stack_stream(mi355asr_model* m, int32_t stack, const float* x_dev, int32_t T,
                                const float* mha_cache_dev, int32_t Cm, const float* cnn_cache_dev, int32_t Cc,
                                float* hidden_dev, float* logits_dev, int32_t* argmax_dev, float* new_mha_dev,
                                float* new_cnn_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* ---- Translator: phoneme ids + encoder output -> text logits -------------------------------
 * replaces: Translator(inp_classes, tar_classes, dmodel, num_blocks, head_size, num_heads, kernel_size, dropout,
 * fc_factor) (test_asr.py:76-84; conformer_blocks.py:505-548): Embedding(inp_classes -> d) -> num_blocks x RBlock
 * (FFModule -> cross-attention with q = LN(x + sinusoid PE), k = v = encoder output -> ConvModule -> FFModule -> LN)
 * -> Dense(d -> tar_classes).  Weight names: inp_embedding/embeddings, decoder_conformer_block_<i>/... (as the
 * ConformerBlock), fully_connected/{kernel,bias}. */
typedef struct {
  int32_t dmodel, num_blocks, head_size, num_heads, kernel_size;
  float   fc_factor;
  int32_t inp_classes, tar_classes;
} mi355asr_translator_config;
int mi355asr_translator_create(const mi355asr_translator_config* cfg, mi355asr_model** out);
/* U = token positions per utterance (padded CTC output), T = encoder frames per utterance */
int mi355asr_translator_workspace_bytes(const mi355asr_model* m, int32_t B, int32_t U, int32_t T, size_t* bytes);
/* replaces: translator([ctc_decode, enc_outputs], training=False) and tf.argmax(., -1) (test_asr.py:202-203,
 * streaming :149-150).  ids_dev i32 [B, U] (values clamped to [0, inp_classes)), enc_dev f32 [B, T, d];
 * logits_dev f32 [B, U, tar_classes] or NULL, argmax_dev i32 [B, U] or NULL (first maximum wins). */
int mi355asr_translator_forward(mi355asr_model* m, const int32_t* ids_dev, const float* enc_dev, int32_t B,
                                int32_t U, int32_t T, float* logits_dev, int32_t* argmax_dev, void* ws_dev,
                                size_t ws_bytes, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream around each kernel (off by default).
 * profile_read waits for the recorded events, then returns accumulated milliseconds and launch counts per
 * kernel category below (arrays of at least MI355ASR_NUM_KERNELS); reset != 0 clears the accumulators.
 * bench.py uses this for the live `roofline.achieved` figure. */
#define MI355ASR_K_STFT 0         /* stft_kernel          Spectrogram conv2d x2 + power + log           */
#define MI355ASR_K_UTT_MAX 1      /* utt_max_kernel       per-sample max of the dB spectrogram          */
#define MI355ASR_K_MEL 2          /* mel_kernel           (dB-max).clamp(-80) @ freq2mel                */
#define MI355ASR_K_SUBCONV 3      /* subconv_kernel       Conv2D+ReLU -> Conv2D+ReLU (fused)            */
#define MI355ASR_K_SUBLINEAR 4    /* stream_gemm_kernel   Dense(F2*d -> d)                              */
#define MI355ASR_K_FFN 5          /* chain2_kernel mode 0 FFModule (+ block LayerNorm on the 2nd one)   */
#define MI355ASR_K_QKV 6          /* gemm_rows EPI_QKV    LN + q/k/v projections                        */
#define MI355ASR_K_ATTN 7         /* attention_kernel     softmax(q k^T) v                              */
#define MI355ASR_K_ATTN_OUT 8     /* gemm_rows EPI_RESIDUAL out-projection + residual                   */
#define MI355ASR_K_PW1_GLU 9      /* gemm_rows EPI_GLU    LN + pw_conv_1 + GLU                          */
#define MI355ASR_K_DWCONV 10      /* dwconv_kernel        depthwise conv                                */
#define MI355ASR_K_CONV_TAIL 11   /* chain2_kernel mode 1 pointwise + BN + swish + pw_conv_2 + residual */
#define MI355ASR_K_CTC_PROJECT 12 /* gemm_rows EPI_BIAS   CTCDecoder.project                            */
#define MI355ASR_K_CTC_HEAD 13    /* gemm_rows EPI_HEAD   fully_connected + per-frame argmax            */
#define MI355ASR_K_COLLAPSE 14    /* collapse_kernel      greedy merge/blank-drop                       */
#define MI355ASR_K_FF1_QKV 15     /* ff1_qkv_kernel       FFModule 1 + LN + q/k/v projections (fused, dmodel 144)     */
#define MI355ASR_K_OUT_GLU 16     /* out_glu_kernel       out-projection + residual + LN + pw_conv_1 + GLU (fused)    */
#define MI355ASR_K_TAIL_FF2 17    /* tail_ff2_kernel      ConvModule tail + FFModule 2 + block LayerNorm (fused)      */
#define MI355ASR_K_TAIL_FF1 18    /* tail_ff1_ld_kernel   tail_ff2 of block i + ff1_qkv of block i + 1 in one launch  */
#define MI355ASR_K_ENC_STACK 19   /* stream256_kernel     every ConformerBlock of the streaming encoder, one workgroup per chunk (bf16 mode) */
#define MI355ASR_NUM_KERNELS 20
int mi355asr_profile_enable(mi355asr_model* m, int32_t on);
/* Which arithmetic the LAST launch of each kernel category used (recorded whether or not timing is enabled; -1: the category
 * has not run on this handle).  Several kernels exist for most categories -- chosen by dmodel, row count, whether an operand
 * bound is known, and the MI355ASR_* experiment switches -- and they run on different pipes; whoever prices a kernel against
 * a roofline (bench.py) asks the library instead of re-deriving the choice. */
#define MI355ASR_SCHEME_F32 0     /* exact fp32 products: v_mfma_f32_16x16x4_f32 / fp32 VALU                           */
#define MI355ASR_SCHEME_BF16X3 1  /* fp32 operands as three bf16 terms, six bf16 MFMAs per fragment pair (exact to 2^-24) */
#define MI355ASR_SCHEME_F16X2 2   /* fp32 operands as two fp16 terms, three fp16 MFMAs per fragment pair (2^-22 of the bound) */
#define MI355ASR_SCHEME_BF16 3    /* operands rounded to bf16 (gemm_dtype = 1)                                         */
int mi355asr_profile_schemes(const mi355asr_model* m, int32_t* scheme_out, int32_t n);
int mi355asr_profile_read(mi355asr_model* m, double* ms_out, int64_t* count_out, int32_t n, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* MI355ASR_H */
