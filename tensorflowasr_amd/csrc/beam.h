// internal interface between beam.hip and api.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" {
int mi355asr_beam_host_impl(const float* probs, const int32_t* in_len, int B, int T, int V, int beam_size,
                            double cutoff_prob, int cutoff_top_n, int num_threads, int max_len, int32_t* ids,
                            int32_t* lens, float* scores, int32_t* n_hyp);
int mi355asr_beam_topn_impl(const int32_t* top_idx, const float* top_p, const int32_t* in_len, int B, int T, int V,
                            int N, int beam_size, double cutoff_prob, int cutoff_top_n, int num_threads, int max_len,
                            int32_t* ids, int32_t* lens, float* scores, int32_t* n_hyp);
void* mi355asr_beam_state_new(int V, int beam_size, double cutoff_prob, int cutoff_top_n);
void mi355asr_beam_state_free(void* h);
void mi355asr_beam_state_reset(void* h);
int mi355asr_beam_state_decode(void* h, const float* probs, int T, int max_len, int32_t* ids, int32_t* lens, float* scores);
// device prefix search (beam_device.hip): every pointer is a DEVICE pointer
struct BeamDeviceArgs {
  const int32_t* top_idx;   // [B, T, N] classes in descending probability (topn_kernel)
  const float* top_p;       // [B, T, N]
  const int32_t* in_len;    // [B] or null
  int B, T, V, N, beam, cutoff_top_n, max_len;
  double cutoff_prob;
  int32_t* ids;             // [B, beam, max_len] padded with -1
  int32_t* lens;            // [B, beam]
  float* scores;            // [B, beam]
  int32_t* n_hyp;           // [B]
  int2* arena;              // [B, T * beam + 1] back-pointers (parent link, character)
  long long* prof;          // null, or 9 counters of utterance 0 (MI355ASR_BEAM_PROF)
};
bool mi355asr_beam_device_applicable(int V, int N, int beam);
size_t mi355asr_beam_device_ws_bytes(int B, int T, int beam, int max_len);
// the ONE place that lays the device search's buffers out in its workspace (16-byte aligned segments; ws null: size only):
// fills a->arena / ids / lens / scores / n_hyp, *d_len (staging of in_len) and *prof (9 x int64 counters); returns the bytes used
size_t mi355asr_beam_device_carve(char* ws, int B, int T, int beam, int max_len, BeamDeviceArgs* a, int32_t** d_len, long long** prof);
int mi355asr_launch_beam_device(const BeamDeviceArgs* a, hipStream_t s);
// kind 0 expf(in[i]) -> float, 1 logf -> float, 2 log((double)in[i] + FLT_MIN) -> double, 3 log_sum_exp(in[i], in[n + i]) -> float,
// evaluated by the device search's own routines (refmath.h)
int mi355asr_launch_refmath_eval(int kind, const float* in, void* out, int n, hipStream_t s);
int mi355asr_launch_topn(const float* x_dev, int frames, int V, int N, int is_logits, int32_t* idx_dev, float* p_dev,
                         hipStream_t s);
}
