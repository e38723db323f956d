// Frontend + subsampling + decode kernels for gfx950.
//   stft_kernel      Spectrogram._spectrogram_mono (asr/models/layers/time_frequency.py:100-122) as an
//                    fp32-MFMA GEMM against the model's own DFT kernels + power + log (backend_keras.py:14)
//   utt_max_kernel   the per-sample K.max over (frames, freq)                   (backend_keras.py:20)
//   mel_kernel       (dB - max).clamp(-80) @ freq2mel        (backend_keras.py:20-22, time_frequency.py:181)
//   subconv_kernel   Conv2D(3x3,s2,SAME)+ReLU -> Conv2D(3x3,s2,SAME)+ReLU fused: conv1 is recomputed on the fly
//                    as the operand of the conv2 implicit GEMM            (conformer_blocks.py:76-92)
//   stream_gemm      Dense(F2*d -> d) of ConvSubsampling                   (conformer_blocks.py:87,95)
//   collapse_kernel  CTC greedy merge-repeated / drop-blank / pad -1       (test_asr.py:196-200,
//                    Inference/CppInference/onnx/src/core/ctc_greedy_decoder.h:22-43)
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "env.h"
#include "launch.h"

// ---------------------------------------------------------------------------------------------------
// STFT power -> log.  One wave = 16 frames of one utterance x CT column tiles (8 bins per tile, re/im
// interleaved so that a lane's float4 accumulator is (re,im,re,im) of two adjacent bins).
// ---------------------------------------------------------------------------------------------------
template <int RT, int CT>
__global__ __launch_bounds__(BLOCK_THREADS, 2) void stft_kernel(StftArgs a) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, g4 = g * 4, c = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int FTP = (a.FT + RT - 1) / RT;           // groups of RT frame tiles per utterance
  if (wid >= a.B * FTP) return;
  const int b = wid / FTP, ftp = wid % FTP;
  const int chunk = blockIdx.y;
  const int c0 = chunk * CT;
  const float* __restrict__ wav = a.wav + (size_t)b * a.L;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + lane;
  const int KBT = a.n_dft / 16;
  const int L = a.L;
  int f[RT], fc[RT], sbase[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f[rt] = (ftp * RT + rt) * 16 + c;
    fc[rt] = min(f[rt], a.F - 1);
    sbase[rt] = fc[rt] * a.hop - a.pad_left + g4;
  }

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[rt][i] = splat4(0.f);

  // frame fragment: 4 consecutive samples of the zero-padded signal (TF 'SAME' padding); branch-free so that
  // the loads can be hoisted a whole k-step ahead of their MFMAs
  auto xp = [&](int rt, int kb) -> f32x4 {
    const int s0 = sbase[rt] + 16 * kb;
    f32x4 x;
    x.x = wav[(unsigned)(s0 + 0) < (unsigned)L ? s0 + 0 : 0];
    x.y = wav[(unsigned)(s0 + 1) < (unsigned)L ? s0 + 1 : 0];
    x.z = wav[(unsigned)(s0 + 2) < (unsigned)L ? s0 + 2 : 0];
    x.w = wav[(unsigned)(s0 + 3) < (unsigned)L ? s0 + 3 : 0];
    return x;
  };
  auto fx = [&](int rt, int kb, f32x4 x) -> f32x4 {
    const int s0 = sbase[rt] + 16 * kb;
    x.x = (unsigned)(s0 + 0) < (unsigned)L ? x.x : 0.f;
    x.y = (unsigned)(s0 + 1) < (unsigned)L ? x.y : 0.f;
    x.z = (unsigned)(s0 + 2) < (unsigned)L ? x.z : 0.f;
    x.w = (unsigned)(s0 + 3) < (unsigned)L ? x.w : 0.f;
    return x;
  };
  sweep_k<RT, CT>(acc, wp, a.NT, c0, KBT, xp, fx);

#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int ft = ftp * RT + rt;
    float mx = -INFINITY;
    const bool fvalid = f[rt] < a.F;
    float* orow = a.logp + ((size_t)b * a.F + fc[rt]) * a.LP;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int bin0 = 8 * (c0 + i) + 2 * g;
      const f32x4 v = acc[rt][i];
      const float p0 = v.x * v.x + v.y * v.y;
      const float p1 = v.z * v.z + v.w * v.w;
      float l0 = logf(fmaxf(p0, 1e-10f));
      float l1 = logf(fmaxf(p1, 1e-10f));
      if (a.db10) { l0 = 10.0f * l0 / 2.30258509f; l1 = 10.0f * l1 / 2.30258509f; }
      else { l0 = l0 / 2.30258509f; l1 = l1 / 2.30258509f; }
      if (fvalid) {
        f32x2 o = {l0, l1};
        *reinterpret_cast<f32x2*>(orow + bin0) = o;
        if (bin0 < a.nbins) mx = fmaxf(mx, l0);
        if (bin0 + 1 < a.nbins) mx = fmaxf(mx, l1);
      }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0 && ft < a.FT) a.pmax[(size_t)b * a.FT * a.NCH + ft * a.NCH + chunk] = mx;
  }
}

int launch_stft(const StftArgs& a, hipStream_t s) {
  constexpr int CT = 13, RT = 2;
  if (a.NT % CT != 0 || a.NCH != a.NT / CT) return -1;
  const int FTP = (a.FT + RT - 1) / RT;
  dim3 grid((a.B * FTP + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, a.NCH);
  hipLaunchKernelGGL((stft_kernel<RT, CT>), grid, dim3(BLOCK_THREADS), 0, s, a);
  return 0;
}

__global__ void utt_max_kernel(UttMaxArgs a) {
  const int b = blockIdx.x;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < a.n; i += 64) mx = fmaxf(mx, a.pmax[(size_t)b * a.n + i]);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (threadIdx.x == 0) a.umax[b] = mx;
}

int launch_utt_max(const UttMaxArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(utt_max_kernel, dim3(B), dim3(64), 0, s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// dB normalisation + mel projection.  One wave = 16 frames x all mel tiles.
// ---------------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(BLOCK_THREADS) void mel_kernel(MelArgs a) {
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (wid >= a.B * a.FT) return;
  const int b = wid / a.FT, ft = wid % a.FT;
  const int f = ft * 16 + c;
  const int fc = min(f, a.F - 1);
  const float* __restrict__ row = a.logp + ((size_t)b * a.F + fc) * a.LP;
  const bool norm = a.umax != nullptr;
  const float um = norm ? a.umax[b] : 0.f;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + lane;
  f32x4 acc[1][CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) acc[0][i] = splat4(0.f);
  const int LP = a.LP, nbins = a.nbins;
  const float floor_db = a.floor_db;
  auto xp = [&](int, int kb) -> f32x4 {
    const int k = 16 * kb + g4;
    return ldg4(row + (k < LP ? k : 0));
  };
  auto fx = [&](int, int kb, f32x4 x) -> f32x4 {
    const int k = 16 * kb + g4;
    if (norm) {
      x.x = fmaxf(x.x - um, floor_db);
      x.y = fmaxf(x.y - um, floor_db);
      x.z = fmaxf(x.z - um, floor_db);
      x.w = fmaxf(x.w - um, floor_db);
    }
    x.x = (k + 0 < nbins) ? x.x : 0.f;
    x.y = (k + 1 < nbins) ? x.y : 0.f;
    x.z = (k + 2 < nbins) ? x.z : 0.f;
    x.w = (k + 3 < nbins) ? x.w : 0.f;
    return x;
  };
  sweep_k<1, CT>(acc, wp, a.NTm, 0, a.KBm, xp, fx);
  if (f < a.F) {
    float* orow = a.mel + ((size_t)b * a.F + f) * a.NM;
#pragma unroll
    for (int i = 0; i < CT; ++i)
      if (16 * i + g4 + 3 < a.NM) stg4(orow + 16 * i + g4, acc[0][i]);
  }
}

// dB normalisation + mel projection for a BANDED freq2mel -- what backend.mel() builds: triangular filters, each over a few
// dozen neighbouring bins, 90 % of the 513 x 80 matrix is zero.  The dense GEMM above spends 77 TFLOP/s of fp32 MFMA on
// those zeros (69 us: MFMA-bound); this kernel reads the 135 MB of log-power once (HBM-bound) and does ~1000 FMAs per frame.
// A workgroup stages MEL_FT normalised frames in LDS; thread t sums outputs t, t + 256, ... (frame-major, mel-minor).
// finalize chooses it when every filter's support is at most MEL_BW bins (a trained, dense freq2mel keeps mel_kernel).
// frames per workgroup.  Round 5: 8 instead of 16 -- 37 KB of LDS instead of 54, four resident workgroups per CU instead of two: the
// kernel's memory-level parallelism is the turnover of short-lived workgroups (43.6 -> 38.2 us at 64 x 1000 frames; 12: 39.5, 6:
// 38.2, 4: 39.5; as a LOOP over tiles with the next tile prefetched: 58.6 -- profiles/r05_stft_variants.md)
#ifndef MI355ASR_MEL_FT
#define MI355ASR_MEL_FT 8
#endif
constexpr int MEL_BW = 64, MEL_FT = MI355ASR_MEL_FT;
__global__ __launch_bounds__(256) void mel_band_kernel(MelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];    // rows [MEL_FT][RS] | weights [NM][BW] | band [NM][2]
  const int nbins = a.nbins, RS = ((nbins + 15) / 16) * 16, RS4 = RS / 4;   // staged row: the bins, padded to 16 (<= LP)
  float* rows = lds;
  float* wl = lds + MEL_FT * RS;
  int* bl = reinterpret_cast<int*>(wl + a.NM * a.BW);
  const int b = blockIdx.y, f0 = blockIdx.x * MEL_FT;
  const int nf = min(MEL_FT, a.F - f0);
  const bool norm = a.umax != nullptr;
  const float um = norm ? a.umax[b] : 0.f, floor_db = a.floor_db;
  const float* __restrict__ src = a.logp + ((size_t)b * a.F + f0) * a.LP;
  // every load of the workgroup -- rows, band weights, band table -- is requested before the first is used (a load per
  // trip would cost a memory latency each: the whole kernel): up to RMAX + WMAX float4 per thread in flight
  constexpr int RMAX = (MEL_FT * 132 + 255) / 256, WMAX = 5;            // MEL_FT rows x 132 float4 / 256 threads; 80 x 64 floats / 4 / 256
  const int NM = a.NM, BW = a.BW;
  const int total = nf * RS4, wtotal = NM * BW / 4;
  f32x4 vv[RMAX], ww[WMAX];
  int kk[RMAX];
#pragma unroll
  for (int u = 0; u < RMAX; ++u) {
    const int i = min((int)threadIdx.x + 256 * u, total - 1);
    const int fr = i / RS4;
    kk[u] = 4 * (i - fr * RS4);
    vv[u] = ldg4(src + (size_t)fr * a.LP + kk[u]);
  }
#pragma unroll
  for (int u = 0; u < WMAX; ++u) ww[u] = reinterpret_cast<const f32x4*>(a.bw)[min((int)threadIdx.x + 256 * u, wtotal - 1)];
  const int bnd = threadIdx.x < 2 * NM ? a.band[threadIdx.x] : 0;
#pragma unroll
  for (int u = 0; u < RMAX; ++u) {
    const int i = threadIdx.x + 256 * u, k = kk[u];
    if (i < total) {
      f32x4 v = vv[u];
      if (norm) { v.x = fmaxf(v.x - um, floor_db); v.y = fmaxf(v.y - um, floor_db); v.z = fmaxf(v.z - um, floor_db); v.w = fmaxf(v.w - um, floor_db); }
      // the padding bins of a row meet zero weights below: they must be finite
      v.x = k + 0 < nbins ? v.x : 0.f; v.y = k + 1 < nbins ? v.y : 0.f; v.z = k + 2 < nbins ? v.z : 0.f; v.w = k + 3 < nbins ? v.w : 0.f;
      reinterpret_cast<f32x4*>(rows)[i] = v;
    }
  }
#pragma unroll
  for (int u = 0; u < WMAX; ++u)
    if ((int)threadIdx.x + 256 * u < wtotal) reinterpret_cast<f32x4*>(wl)[threadIdx.x + 256 * u] = ww[u];
  if (threadIdx.x < 2 * NM) bl[threadIdx.x] = bnd;
  __syncthreads();
  float amax = 0.f;
  for (int o = threadIdx.x; o < nf * NM; o += 256) {
    const int fr = o / NM, m = o - fr * NM;
    const int lo = bl[2 * m], len4 = (bl[2 * m + 1] + 3) & ~3;     // weights are zero-padded to BW, rows to RS >= lo + BW
    const float* x = rows + fr * RS + lo;
    const float* w = wl + m * BW;
    float acc = 0.f;
#pragma unroll 2
    for (int j = 0; j < len4; j += 4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + j);
      acc = __builtin_fmaf(x[j + 0], w4.x, acc);
      acc = __builtin_fmaf(x[j + 1], w4.y, acc);
      acc = __builtin_fmaf(x[j + 2], w4.z, acc);
      acc = __builtin_fmaf(x[j + 3], w4.w, acc);
    }
    a.mel[((size_t)b * a.F + f0 + fr) * NM + m] = acc;
    amax = fmaxf(amax, fabsf(acc));
  }
  if (a.absmax) {
    // |x| >= 0, so the float order is the order of the bit patterns.  One candidate per workgroup, and an atomic only when it
    // beats what is already there (a plain read of the word first): thousands of workgroups hammering one address with
    // atomicMax cost 115 us of a 35 us kernel
    __shared__ float wmax[4];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned mine = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
      unsigned* word = a.absmax + blockIdx.y;           // one word per utterance: an utterance's scale is its own (batch-invariant)
      if (mine > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, mine);
    }
  }
}
int launch_mel_band(const MelArgs& a, hipStream_t s) {
  if (!a.band || !a.bw || a.BW > MEL_BW || (a.BW & 3) != 0 || (a.LP & 3) != 0) return -1;
  const int RS = ((a.nbins + 15) / 16) * 16;
  // the kernel's fixed load counts: 16 rows of RS floats in 9 float4 per thread, the band weights in 5, the table in 1
  if (RS > a.LP || MEL_FT * (RS / 4) > ((MEL_FT * 132 + 255) / 256) * 256 || a.NM * a.BW / 4 > 5 * 256 || 2 * a.NM > 256) return -1;
  const size_t lds = ((size_t)MEL_FT * RS + (size_t)a.NM * a.BW + 2 * (size_t)a.NM) * sizeof(float);
  hipLaunchKernelGGL(mel_band_kernel, dim3((a.F + MEL_FT - 1) / MEL_FT, a.B), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// dB normalisation alone: the plain Spectrogram layer (time_frequency.py:7-123 with return_decibel_spectrogram = True --
// mel_layer_type 'Spectrogram', conformer_blocks.py:318-323): out[b, f, k] = max(logp[b, f, k] - max_b, -80), k < nbins.
// Melspectrogram is this followed by freq2mel (mel_kernel above does both in one pass).
__global__ __launch_bounds__(256) void db_norm_kernel(MelArgs a) {
  const size_t n = (size_t)a.B * a.F * a.nbins;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / a.nbins;
    const int k = (int)(i - row * a.nbins);
    const int b = (int)(row / a.F);
    float x = a.logp[row * a.LP + k];
    if (a.umax) x = fmaxf(x - a.umax[b], a.floor_db);
    a.mel[row * a.NM + k] = x;
  }
}
int launch_db_norm(const MelArgs& a, hipStream_t s) {
  if (a.NM != a.nbins) return -1;
  const size_t n = (size_t)a.B * a.F * a.nbins;
  hipLaunchKernelGGL(db_norm_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, a);
  return 0;
}

int launch_mel(const MelArgs& a, hipStream_t s) {
  dim3 grid((a.B * a.FT + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  if (a.NTm == 5) hipLaunchKernelGGL((mel_kernel<5>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else if (a.NTm == 8) hipLaunchKernelGGL((mel_kernel<8>), grid, dim3(BLOCK_THREADS), 0, s, a);
  else return -1;
  return 0;
}

// Same implicit GEMM for any dmodel that is a multiple of 128 (256: ConformerM / StreamingS, 512: ConformerL), with the
// output channels split over grid.y in chunks of NBW column tiles: a wave keeps 16 positions x NBW tiles of
// accumulators (32 VGPRs) instead of the whole row, so the kernel fits 256 registers with two waves per SIMD at any
// width, and small position counts (streaming: 64 chunks x 13 x 20 positions) still give a few thousand waves.
// conv1 is recomputed per column chunk (VALU work x D/128); dmodel is a run-time value here.
// ST1 = conv1's time stride (reduction_factor / 2; conformer_blocks.py:76-80): the mel window of one conv2 output is
// (2 ST1 + 3) x 7.  ST1 = 2 is every shipped config and has the faster kernels in subconv.hip in front of this one;
// ST1 = 1, 3, 4 run here at any dmodel that is a multiple of 16 (column tiles past D / 16 are computed on tile
// D / 16 - 1 and dropped).
template <int NBW, int ST1>
__global__ __launch_bounds__(BLOCK_THREADS, 2) void subconv_split_kernel(SubConvArgs a, int D) {
  constexpr int WR = 2 * ST1 + 3;
  const int KB = D / 16, NT = D / 16;
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int c0 = blockIdx.y * NBW;
  const int P = a.B * a.T2 * a.F2;
  if ((size_t)wid * 16 >= (size_t)P) return;
  const int pos = wid * 16 + c;
  float win[WR][7];
  bool tv[3], fv[3];
  int ct[NBW];                          // column tile of accumulator n (clamped: the surplus ones are not stored)
#pragma unroll
  for (int n = 0; n < NBW; ++n) ct[n] = min(c0 + n, NT - 1);
  {
    const int p = min(pos, P - 1);
    const int b = p / (a.T2 * a.F2);
    const int r = p % (a.T2 * a.F2);
    const int t2 = r / a.F2, f2 = r % a.F2;
    const int tm0 = 2 * ST1 * t2 - ST1 * a.pt2 - a.pt1;
    const int fm0 = 4 * f2 - 2 * a.pf2 - a.pf1;
    const float* __restrict__ mb = a.mel + (size_t)b * a.F * a.NM;
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int tm = tm0 + i, fm = fm0 + j;
        win[i][j] = (tm >= 0 && tm < a.F && fm >= 0 && fm < a.NM) ? mb[(size_t)tm * a.NM + fm] : 0.f;
      }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int t1 = 2 * t2 + k - a.pt2, f1 = 2 * f2 + k - a.pf2;
      tv[k] = (t1 >= 0 && t1 < a.T1);
      fv[k] = (f1 >= 0 && f1 < a.F1);
    }
  }
  f32x4 acc[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) acc[n] = ldg4(a.b2 + 16 * ct[n] + g4);
  const f32x4* __restrict__ w2 = reinterpret_cast<const f32x4*>(a.w2p) + lane;
  f32x4 wb[2][NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) wb[0][n] = w2[(size_t)(0 * NT + ct[n]) * 64];
#pragma unroll 1
  for (int cb = 0; cb < KB; ++cb) {
    f32x4 w1v[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) w1v[i][j] = ldg4(a.w1 + (size_t)(i * 3 + j) * D + 16 * cb + g4);
    const f32x4 b1v = ldg4(a.b1 + 16 * cb + g4);
    const int cbn = (cb + 1 < KB) ? cb + 1 : cb;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int kt = q / 3, kf = q % 3;
      const int kbn = (q + 1 < 9) ? cb * 9 + q + 1 : cbn * 9;
#pragma unroll
      for (int n = 0; n < NBW; ++n) wb[(q + 1) & 1][n] = w2[(size_t)(kbn * NT + ct[n]) * 64];
      __builtin_amdgcn_sched_barrier(0);
      f32x4 v = b1v;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float m = win[ST1 * kt + i][2 * kf + j];
          v.x = __builtin_fmaf(m, w1v[i][j].x, v.x); v.y = __builtin_fmaf(m, w1v[i][j].y, v.y);
          v.z = __builtin_fmaf(m, w1v[i][j].z, v.z); v.w = __builtin_fmaf(m, w1v[i][j].w, v.w);
        }
      const bool ok = tv[kt] & fv[kf];
      v.x = ok ? fmaxf(v.x, 0.f) : 0.f; v.y = ok ? fmaxf(v.y, 0.f) : 0.f;
      v.z = ok ? fmaxf(v.z, 0.f) : 0.f; v.w = ok ? fmaxf(v.w, 0.f) : 0.f;
      mma_batch<NBW>(acc, wb[q & 1], v);
      __builtin_amdgcn_sched_barrier(0);
    }
    // nine steps per channel block: the batch prefetched last sits in wb[1]; the next block starts from wb[0]
#pragma unroll
    for (int n = 0; n < NBW; ++n) wb[0][n] = wb[1][n];
  }
  if (pos < P) {
    float* orow = a.out + (size_t)pos * D;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      if (c0 + n >= NT) break;
      f32x4 v = acc[n];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      stg4(orow + 16 * (c0 + n) + g4, v);
    }
  }
}

template <int NBW, int ST1>
static int launch_subconv_cols(int D, const SubConvArgs& a, hipStream_t s) {
  const int P = a.B * a.T2 * a.F2;
  const int tiles = (P + 15) / 16;
  hipLaunchKernelGGL((subconv_split_kernel<NBW, ST1>), dim3((tiles + 3) / 4, (D / 16 + NBW - 1) / NBW), dim3(BLOCK_THREADS), 0, s, a, D);
  return 0;
}

int launch_subconv(int D, const SubConvArgs& a, hipStream_t s) {
  if (a.st1 != 2) {                                                 // reduction_factor 2 / 6 / 8: one general kernel
    if (D % 16) return -1;
    note_scheme(SCHEME_F32);
    if (a.st1 == 1) return launch_subconv_cols<4, 1>(D, a, s);
    if (a.st1 == 3) return launch_subconv_cols<4, 3>(D, a, s);
    if (a.st1 == 4) return launch_subconv_cols<4, 4>(D, a, s);
    return -1;
  }
  // MI355ASR_SUBCONV_F32=1: the fp32-MFMA register-stream kernel instead of the split-operand one (both in subconv.hip)
  static const bool f32k = mi355_env("MI355ASR_SUBCONV_F32", 0) != 0;
  if (!f32k && launch_subconv_split(D, a, s) == 0) return 0;       // dmodel 144 / 256 / 512 with the split pack
  note_scheme(SCHEME_F32);
  if (D == 144) return launch_subconv144(a, s);
  if (D % 128 == 0) return launch_subconv_cols<8, 2>(D, a, s);
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// stream_gemm: y = x W + b with K too large to keep the input row in registers (K = F2*D = 2880).
// ---------------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(BLOCK_THREADS) void stream_gemm_kernel(StreamGemmArgs a) {
  const int lane = threadIdx.x & 63;
  const int g4 = (lane >> 4) * 4, c = lane & 15;
  const int wid = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if ((size_t)wid * 16 >= (size_t)a.M) return;
  const int tok = wid * 16 + c;
  const float* __restrict__ xr = a.x + (size_t)min(tok, a.M - 1) * a.K + g4;
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + lane;
  const int c0 = blockIdx.y * CT;        // first column tile of this wave (grid.y > 1: few rows, columns split)
  f32x4 acc[1][CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) acc[0][i] = ldg4(a.bias + 16 * (c0 + i) + g4);
  const int KBT = a.K / 16;
  auto xp = [&](int, int kb) -> f32x4 { return ldg4(xr + 16 * kb); };
  sweep_k<1, CT>(acc, wp, a.NT, c0, KBT, xp, [](int, int, f32x4 x) { return x; });
  if (tok < a.M) {
    float* orow = a.y + (size_t)tok * a.ldy;
#pragma unroll
    for (int i = 0; i < CT; ++i)
      if (16 * (c0 + i) + g4 + 3 < a.n_valid) stg4(orow + 16 * (c0 + i) + g4, acc[0][i]);
  }
}

int launch_stream_gemm(int D, const StreamGemmArgs& a, hipStream_t s) {
  const int tiles = (a.M + 15) / 16;
  if (a.K % 32 != 0) return -1;   // sweep_k consumes k-blocks in pairs
  // One wave owns 16 rows x CT column tiles over the whole K.  With few rows (streaming: 64 chunks x 13 frames =
  // 52 row tiles, K = 5120) that leaves most SIMDs idle for the length of one very long wave, so the columns are
  // split over grid.y until there is about a wave per SIMD; the rows are then re-read NT/CT times (from L2).
#define SG(CTV) do { dim3 grid((tiles + 3) / 4, a.NT / (CTV)); \
    hipLaunchKernelGGL((stream_gemm_kernel<CTV>), grid, dim3(BLOCK_THREADS), 0, s, a); return 0; } while (0)
  const int want = 1024;          // SIMDs
  if (D == 144 && a.NT == 9) {
    if (tiles >= want) SG(9);
    if (tiles * 3 >= want) SG(3);
    SG(1);
  }
  if (D == 256 && a.NT == 16) {
    if (tiles >= want) SG(16);
    if (tiles * 2 >= want) SG(8);
    if (tiles * 4 >= want) SG(4);
    if (tiles * 8 >= want) SG(2);
    SG(1);
  }
#undef SG
  return -1;
}

// ---------------------------------------------------------------------------------------------------
// CTC greedy collapse: one wave per utterance, ballot + popcount stream compaction.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void collapse_kernel(CollapseArgs a) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int T = a.T;
  int n = a.in_len ? a.in_len[b] : T;
  n = max(0, min(n, T));
  const int32_t* __restrict__ src = a.frame_ids + (size_t)b * T;
  int32_t* dst = a.ids + (size_t)b * T;
  int count = 0;
  int carry = -1;
  for (int t0 = 0; t0 < n; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < n;
    const int id = valid ? src[t] : -2;
    int prev = __shfl_up(id, 1);
    if (lane == 0) prev = carry;
    const bool keep = valid && id != a.blank && id != prev;
    const unsigned long long mask = __ballot(keep);
    const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) dst[pos] = id;
    count += __popcll(mask);
    carry = __shfl(id, 63);
  }
  for (int t = count + lane; t < T; t += 64) dst[t] = -1;
  if (lane == 0) a.out_len[b] = count;
}

int launch_collapse(const CollapseArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(collapse_kernel, dim3(a.B), dim3(64), 0, s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// feature_pick (chunk_conformer_blocks.py:913-999): keep the frames whose phone argmax is not the blank,
// compacted per utterance and zero-padded to the batch maximum.  pick_kernel = stream compaction of the
// frame indices (one wave per utterance, ballot + popcount); gather_kernel = row gather.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pick_kernel(PickArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  // no __restrict__: mi355asr_feature_pick_count runs this in place (idx == frame_ids).  That is safe by construction -- a
  // 64-frame group is read before its ballot, and kept frames land at or before positions already read -- but only as long
  // as the compiler may not assume the two pointers are distinct
  const int32_t* src = a.frame_ids + (size_t)b * a.T;
  int32_t* dst = a.idx + (size_t)b * a.T;
  int count = 0;
  for (int t0 = 0; t0 < a.T; t0 += 64) {
    const int t = t0 + lane;
    const bool keep = t < a.T && src[t] != a.blank;
    const unsigned long long mask = __ballot(keep);
    if (keep) dst[count + __popcll(mask & ((1ull << lane) - 1ull))] = t;
    count += __popcll(mask);
  }
  if (lane == 0) a.cnt[b] = count;
}

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
  const int d4n = a.D / 4;
  const size_t total = (size_t)a.B * a.Tp * d4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4n) * 4;
    const int j = (int)((i / d4n) % a.Tp);
    const int b = (int)(i / ((size_t)d4n * a.Tp));
    f32x4 v = splat4(0.f);
    if (j < a.cnt[b]) v = ldg4(a.src + ((size_t)b * a.T + a.idx[(size_t)b * a.T + j]) * a.D + c4);
    stg4(a.dst + ((size_t)b * a.Tp + j) * a.D + c4, v);
  }
}

// per-row argmax of [M, V] logits, first maximum wins (tf.argmax): one wave per row
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ x, int32_t* __restrict__ out, int M, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* __restrict__ r = x + (size_t)row * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < V; i += 64) {
    const float v = r[i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;     // all-NaN row: class 0
}
int launch_row_argmax(const float* x, int32_t* out, int M, int V, hipStream_t s) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(row_argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, out, M, V);
  return 0;
}

int launch_pick(const PickArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(pick_kernel, dim3(a.B), dim3(64), 0, s, a);
  return 0;
}
// rows whose width is not a multiple of 4 floats (e.g. the picker's 277 classes): element-wise
__global__ __launch_bounds__(256) void gather1_kernel(GatherArgs a) {
  const size_t total = (size_t)a.B * a.Tp * a.D;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % a.D);
    const int j = (int)((i / a.D) % a.Tp);
    const int b = (int)(i / ((size_t)a.D * a.Tp));
    a.dst[i] = j < a.cnt[b] ? a.src[((size_t)b * a.T + a.idx[(size_t)b * a.T + j]) * a.D + c] : 0.f;
  }
}

int launch_gather(const GatherArgs& a, hipStream_t s) {
  if (a.D % 4) {
    const size_t n = (size_t)a.B * a.Tp * a.D;
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather1_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, a);
    return 0;
  }
  const size_t total = (size_t)a.B * a.Tp * (a.D / 4);
  if (total == 0) return 0;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Translator input side (conformer_blocks.py:537-539, 455-457): Embedding lookup of the phoneme ids, and the
// sinusoidal positional term added to the *query* input of the cross-attention (positional_encoding.py:19-53;
// table computed on the host in double precision, [max_len, D]).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
  const int d4n = a.D / 4;
  const size_t total = (size_t)a.M * d4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4n) * 4;
    const size_t row = i / d4n;
    const int id = min(max(a.ids[row], 0), a.V - 1);
    stg4(a.dst + row * a.D + c4, ldg4(a.table + (size_t)id * a.D + c4));
  }
}
__global__ __launch_bounds__(256) void add_pe_kernel(AddPeArgs a) {
  const int d4n = a.D / 4;
  const size_t total = (size_t)a.B * a.U * d4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4n) * 4;
    const size_t row = i / d4n;
    const int u = (int)(row % a.U);
    stg4(a.dst + row * a.D + c4, ldg4(a.src + row * a.D + c4) + ldg4(a.pe + (size_t)u * a.D + c4));
  }
}
int launch_embed(const EmbedArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.M * (a.D / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(embed_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, a);
  return 0;
}
int launch_add_pe(const AddPeArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * a.U * (a.D / 4);
  if (total == 0) return 0;
  hipLaunchKernelGGL(add_pe_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, a);
  return 0;
}
