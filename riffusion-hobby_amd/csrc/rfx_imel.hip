// rfx_imel.hip - InverseMelScale on gfx950 (replaces torchaudio 0.13 transforms.InverseMelScale as
// constructed at riffusion/spectrogram_converter.py:87-99 and called at :201).
//
// The reference minimises  mean_{c,t} sum_m (mel - spec @ fb)^2  with torch.optim.SGD(lr 0.1,
// momentum 0.9) from a uniform random start, clamping at zero after every step, for max_iter steps
// (an early exit on the clip loss practically never fires at spectrogram scales).  Frames only
// interact through the 1/(C*T) factor of the mean and through that early exit, and the HTK
// filterbank is banded (<= 2 adjacent mel filters per linear bin), so each frame's whole
// optimisation runs inside one workgroup with all state on chip:
//   phase A  (thread per mel)   pred_m = sum_{f in band(m)} w * spec_f      spec, w in LDS
//                               diff_m = mel_m - pred_m                     -> LDS, sum diff^2 -> history
//   phase B  (thread per bin)   g = -(2/(C*T)) (diff_m0 w0 + diff_m0+1 w1); buf = mom*buf + g;
//                               spec = max(0, spec - lr*buf)                spec, buf, w in registers
// Bins whose filterbank row is zero never move: they pass their initial value through, exactly like
// the reference.  The per-frame loss history lets a follow-up scan reproduce the reference's early
// exit (it_stop per clip) and a fix-up launch re-runs the affected clips with that step count.
#include <hip/hip_runtime.h>

#include "rfx_core.h"
#include "rfx_kernels.h"

namespace rfx {

constexpr int kImelThreads = 256;

RFX_HD float rand_unit(unsigned long long seed, unsigned long long ctr) {
  unsigned long long z = ctr * 0x9E3779B97F4A7C15ull + seed + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)((unsigned)(z >> 40)) * (1.0f / 16777216.0f);
}

template <int BPT>  // bins per thread
__global__ void __launch_bounds__(kImelThreads) imel_kernel(ImelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ImelTables& tb = a.tb;
  const int nb = tb.f_hi - tb.f_lo;
  float* spec_s = reinterpret_cast<float*>(smem);            // [nb]
  float* w_s = spec_s + ((nb + 3) & ~3);                     // [nnz]
  float* diff_s = w_s + ((tb.nnz + 3) & ~3);                 // [M + 1] (one pad entry for m0+1 == M)
  float* red_s = diff_s + ((a.M + 1 + 3) & ~3);              // [4] wave partials of sum diff^2
  float* hist_s = red_s + 4;                                 // [max_iter]

  const int frame = blockIdx.x;  // = b*T + t
  const int b = frame / a.T, t = frame - b * a.T;
  const int clip = b / a.C;
  const int tid = threadIdx.x;
  const int steps = a.it_limit ? a.it_limit[clip] : a.max_iter;
  if (a.it_limit && steps >= a.max_iter) return;  // fix-up pass: this clip never stopped early
  const int n_stft = kBins;

  for (int i = tid; i < tb.nnz; i += kImelThreads) w_s[i] = tb.csr_w[i];
  if (tid == 0) diff_s[a.M] = 0.f;

  // ---- phase-B ownership: bins f = f_lo + tid + 256*j
  float spec[BPT], buf[BPT], w0[BPT], w1[BPT];
  int m0[BPT];
  const unsigned long long rbase = (unsigned long long)frame * n_stft;
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    const int f = tb.f_lo + tid + kImelThreads * j;
    const bool ok = f < tb.f_hi;
    m0[j] = ok ? tb.bin_m0[f] : -1;
    w0[j] = ok ? tb.bin_w0[f] : 0.f;
    w1[j] = ok ? tb.bin_w1[f] : 0.f;
    spec[j] = ok ? (a.spec0 ? a.spec0[(size_t)frame * n_stft + f] : rand_unit(a.seed, rbase + f)) : 0.f;
    buf[j] = 0.f;
    if (m0[j] < 0) { m0[j] = a.M; w0[j] = 0.f; w1[j] = 0.f; }  // a zero row inside the range: reads the pad, never moves
    if (ok) spec_s[f - tb.f_lo] = spec[j];
  }
  // ---- phase-A ownership: even r counts mels up from 0, odd r counts down from M-1, so every
  // thread pairs a short low-frequency band with a long high-frequency one
  constexpr int kMaxMelPerThread = 4;  // M <= 1024
  const int n_rounds = (a.M + kImelThreads - 1) / kImelThreads;
  const int up_bound = min(a.M, kImelThreads * ((n_rounds + 1) / 2));
  float melv[kMaxMelPerThread];
  int mlist[kMaxMelPerThread];
#pragma unroll
  for (int r = 0; r < kMaxMelPerThread; ++r) {
    int m = -1;
    if (r < n_rounds) {
      if ((r & 1) == 0) {
        m = (r / 2) * kImelThreads + tid;
        if (m >= up_bound) m = -1;
      } else {
        m = a.M - 1 - (r / 2) * kImelThreads - tid;
        if (m < up_bound) m = -1;
      }
    }
    mlist[r] = m;
    melv[r] = m >= 0 ? a.mel[((size_t)b * a.M + m) * a.T + t] : 0.f;
  }
  const float gscale = -2.0f / (float)(a.C * a.T);
  __syncthreads();

  for (int it = 0; it < steps; ++it) {
    // ---- phase A
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < kMaxMelPerThread; ++r) {
      const int m = mlist[r];
      if (m < 0) continue;
      const int p0 = tb.csr_ptr[m], p1 = tb.csr_ptr[m + 1];
      const float* sp = spec_s + (tb.band_lo[m] - tb.f_lo);
      float acc = 0.f;
      for (int p = p0; p < p1; ++p) acc = fmaf(w_s[p], sp[p - p0], acc);
      const float d = melv[r] - acc;
      diff_s[m] = d;
      sq = fmaf(d, d, sq);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red_s[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) hist_s[it] = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    // ---- phase B
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const float g = gscale * fmaf(diff_s[m0[j]], w0[j], diff_s[min(m0[j] + 1, a.M)] * w1[j]);
      buf[j] = (it == 0) ? g : fmaf(a.momentum, buf[j], g);
      spec[j] = fmaxf(0.f, fmaf(-a.lr, buf[j], spec[j]));
      const int f = tb.f_lo + tid + kImelThreads * j;
      if (f < tb.f_hi) spec_s[f - tb.f_lo] = spec[j];
    }
    __syncthreads();
  }

  // ---- results: moved bins from registers, untouched bins straight from the init
  float* out = a.out_slots + (size_t)frame * kFrameStride;
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    const int f = tb.f_lo + tid + kImelThreads * j;
    if (f < tb.f_hi) {
      out[tb.bin_pos[f]] = spec[j];
      const int p2 = tb.bin_pos2[f];
      if (p2 >= 0) out[p2] = spec[j];
    }
  }
  for (int f = tid; f < n_stft; f += kImelThreads) {
    if (f >= tb.f_lo && f < tb.f_hi) continue;
    const float v = a.spec0 ? a.spec0[(size_t)frame * n_stft + f] : rand_unit(a.seed, rbase + f);
    out[tb.bin_pos[f]] = v;
    const int p2 = tb.bin_pos2[f];
    if (p2 >= 0) out[p2] = v;
  }
  // padding lanes of the slot layout are zeroed so that later consumers never see garbage
  for (int p = tid; p < kFrameStride; p += kImelThreads) {
    int q, kb;
    if (!pos_f_to_slot(p, q, kb)) out[p] = 0.f;
  }
  if (a.loss_hist && !a.it_limit)
    for (int i = tid; i < a.max_iter; i += kImelThreads) a.loss_hist[(size_t)frame * a.max_iter + i] = i < steps ? hist_s[i] : 0.f;
}

// one thread per clip: replays the reference's stopping rule on the clip-mean loss
__global__ void imel_scan_kernel(const float* __restrict__ loss_hist, int* __restrict__ it_stop, int* __restrict__ any_early,
                                 int nclips, int C, int T, int max_iter, float tol_loss, float tol_change) {
  const int clip = blockIdx.x;
  if (clip >= nclips) return;
  __shared__ float red[256];
  const int nframes = C * T;
  float prev = __builtin_inff();
  int stop = max_iter;
  for (int it = 0; it < max_iter; ++it) {
    float s = 0.f;
    for (int f = threadIdx.x; f < nframes; f += blockDim.x) s += loss_hist[((size_t)clip * nframes + f) * max_iter + it];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    const float loss = red[0] / (float)nframes;
    __syncthreads();
    if (loss < tol_loss || fabsf(prev - loss) < tol_change) { stop = it + 1; break; }
    prev = loss;
  }
  if (threadIdx.x == 0) {
    it_stop[clip] = stop;
    if (stop < max_iter) atomicExch(any_early, 1);
  }
}

hipError_t launch_imel(const ImelArgs& a, hipStream_t stream) {
  const int nb = a.tb.f_hi - a.tb.f_lo;
  const size_t lds = sizeof(float) * (((nb + 3) & ~3) + ((a.tb.nnz + 3) & ~3) + ((a.M + 1 + 3) & ~3) + 4 + a.max_iter);
  const int bpt = (nb + kImelThreads - 1) / kImelThreads;
  if (bpt <= 16) {
    (void)hipFuncSetAttribute((const void*)imel_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(imel_kernel<16>, dim3(a.B * a.T), dim3(kImelThreads), lds, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)imel_kernel<36>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(imel_kernel<36>, dim3(a.B * a.T), dim3(kImelThreads), lds, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_imel_scan(const float* loss_hist, int* it_stop, int* any_early, int nclips, int C, int T, int max_iter,
                            float tol_loss, float tol_change, hipStream_t stream) {
  hipLaunchKernelGGL(imel_scan_kernel, dim3(nclips), dim3(256), 0, stream, loss_hist, it_stop, any_early, nclips, C, T,
                     max_iter, tol_loss, tol_change);
  return hipGetLastError();
}

}  // namespace rfx
