// rfx_gl.hip - Griffin-Lim phase reconstruction on gfx950 (replaces torchaudio.transforms.GriffinLim
// as constructed at riffusion/spectrogram_converter.py:62-73 and called at :204).
//
// One launch = one Griffin-Lim iteration over every frame of every clip-channel:
//   MODE 0 (init)  : Z = |S| * angles0                          -> ISTFT -> audio_0
//   MODE 1 (first) : rebuilt = STFT(audio_0); tprev == 0         -> update -> ISTFT -> audio_1
//   MODE 2 (iter)  : rebuilt = STFT(audio_k); momentum update    -> ISTFT -> audio_{k+1}
// A workgroup walks a run of consecutive frames of one clip-channel, so that the 10-way overlap-add
// of torch.istft is a register sliding window (thread n' owns sample n' of every hop block) and the
// only cross-workgroup traffic is the 9-block halo at each end of a run.  Halo blocks are never
// combined with atomics: run r writes its partial sums to the parity-(r&1) audio buffer and the
// reader adds the two parity buffers, which keeps results bit-reproducible run to run.
//
// HBM traffic per frame and iteration: |S| 4 B + tprev 8 B read + 8 B written per slot (9261 slots
// for 8821 bins) = the (20n+4)*F*T formulation of SURVEY.md 8(d); `angles` never exists in memory.
#include "rfx_frame.hip.h"
#include "rfx_kernels.h"

namespace rfx {

template <int MODE>
__global__ void __launch_bounds__(kThreads) gl_iter_kernel(GlArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* cube = reinterpret_cast<cf*>(smem);

  const ThreadId t = thread_id();
  ThreadConst c;
  load_thread_const(c, t, g.tw1, g.tw2, g.win);

  const int clip = blockIdx.x / g.nruns;
  const int run = blockIdx.x - clip * g.nruns;
  const int t0 = (int)(((long long)run * g.T) / g.nruns);
  const int t1 = (int)(((long long)(run + 1) * g.T) / g.nruns) - 1;
  const int par = run & 1;
  const int nblk = g.T - 1;  // hop blocks kept by istft's centre trim

  const float* __restrict__ in0 = g.audio_in[0] + (size_t)clip * g.Lpad;
  const float* __restrict__ in1 = g.audio_in[1] + (size_t)clip * g.Lpad;
  float* __restrict__ outA = g.audio_out[par] + (size_t)clip * g.Lpad;
  float* __restrict__ outB = g.audio_out[par ^ 1] + (size_t)clip * g.Lpad;

  float acc[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) acc[j] = 0.f;

  auto emit = [&](int blk, float val) {
    if (blk < 0 || blk >= nblk || !t.active) return;
    const int p = blk * kHop + t.npr;
    const bool full = (max(blk - 4, 0) >= t0) && (min(blk + 5, g.T - 1) <= t1);
    outA[p] = val * g.out_scale[p];
    if (full) outB[p] = 0.f;
  };

  for (int fr = t0; fr <= t1; ++fr) {
    const size_t fbase = ((size_t)clip * g.T + fr) * kFrameStride;
    const int q = t.npr;

    // ---- issue the streaming loads of this frame first: |S| (4 kb per 16 B) and tprev (2 kb per 16 B)
    float Sm[21];
    {
      const v4f* s4 = reinterpret_cast<const v4f*>(g.S + fbase);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        v4f v = __builtin_nontemporal_load(s4 + i * kHop + q);
        Sm[4 * i] = v.x; Sm[4 * i + 1] = v.y; Sm[4 * i + 2] = v.z; Sm[4 * i + 3] = v.w;
      }
      Sm[20] = __builtin_nontemporal_load(g.S + fbase + 20 * kHop + q);
    }
    cf tp[21];
    if (MODE != 1) {
      const cf* src = (MODE == 0) ? g.angles0 : g.tprev;
      if (MODE == 2 || src != nullptr) {
        const v4f* t4 = reinterpret_cast<const v4f*>(src + fbase);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          v4f v = __builtin_nontemporal_load(t4 + i * kHop + q);
          tp[2 * i] = cf{v.x, v.y};
          tp[2 * i + 1] = cf{v.z, v.w};
        }
        const v2f* t2 = reinterpret_cast<const v2f*>(src + fbase);
        v2f w = __builtin_nontemporal_load(t2 + 20 * kHop + q);
        tp[20] = cf{w.x, w.y};
      } else {
        // rand_init=True (spectrogram_converter.py:72): U[0,1) real and imaginary parts per BIN, so
        // a conjugate slot draws the same pair as its primary and conjugates it
#pragma unroll
        for (int kb = 0; kb < 21; ++kb) {
          bool cj;
          const int bin = slot_bin(t.k1, t.idx, kb, &cj);
          cf r = rand_unit_pair(g.seed, ((unsigned long long)clip * g.T + fr) * kBins + bin);
          tp[kb] = cf{r.re, cj ? -r.im : r.im};
        }
      }
    }

    cf Z[21];
    if (MODE == 0) {
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) Z[kb] = cf{Sm[kb] * tp[kb].re, Sm[kb] * tp[kb].im};
    } else {
      // ---- analysis: reflect-padded, Hann-windowed frame centred on sample 441*fr
      float u[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int p = reflect_index((fr + j - kHalfHops) * kHop + t.npr, g.L);
        u[j] = (in0[p] + in1[p]) * c.win[j];
      }
      cf R[21];
      frame_forward(u, R, cube, t, c);

      // ---- momentum phase update; tprev <- rebuilt
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) {
        const cf prev = (MODE == 1) ? cf{0.f, 0.f} : tp[kb];
        Z[kb] = gl_update(R[kb], prev, (MODE == 1) ? 0.f : g.mom, Sm[kb]);
      }
      if (t.active) {
        v4f* t4 = reinterpret_cast<v4f*>(g.tprev + fbase);
#pragma unroll
        for (int i = 0; i < 10; ++i)
          __builtin_nontemporal_store(v4f{R[2 * i].re, R[2 * i].im, R[2 * i + 1].re, R[2 * i + 1].im},
                                      t4 + i * kHop + q);
        v2f* t2 = reinterpret_cast<v2f*>(g.tprev + fbase);
        __builtin_nontemporal_store(v2f{R[20].re, R[20].im}, t2 + 20 * kHop + q);
      }
    }

    // ---- synthesis: inverse transform, synthesis window, overlap-add into the sliding window
    float y[10];
    frame_inverse(Z, y, cube, t, c);
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = fmaf(y[j], c.win[j], acc[j]);
    emit(fr - kHalfHops, acc[0]);
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = acc[j + 1];
    acc[9] = 0.f;
    // MODE 0 has no analysis half: its next P3' store would overwrite rows whose columns other
    // waves are still gathering in P1'
    if (MODE == 0) __syncthreads();
  }
  // ---- flush the right halo of the run
#pragma unroll
  for (int j = 0; j < 9; ++j) emit(t1 - 4 + j, acc[j]);
}

// wave[b][p] = A0 + A1 : fold the two parity buffers into the caller's (B, L) tensor
__global__ void gl_combine_kernel(const float* a0, const float* a1, float* out, int L, int Lpad, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const size_t b = i / L, p = i - b * L;
    out[i] = a0[b * Lpad + p] + a1[b * Lpad + p];
  }
}

hipError_t launch_gl_iter(int mode, const GlArgs& g, int nblocks, hipStream_t stream) {
  const size_t lds = sizeof(cf) * kSlots;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gl_iter_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gl_iter_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gl_iter_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  switch (mode) {
    case 0: hipLaunchKernelGGL(gl_iter_kernel<0>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
    case 1: hipLaunchKernelGGL(gl_iter_kernel<1>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
    default: hipLaunchKernelGGL(gl_iter_kernel<2>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
  }
  return hipGetLastError();
}

hipError_t launch_gl_combine(const float* a0, const float* a1, float* out, int B, int L, int Lpad, hipStream_t stream) {
  const size_t total = (size_t)B * L;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gl_combine_kernel, dim3(blocks), dim3(256), 0, stream, a0, a1, out, L, Lpad, total);
  return hipGetLastError();
}

}  // namespace rfx
