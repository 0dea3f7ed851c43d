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
#include <stdlib.h>

namespace rfx {

#ifndef RFX_MIN_WAVES
#define RFX_MIN_WAVES 4
#endif

// descriptors of the per-clip streams (wave-uniform, live in SGPRs)
struct GlStreams {
  rsrc_t S, tprev, tprev_out, init;  // this clip's slot-major frames (tprev is read, tprev_out written)
  bool have_init;
};

#ifndef RFX_STREAM_AUX
#define RFX_STREAM_AUX 2  // gfx950 'nt'
#endif
#ifndef RFX_STORE_AUX
#define RFX_STORE_AUX RFX_STREAM_AUX
#endif
// The per-bin update streams |S| (4 B) and tprev (8 B) per slot from HBM.  A thread's 21 slots are
// fetched in three stages whose loads are put in flight well ahead of their use (stage A under
// P2/P3, stage B under A's arithmetic, stage C under B's), each stage costing 24 / 24 / 15 VGPRs:
//   A: kb 0..7   B: kb 8..15   C: kb 16..20
template <int KB0, int N>
struct Stage {
  static constexpr int kNS4 = N / 4, kNT4 = N / 2;  // 16-B loads of |S| and of tprev
  v4f s4[kNS4 > 0 ? kNS4 : 1];
  v4f t4[kNT4 > 0 ? kNT4 : 1];
  float s_tail;  // kb 20 (only when KB0 + N == 21)
  v2f t_tail;
};

template <int MODE, int KB0, int N>
__device__ __forceinline__ void stage_issue(Stage<KB0, N>& g, const GlStreams& st, unsigned foff, unsigned q16) {
  constexpr int NB = (KB0 + N == 21) ? N - 1 : N;  // kb handled by 16-B loads
#pragma unroll
  for (int i = 0; i < NB / 4; ++i) g.s4[i] = ld4<RFX_STREAM_AUX>(st.S, q16, foff + (unsigned)(KB0 / 4 + i) * (kQPad * 16u));
  if (KB0 + N == 21) g.s_tail = ld1<RFX_STREAM_AUX>(st.S, q16 >> 2, foff + 20u * kQPad * 4u);
  if (MODE == 2 || (MODE == 0 && st.have_init)) {
    const rsrc_t src = (MODE == 0) ? st.init : st.tprev;
#pragma unroll
    for (int i = 0; i < NB / 2; ++i) g.t4[i] = ld4<RFX_STREAM_AUX>(src, q16, 2u * foff + (unsigned)(KB0 / 2 + i) * (kQPad * 16u));
    if (KB0 + N == 21) g.t_tail = ld2<RFX_STREAM_AUX>(src, q16 >> 1, 2u * foff + 20u * kQPad * 8u);
  }
}

// tprev_out <- rebuilt for the whole thread (11 stores).  Issued in ONE burst right after the first
// stage's loads: gfx950's vmcnt retires loads and stores in issue order, so a store issued between
// two load stages would put its full HBM write latency in front of the second stage's data.  The
// burst precedes the later stages' tprev loads, hence the ping-pong: it never writes the buffer
// this launch reads.
__device__ __forceinline__ void store_rebuilt(const cf (&R)[21], const GlStreams& st, unsigned foff, unsigned q16) {
#pragma unroll
  for (int i = 0; i < 10; ++i)
    st4<RFX_STORE_AUX>(v4f{R[2 * i].re, R[2 * i].im, R[2 * i + 1].re, R[2 * i + 1].im}, st.tprev_out, q16,
                       2u * foff + (unsigned)i * (kQPad * 16u));
  st2<RFX_STORE_AUX>(v2f{R[20].re, R[20].im}, st.tprev_out, q16 >> 1, 2u * foff + 20u * kQPad * 8u);
}

template <int MODE, int KB0, int N>
__device__ __forceinline__ void stage_apply(cf (&R)[21], const Stage<KB0, N>& g, const GlStreams& st, float mom,
                                            unsigned long long seed, unsigned long long rng_base, int k1, int ka) {
  constexpr int NB = (KB0 + N == 21) ? N - 1 : N;
  float Sm[N];
  cf tp[N];
#pragma unroll
  for (int i = 0; i < NB; ++i) Sm[i] = g.s4[i / 4][i % 4];
  if (KB0 + N == 21) Sm[N - 1] = g.s_tail;
  if (MODE == 2 || (MODE == 0 && st.have_init)) {
#pragma unroll
    for (int i = 0; i < NB; ++i) tp[i] = cf{g.t4[i / 2][2 * (i % 2)], g.t4[i / 2][2 * (i % 2) + 1]};
    if (KB0 + N == 21) tp[N - 1] = cf{g.t_tail.x, g.t_tail.y};
  } else if (MODE == 0) {
    // rand_init=True (spectrogram_converter.py:72): U[0,1) real and imaginary parts per BIN, so a
    // conjugate slot draws the same pair as its primary and conjugates it
#pragma unroll
    for (int i = 0; i < N; ++i) {
      bool cj;
      const int bin = slot_bin(k1, ka, KB0 + i, &cj);
      cf r = rand_unit_pair(seed, rng_base + bin);
      tp[i] = cf{r.re, cj ? -r.im : r.im};
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) R[KB0 + i] = cf{Sm[i] * tp[i].re, Sm[i] * tp[i].im};
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const cf prev = (MODE == 1) ? cf{0.f, 0.f} : tp[i];
      R[KB0 + i] = gl_update(R[KB0 + i], prev, (MODE == 1) ? 0.f : mom, Sm[i]);
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(kThreads, RFX_MIN_WAVES) gl_iter_kernel(GlArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, g.tw1, g.tw2);

  const int clip = blockIdx.x / g.nruns;
  const int run = blockIdx.x - clip * g.nruns;
  const int t0 = (int)(((long long)run * g.T) / g.nruns);
  const int t1 = (int)(((long long)(run + 1) * g.T) / g.nruns) - 1;
  const int par = run & 1;
  const int nblk = g.T - 1;  // hop blocks kept by istft's centre trim

  const size_t clip_slots = (size_t)g.T * kFrameStride;
  GlStreams st;
  st.S = make_rsrc(g.S + clip * clip_slots, clip_slots * sizeof(float));
  st.tprev = make_rsrc(g.tprev_in + clip * clip_slots, clip_slots * sizeof(cf));
  st.tprev_out = make_rsrc(g.tprev_out + clip * clip_slots, clip_slots * sizeof(cf));
  st.have_init = g.angles0 != nullptr;
  st.init = make_rsrc(st.have_init ? g.angles0 + clip * clip_slots : g.tprev_in, clip_slots * sizeof(cf));
  const rsrc_t in0 = make_rsrc(g.audio_in[0] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t in1 = make_rsrc(g.audio_in[1] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t outA = make_rsrc(g.audio_out[par] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t outB = make_rsrc(g.audio_out[par ^ 1] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t scl = make_rsrc(g.out_scale, (size_t)g.L * 4);
  const rsrc_t win = make_rsrc(g.win, kWin * 4);
  const unsigned npr4 = (unsigned)t.npr * 4u;

  float acc[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) acc[j] = 0.f;

  auto emit = [&](int blk, float val) {
    if (blk < 0 || blk >= nblk || !t.active) return;  // blk is wave-uniform
    const unsigned boff = (unsigned)blk * (kHop * 4u);
    const bool full = (max(blk - 4, 0) >= t0) && (min(blk + 5, g.T - 1) <= t1);
    st1(val * ld1(scl, npr4, boff), outA, npr4, boff);
    if (full) st1(0.f, outB, npr4, boff);
  };
  __syncthreads();  // tw2 table in LDS

#ifdef RFX_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = wall_clock64();
#define RFX_STAMP(i) do { unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define RFX_STAMP(i) ((void)0)
#endif
  for (int fr = t0; fr <= t1; ++fr) {
    const unsigned foff = (unsigned)fr * (kFrameStride * 4u);
    const unsigned q16 = threadIdx.x * 16u;  // byte offset of this thread's 16-B slot pairs/quads in the streams
    const unsigned long long rng_base = ((unsigned long long)clip * g.T + fr) * kBins;

    cf R[21];
#ifndef RFX_SPLIT_A
#define RFX_SPLIT_A 8
#endif
#ifndef RFX_SPLIT_B
#define RFX_SPLIT_B 8
#endif
#ifndef RFX_EARLY_ISSUE
#define RFX_EARLY_ISSUE 1  // stage A goes in flight right after the analysis barrier, under P2/P3
#endif
#ifndef RFX_STREAM_AUX
#define RFX_STREAM_AUX 2
#endif
    Stage<0, RFX_SPLIT_A> sa;
    Stage<RFX_SPLIT_A, RFX_SPLIT_B> sb;
    Stage<RFX_SPLIT_A + RFX_SPLIT_B, 21 - RFX_SPLIT_A - RFX_SPLIT_B> sc;
    auto issue_a = [&] {
#if !defined(RFX_ABL_NOMEM) && RFX_EARLY_ISSUE
      stage_issue<MODE>(sa, st, foff, q16);
#endif
    };
    if (MODE != 0) {
      // ---- analysis: reflect-padded, Hann-windowed frame centred on sample 441*fr
      float u[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const unsigned p4 = (unsigned)reflect_index((fr + j - kHalfHops) * kHop + t.npr, g.L) * 4u;
        u[j] = (ld1(in0, p4, 0) + ld1(in1, p4, 0)) * ld1(win, npr4, (unsigned)j * (kHop * 4u));
      }
#ifdef RFX_ABL_NOFFT
      issue_a();
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) R[kb] = cf{u[kb % 10], u[(kb + 3) % 10]};
#else
      frame_forward(u, R, f, t, [&] { RFX_STAMP(1); issue_a(); }, [&] { RFX_STAMP(0); });
      RFX_STAMP(2);
#endif
    } else {
      issue_a();
    }
    // ---- momentum phase update; R becomes the next spectrum estimate.  Order of VMEM issue:
    // stage A loads, ALL tprev stores, stage B loads | apply A | stage C loads | apply B | apply C
#ifdef RFX_ABL_NOMEM
#pragma unroll
    for (int i = 0; i < 2; ++i) { sa.s4[i] = v4f{1.f, 2.f, 3.f, 4.f}; sb.s4[i] = v4f{1.f, 2.f, 3.f, 4.f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { sa.t4[i] = v4f{.1f, .2f, .3f, .4f}; sb.t4[i] = v4f{.1f, .2f, .3f, .4f}; }
    sc.s4[0] = v4f{1.f, 2.f, 3.f, 4.f}; sc.t4[0] = sc.t4[1] = v4f{.1f, .2f, .3f, .4f}; sc.s_tail = 2.f; sc.t_tail = v2f{.3f, .1f};
#else
#if !RFX_EARLY_ISSUE
    stage_issue<MODE>(sa, st, foff, q16);
#endif
    if (MODE != 0 && t.active) store_rebuilt(R, st, foff, q16);
    stage_issue<MODE>(sb, st, foff, q16);
    RFX_SCHED_FENCE();
#endif
    stage_apply<MODE>(R, sa, st, g.mom, g.seed, rng_base, t.k1, t.idx);
    RFX_SCHED_FENCE();
#ifndef RFX_ABL_NOMEM
    stage_issue<MODE>(sc, st, foff, q16);
    RFX_SCHED_FENCE();
#endif
    stage_apply<MODE>(R, sb, st, g.mom, g.seed, rng_base, t.k1, t.idx);
    RFX_SCHED_FENCE();
    stage_apply<MODE>(R, sc, st, g.mom, g.seed, rng_base, t.k1, t.idx);
    RFX_SCHED_FENCE();
    RFX_STAMP(3);

    // ---- synthesis: inverse transform, synthesis window, overlap-add into the sliding window
    float y[10];
#ifdef RFX_ABL_NOFFT
#pragma unroll
    for (int j = 0; j < 10; ++j) y[j] = R[j].re + R[j + 10].im + R[20].re;
#else
    frame_inverse(R, y, f, t, [&] { RFX_STAMP(4); }, [&] { RFX_STAMP(5); });
#endif
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = fmaf(y[j], ld1(win, npr4, (unsigned)j * (kHop * 4u)), acc[j]);
    emit(fr - kHalfHops, acc[0]);
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = acc[j + 1];
    acc[9] = 0.f;
    // MODE 0 has no analysis half: its next P3' store would overwrite rows whose columns other
    // waves are still gathering in P1'
    if (MODE == 0) __syncthreads();
    RFX_STAMP(6);
  }
#ifdef RFX_TIMING
  if (g.timing && (threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    for (int i = 0; i < 8; ++i) g.timing[((size_t)blockIdx.x * 7 + w) * 8 + i] = tacc[i];
  }
#endif
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
  const size_t lds = kFrameLdsBytes;
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

int gl_blocks_per_cu() {
  static int cached = 0;
  if (cached) return cached;
  if (const char* e = getenv("RFX_GL_WGS_PER_CU")) {
    cached = atoi(e) > 0 ? atoi(e) : 1;
    return cached;
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gl_iter_kernel<2>, kThreads, kFrameLdsBytes) != hipSuccess || n < 1) n = 1;
  cached = n;
  return cached;
}

hipError_t launch_gl_combine(const float* a0, const float* a1, float* out, int B, int L, int Lpad, hipStream_t stream) {
  const size_t total = (size_t)B * L;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gl_combine_kernel, dim3(blocks), dim3(256), 0, stream, a0, a1, out, L, Lpad, total);
  return hipGetLastError();
}

}  // namespace rfx
