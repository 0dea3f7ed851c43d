// rfx_gl.hip - Griffin-Lim phase reconstruction on gfx950 (replaces torchaudio.transforms.GriffinLim
// as constructed at riffusion/spectrogram_converter.py:62-73 and called at :204).
//
// The reference iterates (torchaudio functional.griffinlim)
//     x_k     = ISTFT(|S| * angles_k)
//     rebuilt = STFT(x_k)
//     angles_{k+1} = normalise(rebuilt - m * tprev),   tprev <- rebuilt,   m = 0.99 / 1.99
// and, executed op by op, streams `rebuilt`/`tprev` (8 B per bin each way) and |S| through memory on
// every iteration: 20 B per bin and iteration in the canonical fused form.
// The STFT is LINEAR and tprev is just the STFT of the previous estimate, so
//     rebuilt - m * tprev  =  STFT(x_k) - m * STFT(x_{k-1})  =  STFT(x_k - m * x_{k-1}).
// This kernel therefore analyses the time-domain combination x_k - m*x_{k-1} (two 0.9 MB audio
// buffers per clip, L2-resident) and never materialises `rebuilt` or `tprev`: per bin and iteration
// only |S| (4 B) is read from HBM, with exactly the same number of transforms.  The result differs
// from the op-by-op order by fp32 rounding only (parity tests: tests/test_gpu_stft_gl.py).
//
// One launch = one iteration over every frame of every clip-channel:
//   MODE 0 (init)  : Z = |S| * angles0                         -> ISTFT -> x_0
//   MODE 1 (first) : a = STFT(x_0)            (tprev == 0)     -> normalise -> ISTFT -> x_1
//   MODE 2 (iter)  : a = STFT(x_k - m x_{k-1})                 -> normalise -> ISTFT -> x_{k+1}
// A workgroup walks a run of consecutive frames of one clip-channel, so that the 10-way overlap-add
// of torch.istft is a register sliding window (thread n' owns sample n' of every hop block) and the
// only cross-workgroup traffic is the 9-block halo at each end of a run.  Halo blocks are never
// combined with atomics: partial sums go to one of two parity audio buffers and the reader adds the
// two.  Round 6: the parity is that of the frame's GROUP (kGlGroup = 16 consecutive frames of a row,
// rfx_kernels.h) and the nine blocks across EVERY group boundary are split into two partial sums -
// inside a run exactly as between two runs - so a clip's bits do not depend on where the launch's run
// boundaries fall (on the batch the clip is converted in), and the per-frame form's fold reproduces them.
#define RFX_PK 1  // packed fp32 butterflies (rfx_core.h)
#include "rfx_frame.hip.h"
#include "rfx_kernels.h"

namespace rfx {

#ifndef RFX_MIN_WAVES
#define RFX_MIN_WAVES 4
#endif
#ifndef RFX_NO_PREFETCH_D
#define RFX_PREFETCH_D 1  // next frame's new analysis sample is loaded across the synthesis barrier
#endif
#ifndef RFX_STREAM_AUX
#define RFX_STREAM_AUX 2  // gfx950 'nt': |S| and the injected angles are read once per launch
#endif

// |S| of one thread's 21 slots: five 16-B loads + one 4-B load, issued early (under P2/P3)
struct MagRegs {
  v4f s4[5];
  float tail;
};
__device__ __forceinline__ void mag_issue(MagRegs& m, rsrc_t S, unsigned foff, unsigned q16) {
#pragma unroll
  for (int i = 0; i < 5; ++i) m.s4[i] = ld4<RFX_STREAM_AUX>(S, q16, foff + (unsigned)i * (kQPad * 16u));
  m.tail = ld1<RFX_STREAM_AUX>(S, q16 >> 2, foff + 20u * kQPad * 4u);
}
// the same in two parts: the first EARLY 16-byte groups right after the analysis barrier (HBM latency is longer than P3
// alone), the rest after P2, when its registers have become free
#ifndef RFX_MAG_SPLIT
#define RFX_MAG_SPLIT 2
#endif
template <int PART>
__device__ __forceinline__ void mag_issue_part(MagRegs& m, rsrc_t S, unsigned foff, unsigned q16) {
  constexpr int lo = PART == 0 ? 0 : RFX_MAG_SPLIT, hi = PART == 0 ? RFX_MAG_SPLIT : 5;
#pragma unroll
  for (int i = lo; i < hi; ++i) m.s4[i] = ld4<RFX_STREAM_AUX>(S, q16, foff + (unsigned)i * (kQPad * 16u));
  if (PART == 1) m.tail = ld1<RFX_STREAM_AUX>(S, q16 >> 2, foff + 20u * kQPad * 4u);
}
__device__ __forceinline__ float mag_at(const MagRegs& m, int kb) { return kb < 20 ? m.s4[kb >> 2][kb & 3] : m.tail; }

template <int MODE>
__global__ void __launch_bounds__(kThreads, RFX_MIN_WAVES) gl_iter_kernel(GlArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, g.tw1, g.tw2);
#ifdef RFX_WGCLOCK
  // diagnostic build (tools/probe_wgclock.py): when does every workgroup of every launch start and end, and on which XCD / CU -
  // is the tail a launch boundary exposes the same workgroups every time (placement) or a different set each launch (chance)?
  const unsigned long long wg_t0 = wall_clock64(), wg_c0 = __builtin_readcyclecounter();  // 100 MHz wall clock, shader-clock counter
#endif

  // The launch's rows (clip-channels) are cut into groups of kGlGroup frames; run r of the launch's gridDim.x runs owns a whole
  // number of consecutive groups, counted row after row (gl_run_start_frame: the first N mod R runs one group more than the others -
  // the workgroups dispatched first, which finish early; the host keeps R <= the chip's resident workgroup slots).  A run that
  // crosses a row boundary is walked as one SEGMENT per row.  Which run a group falls into changes nothing in its arithmetic (the
  // header comment): B T < 2^31 (checked by the host).
  const int gf_end = (int)gl_run_start_frame((long long)blockIdx.x + 1, gridDim.x, g.B, g.T, g.run_h, g.run_w1, g.run_w2);
  int gf = (int)gl_run_start_frame(blockIdx.x, gridDim.x, g.B, g.T, g.run_h, g.run_w1, g.run_w2);
  const int nblk = g.T - 1;  // hop blocks kept by istft's centre trim

  const size_t clip_slots = (size_t)g.T * kFrameStride;
  const bool have_init = g.angles0 != nullptr;
  const rsrc_t scl = make_rsrc(g.out_scale, (size_t)g.L * 4);
  const rsrc_t win = make_rsrc(g.win, kWin * 4);
  const unsigned npr4 = (unsigned)t.npr * 4u;
  const unsigned q16 = (unsigned)slot_qp(t.npr) * 16u;  // byte offset of this thread's 16-B slot groups in the streams
  __syncthreads();  // tw2 table in LDS
  Tw1 tw1;  // g(n')^k1 of this thread: fetched in the synthesis half, reused by the next frame's analysis
  if (MODE != 0) load_tw1(tw1, f);
  // the thread's ten Hann samples w[441 j + n']: fetched with the twiddles (in flight across the synthesis
  // barrier), used by the overlap-add and by the next frame's analysis
  float wv[10];
  auto load_window = [&] {
#pragma unroll
    for (int j = 0; j < 10; ++j) wv[j] = ld1(win, npr4, (unsigned)j * (kHop * 4u));
  };
  if (MODE != 0) load_window();

  while (gf < gf_end) {
  const int clip = gf / g.T;
  const int t0 = gf - clip * g.T;
  const int t1 = min(g.T - 1, t0 + (gf_end - gf) - 1);
  gf += t1 - t0 + 1;
  const rsrc_t Ssrc = make_rsrc(g.S + clip * clip_slots, clip_slots * sizeof(float));
  const rsrc_t init = make_rsrc(have_init ? (const void*)(g.angles0 + clip * clip_slots) : (const void*)g.S,
                                clip_slots * sizeof(cf));
  const rsrc_t in0 = make_rsrc(g.audio_in[0] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t in1 = make_rsrc(g.audio_in[1] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t pv0 = make_rsrc(g.audio_prev[0] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  const rsrc_t pv1 = make_rsrc(g.audio_prev[1] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  // numeric range (round 6): the row's analysis input times a power of two, eps^2 in those units (GlArgs::row_scale)
  const float ks = g.row_scale ? g.row_scale[2 * clip] : 1.f, eps2 = g.row_scale ? g.row_scale[2 * clip + 1] : 1e-32f;
  (void)ks;
  (void)eps2;
  // the group of the frame being synthesised: [tg0, tg1], its partial sums go to the buffer of its parity (outA), explicit zeros
  // for blocks no other group touches to the other one (outB); t0 is a group start (runs are whole groups)
  int tg0 = t0, tg1 = min(g.T - 1, t0 + kGlGroup - 1);
  const int par = (t0 / kGlGroup) & 1;
  rsrc_t outA = make_rsrc(g.audio_out[par] + (size_t)clip * g.Lpad, (size_t)g.L * 4);
  rsrc_t outB = make_rsrc(g.audio_out[par ^ 1] + (size_t)clip * g.Lpad, (size_t)g.L * 4);

  float acc[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) acc[j] = 0.f;

  // analysis input d = x_k - m*x_{k-1} of thread n' for hop blocks fr-5 .. fr+4: a register sliding
  // window like `acc` (the reflect-padded signal is a fixed function of the padded position, so the
  // value a frame needs for block beta is the one its predecessor loaded): 4 loads per frame, not 40
  // The next frame's new sample is REQUESTED before the synthesis barrier and COMBINED after it (at the top of the next trip):
  // with the sum formed where the loads are issued, hipcc put it - and an `s_waitcnt vmcnt(0)` for all 35 requests of the phase,
  // twiddles and window included - in front of the barrier (seen in the ISA, round 4), which is not what "in flight across the
  // barrier" means.  (Measured on one box: 24.5-25.1 ms with the wait, 24.6-25.0 without - the requests had landed by then.)
  struct DRaw { float a0, a1, p0, p1; };
  auto request_d = [&](int blk) {
    const unsigned p4 = (unsigned)reflect_index(blk * kHop + t.npr, g.L) * 4u;
    DRaw r{ld1(in0, p4, 0), ld1(in1, p4, 0), 0.f, 0.f};
    if (MODE == 2) { r.p0 = ld1(pv0, p4, 0); r.p1 = ld1(pv1, p4, 0); }
    return r;
  };
  auto combine_d = [&](const DRaw& r) {
    float x = r.a0 + r.a1;
    if (MODE == 2) x = fmaf(-g.mom, r.p0 + r.p1, x);
    return x * ks;
  };
  auto load_d = [&](int blk) { return combine_d(request_d(blk)); };
  float d[10];
  DRaw d_next{0.f, 0.f, 0.f, 0.f};
  if (MODE != 0) {
#pragma unroll
    for (int j = 1; j < 10; ++j) d[j] = load_d(t0 + j - 1 - kHalfHops);  // blocks of frame t0-1 shifted in below
    d_next = request_d(t0 + 9 - kHalfHops);
  }
  (void)d_next;

  // a finished hop block: value * istft normalisation -> this run's parity buffer (and an explicit zero
  // in the other one when no neighbouring run contributes to the block)
  auto emit_scaled = [&](int blk, float scaled) {
    if (blk < 0 || blk >= nblk || !t.active) return;  // blk is wave-uniform
    const unsigned boff = (unsigned)blk * (kHop * 4u);
    const bool full = (max(blk - 4, 0) >= tg0) && (min(blk + 5, g.T - 1) <= tg1);
    st1(scaled, outA, npr4, boff);
    if (full) st1(0.f, outB, npr4, boff);
  };
  auto scale_of = [&](int blk) {
    return (blk >= 0 && blk < nblk) ? ld1(scl, npr4, (unsigned)blk * (kHop * 4u)) : 0.f;
  };
  auto emit = [&](int blk, float val) { emit_scaled(blk, val * scale_of(blk)); };
  // gfx950 retires VMEM loads and stores in issue order: a store issued at the end of a frame would sit
  // in front of the next frame's first loads and expose its write latency.  The finished block is
  // therefore parked in a register and stored after the next frame's analysis barrier, where a long
  // LDS/VALU stretch follows; its normalisation factor is fetched across the synthesis barrier.
  float pend_val = 0.f, pend_scale = 0.f;
  int pend_blk = -1;

#ifdef RFX_TIMING
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = wall_clock64();
#define RFX_STAMP(i) do { unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define RFX_STAMP(i) ((void)0)
#endif
  for (int fr = t0; fr <= t1; ++fr) {
    if ((fr & (kGlGroup - 1)) == 0 && fr != t0) {
      // ---- group boundary inside the run (wave-uniform): what the end of a run does - the parked block (complete, of the old
      // group) and the partial sums of the nine blocks across the cut leave for the old group's buffer; the new group starts its
      // own chains from zero in the other one
      asm volatile("; RFX_ONCE_PER_GROUP_BEGIN");  // (markers for tools/isa_mix.py: this block runs once per kGlGroup frames)
      emit_scaled(pend_blk, pend_val);
      pend_blk = -1;
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        emit(fr - kHalfHops + j, acc[j]);
        acc[j] = 0.f;
      }
      tg0 = fr;
      tg1 = min(g.T - 1, fr + kGlGroup - 1);
      const rsrc_t tmp = outA;
      outA = outB;
      outB = tmp;
      asm volatile("; RFX_ONCE_PER_GROUP_END");
    }
    const unsigned foff = (unsigned)fr * (kFrameStride * 4u);
    const unsigned rng_key = rand_frame_key(g.seed, g.frame_base + (unsigned long long)clip * g.T + fr);

    cf R[21];
    MagRegs mag;
    if (MODE != 0) {
      // ---- analysis of x_k - m*x_{k-1}: reflect-padded, Hann-windowed frame centred on sample 441*fr
      float u[10];
#pragma unroll
      for (int j = 0; j < 9; ++j) d[j] = d[j + 1];
#ifdef RFX_PREFETCH_D
      d[9] = combine_d(d_next);
#else
      d[9] = load_d(fr + 9 - kHalfHops);
#endif
#pragma unroll
      for (int j = 0; j < 10; ++j) u[j] = d[j] * wv[j];
#ifdef RFX_ABL_NOFFT
      mag_issue(mag, Ssrc, foff, q16);
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) R[kb] = cf{u[kb % 10], u[(kb + 3) % 10]};
#else
      frame_forward_tw(u, R, f, t, tw1,
                    [&] {
                      RFX_STAMP(1);
                      emit_scaled(pend_blk, pend_val);
                      pend_blk = -1;
                      mag_issue_part<0>(mag, Ssrc, foff, q16);
                    },
                    [&] { RFX_STAMP(0); },
                    [&] {
                      mag_issue_part<1>(mag, Ssrc, foff, q16);  // the rest of |S| flies under P3 (its registers are not live during P2)
                    });
      RFX_STAMP(2);
#ifdef RFX_TIMING
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): how long the |S| stream is still outstanding after P3
      RFX_STAMP(7);
#endif
#endif
      // ---- angles = a / (|a| + 1e-16);  next spectrum estimate Z = |S| * angles
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) R[kb] = gl_project(R[kb], mag_at(mag, kb), eps2);
    } else {
      // ---- Z = |S| * angles0 with angles0 injected or drawn (rand_init=True, spectrogram_converter.py:72:
      // U[0,1) real and imaginary parts per BIN; a conjugate slot conjugates its primary's draw)
      mag_issue(mag, Ssrc, foff, q16);
      if (have_init) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          const v4f v = ld4<RFX_STREAM_AUX>(init, q16, 2u * foff + (unsigned)i * (kQPad * 16u));
          R[2 * i] = cf{v.x, v.y};
          R[2 * i + 1] = cf{v.z, v.w};
        }
        const v2f w = ld2<RFX_STREAM_AUX>(init, q16 >> 1, 2u * foff + 20u * kQPad * 8u);
        R[20] = cf{w.x, w.y};
      } else {
#pragma unroll
        for (int kb = 0; kb < 21; ++kb) {
          bool cj;
          const int bin = slot_bin(t.k1, t.idx, kb, &cj);
          const cf r = rand_unit_pair(rng_key, bin);
          R[kb] = cf{r.re, cj ? -r.im : r.im};
        }
      }
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) {
        const float s = mag_at(mag, kb);
        R[kb] = cf{s * R[kb].re, s * R[kb].im};
      }
    }
    RFX_STAMP(3);

    // ---- synthesis: inverse transform, synthesis window, overlap-add into the sliding window
    float y[10];
#ifdef RFX_ABL_NOFFT
#pragma unroll
    for (int j = 0; j < 10; ++j) y[j] = R[j].re + R[j + 10].im + R[20].re;
#else
#ifdef RFX_PREFETCH_D
    // the next frame's new analysis sample goes in flight across the synthesis barrier
    frame_inverse_tw(R, y, f, t, tw1,
                  [&] {
                    RFX_STAMP(4);
                    load_window();
                    if (MODE != 0) d_next = request_d(fr + 10 - kHalfHops);
                    pend_scale = scale_of(fr - kHalfHops);
                  },
                  [&] { RFX_STAMP(5); }, [&](int i) { RFX_STAMP(8 + i); });
#else
    frame_inverse_tw(R, y, f, t, tw1, [&] { RFX_STAMP(4); load_window(); }, [&] { RFX_STAMP(5); });
#endif
#endif
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = fmaf(y[j], wv[j], acc[j]);
#ifdef RFX_PREFETCH_D
    if (MODE != 0) {
      pend_val = acc[0] * pend_scale;
      pend_blk = fr - kHalfHops;
    } else {
      emit(fr - kHalfHops, acc[0]);
    }
#else
    emit(fr - kHalfHops, acc[0]);
#endif
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
    for (int i = 0; i < 12; ++i) g.timing[((size_t)blockIdx.x * kWaves + w) * 12 + i] = tacc[i];
  }
#endif
  // ---- flush the parked block and the right halo of the segment
  emit_scaled(pend_blk, pend_val);
#pragma unroll
  for (int j = 0; j < 9; ++j) emit(t1 - 4 + j, acc[j]);
  }  // next segment of the run (another clip)
#ifdef RFX_WGCLOCK
  __syncthreads();
  if (g.timing && threadIdx.x == 0) {
    unsigned long long* rec = g.timing + ((size_t)g.launch * gridDim.x + blockIdx.x) * 4;
    rec[0] = wg_t0;
    rec[1] = wall_clock64();
    rec[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    rec[3] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20)   // HW_REG_XCC_ID in the low four bits,
             | ((__builtin_readcyclecounter() - wg_c0) << 8);                 // shader cycles the workgroup lived above them
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Small batches (a server decodes ONE tile per request, reference server.py:152-164): the run-based kernel above needs
// >= 10 consecutive frames per workgroup, i.e. at most T/10 = 51 workgroups per clip-channel - a fifth of the chip for a
// mono tile.  Here every frame is its own unit of work: a workgroup analyses frame t of x_k - m x_{k-1}, projects, and
// writes the frame's 4410 windowed synthesis samples to a frame buffer; gl_fold_kernel then overlap-adds the (up to) ten
// frames that cover a sample and applies torch.istft's envelope division.  Twice the launches and 35 KB of extra traffic
// per frame, but 512 workgroups per clip-channel: one iteration of a single tile takes ~25 us instead of ~90 us.
// Chosen by the host for B*T <= 4 frames per resident workgroup slot (RFX_GL_LATENCY_MODE=0 disables).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kFramePitch = 4416;  // 4410 samples per synthesis frame, rounded up to whole 64-byte lines

template <int MODE>
__global__ void __launch_bounds__(kThreads, RFX_MIN_WAVES) gl_frame_kernel(GlFrameArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, g.tw1, g.tw2);
  const rsrc_t win = make_rsrc(g.win, kWin * 4);
  const unsigned npr4 = (unsigned)t.npr * 4u;
  const unsigned q16 = (unsigned)slot_qp(t.npr) * 16u;
  float wv[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) wv[j] = ld1(win, npr4, (unsigned)j * (kHop * 4u));
  Tw1 tw1;
  if (MODE != 0) load_tw1(tw1, f);
  __syncthreads();  // tw2 table in LDS

  const long long nframes = (long long)g.B * g.T;
  for (long long gf = blockIdx.x; gf < nframes; gf += gridDim.x) {
    const int clip = (int)(gf / g.T), fr = (int)(gf - (long long)clip * g.T);
    const size_t clip_slots = (size_t)g.T * kFrameStride;
    const rsrc_t Ssrc = make_rsrc(g.S + clip * clip_slots, clip_slots * sizeof(float));
    const unsigned foff = (unsigned)fr * (kFrameStride * 4u);
    cf R[21];
    MagRegs mag;
    if (MODE != 0) {
      const rsrc_t in = make_rsrc(g.audio_in + (size_t)clip * g.Lpad, (size_t)g.L * 4);
      const rsrc_t pv = make_rsrc(g.audio_prev + (size_t)clip * g.Lpad, (size_t)g.L * 4);
      const float ks = g.row_scale ? g.row_scale[2 * clip] : 1.f, eps2 = g.row_scale ? g.row_scale[2 * clip + 1] : 1e-32f;
      float u[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const unsigned p4 = (unsigned)reflect_index((fr + j - kHalfHops) * kHop + t.npr, g.L) * 4u;
        float x = ld1(in, p4, 0);
        if (MODE == 2) x = fmaf(-g.mom, ld1(pv, p4, 0), x);
        u[j] = (x * ks) * wv[j];  // (the run kernel scales when the sample enters its sliding window: same two products)
      }
      frame_forward_tw(u, R, f, t, tw1, [&] { mag_issue(mag, Ssrc, foff, q16); });
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) R[kb] = gl_project(R[kb], mag_at(mag, kb), eps2);
    } else {
      mag_issue(mag, Ssrc, foff, q16);
      if (g.angles0) {
        const rsrc_t init = make_rsrc(g.angles0 + clip * clip_slots, clip_slots * sizeof(cf));
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          const v4f v = ld4<RFX_STREAM_AUX>(init, q16, 2u * foff + (unsigned)i * (kQPad * 16u));
          R[2 * i] = cf{v.x, v.y};
          R[2 * i + 1] = cf{v.z, v.w};
        }
        const v2f w = ld2<RFX_STREAM_AUX>(init, q16 >> 1, 2u * foff + 20u * kQPad * 8u);
        R[20] = cf{w.x, w.y};
      } else {
        const unsigned rng_key = rand_frame_key(g.seed, g.frame_base + (unsigned long long)clip * g.T + fr);  // same stream as gl_iter_kernel
#pragma unroll
        for (int kb = 0; kb < 21; ++kb) {
          bool cj;
          const int bin = slot_bin(t.k1, t.idx, kb, &cj);
          const cf r = rand_unit_pair(rng_key, bin);
          R[kb] = cf{r.re, cj ? -r.im : r.im};
        }
      }
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) {
        const float s = mag_at(mag, kb);
        R[kb] = cf{s * R[kb].re, s * R[kb].im};
      }
    }
    float y[10];
    frame_inverse_tw(R, y, f, t, tw1);
    if (t.active) {
      float* __restrict__ out = g.frames + (size_t)gf * kFramePitch + t.npr;
#pragma unroll
      for (int j = 0; j < 10; ++j) out[j * kHop] = y[j];  // un-windowed: the fold forms the run kernel's fma chains
    }
    __syncthreads();  // the next frame's first LDS stores overwrite rows other waves are still gathering in P1'
  }
}

// x[clip][p]: overlap-add of the frames t = blk-4 .. blk+5 that cover hop block blk = p / 441, in the run kernel's arithmetic
// (round 6, kGlGroup in rfx_kernels.h): an fma chain y w + acc in increasing t, split where a group boundary of the row falls
// inside the block's frames, each side scaled by istft's normalisation, then added - the two parity buffers of gl_iter_kernel.
// Both forms give a clip the same bits (tests/test_gpu_round6.py).
__global__ void __launch_bounds__(256) gl_fold_kernel(const float* __restrict__ frames, const float* __restrict__ win,
                                                      const float* __restrict__ scale, float* __restrict__ out, int T, int L, size_t out_stride) {
#pragma clang fp contract(off)  // lo s + hi s below is two products and a sum, as in the run form (two stores and a load apart there)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int clip = blockIdx.y;
  if (p >= L) return;
  const int blk = p / kHop, n = p - blk * kHop;
  const int tlo = max(blk - 4, 0), thi = min(blk + 5, T - 1);
  const int cut = thi & ~(kGlGroup - 1);  // first frame of the group that finishes the block
  float lo = 0.f, hi = 0.f;
  for (int t = tlo; t <= thi; ++t) {
    const int j = blk - t + kHalfHops;
    const float y = frames[((size_t)clip * T + t) * kFramePitch + j * kHop + n], w = win[j * kHop + n];
    if (t < cut) lo = __fmaf_rn(y, w, lo);
    else hi = __fmaf_rn(y, w, hi);
  }
  const float s = scale[p];
  const float a = lo * s, b = hi * s;  // (HIP's __fmul_rn is a plain product the compiler may contract: the pragma above is what keeps these apart)
  out[(size_t)clip * out_stride + p] = a + b;
}

hipError_t launch_gl_frame(int mode, const GlFrameArgs& g, int nblocks, hipStream_t stream) {
  const size_t lds = kFrameDynLdsBytes;
  switch (mode) {
    case 0: hipLaunchKernelGGL(gl_frame_kernel<0>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
    case 1: hipLaunchKernelGGL(gl_frame_kernel<1>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
    default: hipLaunchKernelGGL(gl_frame_kernel<2>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
  }
  return hipGetLastError();
}
hipError_t launch_gl_fold(const float* frames, const float* win, const float* scale, float* out, int B, int T, int L, size_t out_stride, hipStream_t stream) {
  hipLaunchKernelGGL(gl_fold_kernel, dim3((L + 255) / 256, B), dim3(256), 0, stream, frames, win, scale, out, T, L, out_stride);
  return hipGetLastError();
}
size_t gl_frame_buffer_bytes(int B, int T) { return (size_t)B * T * kFramePitch * sizeof(float); }

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
  const size_t lds = kFrameDynLdsBytes;
  switch (mode) {
    case 0: hipLaunchKernelGGL(gl_iter_kernel<0>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
    case 1: hipLaunchKernelGGL(gl_iter_kernel<1>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
    default: hipLaunchKernelGGL(gl_iter_kernel<2>, dim3(nblocks), dim3(kThreads), lds, stream, g); break;
  }
  return hipGetLastError();
}

// Called at plan creation with the plan's device current: HIP keeps function attributes per device, so
// there is no process-wide "done" flag anywhere in this library.
hipError_t prepare_gl_kernels() {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)gl_iter_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)gl_iter_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)gl_iter_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)gl_frame_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void*)gl_frame_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes)) != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)gl_frame_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes);
}

int gl_blocks_per_cu() {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gl_iter_kernel<2>, kThreads, kFrameDynLdsBytes) != hipSuccess || n < 1) n = 1;
  return n;
}

hipError_t launch_gl_combine(const float* a0, const float* a1, float* out, int B, int L, int Lpad, hipStream_t stream) {
  const size_t total = (size_t)B * L;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gl_combine_kernel, dim3(blocks), dim3(256), 0, stream, a0, a1, out, L, Lpad, total);
  return hipGetLastError();
}

}  // namespace rfx
