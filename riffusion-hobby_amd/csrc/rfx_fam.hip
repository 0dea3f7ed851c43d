// rfx_fam.hip - Griffin-Lim (torchaudio.transforms.GriffinLim as constructed at riffusion/spectrogram_converter.py:62-73,
// called at :204) and the forward STFT (Spectrogram(power=None), :47-59, :179; fam_fwd_kernel below) for the ROW FAMILY of STFT geometries: n_fft = 40 h, win_length = 10 h, any hop - the reference's default
// 400 / 100 / 10 ms (spectrogram_params.py:24-27, :62-81) at 48, 32, 24, 16 and 8 kHz (cli.py:43 takes the sample rate from
// the input file).  Round 2 / 3 ran these on the generic engine (rfx_generic.hip: a complex FFT of n_fft / 2 points over a
// runtime digit list, eleven barrier-separated phases per frame, 587 tiles/s at 48 kHz against 1960 at 44.1 kHz); here they
// get the specialised engine's factorisation (rfx_core.h) with h in place of 441:
//     P1   thread n' < h          pruned 40-point transform of the ten windowed samples h j + n' (only the window's quarter of
//                                 the frame is ever touched), twiddle g(n')^k1 -> 21 rows x h in LDS
//     A    thread (row, i < RB)   radix-RA butterfly down the row, twiddle W_h^{i p}
//     B    thread (row, p < RA)   radix-RB butterfly -> RB slots in registers: Z = S a / (|a| + 1e-16), inverse butterfly
//     A'   inverse of A           P1'  pruned inverse -> ten windowed synthesis samples -> HBM
// Four barriers per frame.  `a` is the spectrum of x_k - m x_{k-1}: the momentum term of the reference's
// `rebuilt - m tprev` is applied in the time domain (the STFT is linear, rfx_gl.hip), so no spectrum ever reaches HBM.
// One frame per workgroup trip (grid-stride), gen_fold_kernel (rfx_generic.hip) overlap-adds the frames: two launches per
// iteration, same buffers and same random stream as the generic engine, which remains the fallback for every other geometry.
// |S| arrives in the plan's plain bin-ordered layout and is re-ordered ONCE per call into slot order (thread t's slot s at
// s * nthr + t: every wave-wide load is whole lines); duplicate slots get the same magnitude.
// This file is compiled TWICE (round 4).  Translation unit 0 (rfx_fam.hip itself, plain fp32 butterflies): the forward kernels,
// the 48 kHz Griffin-Lim kernel, the launchers.  Translation unit 1 (rfx_fam_pk.hip: `#define RFX_PK 1`, `#define RFX_FAM_TU 1`,
// then this file): the Griffin-Lim kernels of every other geometry with PACKED butterflies (rfx_core.h).  Measured, Griffin-Lim 32
// of 64 tiles, packed against plain: 32 kHz 30.3 / 33.0 ms, 24 kHz 22.4 / 22.9, 22.05 kHz 20.7 / 22.2, 16 kHz 16.4 / 17.2,
// 8 kHz 8.2 / 9.1 - no scratch left in those kernels - but 48 kHz 57.8 / 48.9 (the radix-24 pass spills 100 B with the packed
// complex product's aligned pairs) and the forward kernels spill more everywhere.  The kernel templates carry the unit's number so
// that the two instantiations of one geometry are different symbols.  (-DRFX_FAM_PK: packed in unit 0 as well, A/B runs.)
#ifndef RFX_FAM_TU
#define RFX_FAM_TU 0
#endif
#if defined(RFX_FAM_PK) && !defined(RFX_PK)
#define RFX_PK 1
#endif
#include <hip/hip_runtime.h>

#include "rfx_fam_core.h"
#include "rfx_frame.hip.h"  // buffer-descriptor loads / stores (SRD in SGPRs + one 32-bit lane offset: no 64-bit per-lane addresses)
#include "rfx_kernels.h"

namespace rfx {

constexpr int fam_threads(int ra, int rb, int nr = 40) {
  int n = ra * rb;
  if ((nr / 2 + 1) * rb > n) n = (nr / 2 + 1) * rb;
  if ((nr / 2 + 1) * ra > n) n = (nr / 2 + 1) * ra;
  return (n + 63) / 64 * 64;
}

// Measured at 48 kHz / 16 kHz, 64 tiles x 32 iterations (tools/probe_fam.py): 16-byte LDS accesses in pass B together with a
// raised issue priority for the B phase 52.6 -> 50.4 ms / 19.2 -> 18.2 ms (each alone: 51.0 / 52.6 ms); both halves of the
// forward pass-A twiddles requested before the butterfly: 57.6 ms (registers) - off.
#ifndef RFX_FAM_VEC
#define RFX_FAM_VEC 1
#endif
#ifndef RFX_FAM_PRIO
#define RFX_FAM_PRIO 1
#endif
#ifndef RFX_FAM_TW_EARLY
#define RFX_FAM_TW_EARLY 0
#endif
#ifndef RFX_FAM_GL_STREAM_FWD
#define RFX_FAM_GL_STREAM_FWD true  // 48 kHz Griffin-Lim kernel: streamed radix-24 pass A (rfx_fam_core.h) - forward 38.3 -> 37.7 ms per 64 tiles x 32 iterations, inverse 39.8: off
#endif
#ifndef RFX_FAM_GL_STREAM_INV
#define RFX_FAM_GL_STREAM_INV false
#endif
#ifndef RFX_FAM_STORE_AUX
#define RFX_FAM_STORE_AUX 0  // cache policy of the synthesis-frame stores (the fold reads them back once)
#endif
#ifndef RFX_FAM_STREAM_AUX
#define RFX_FAM_STREAM_AUX kAuxNT  // |S| is read once per iteration: streamed past L2
#endif

#if RFX_FAM_TU == 0
bool fam_row_stride_even(const FamGeom& g) { return RFX_FAM_VEC && g.rb % 2 == 0; }
size_t fam_lds_bytes(const FamGeom& g) { return sizeof(cf) * (size_t)g.rows * g.rs; }
#endif

// Where the pass-A twiddles W_h^{i p} live: in LDS (a static array next to the cube: the compiler then knows that cube stores
// never alias twiddle reads) wherever two workgroups still share a CU with it - every geometry but 48 kHz, whose cube leaves
// 1.2 KB - and in the L1-resident global table otherwise.
constexpr bool fam_twiddles_in_lds(int ra, int rb) { return ra * rb != 480; }
#if RFX_FAM_TU == 0
size_t fam_static_lds_bytes(const FamGeom& g) { return fam_twiddles_in_lds(g.ra, g.rb) ? sizeof(cf) * (size_t)g.rb * (g.ra - 1) : 0; }
#endif

// the thread's RA - 1 pass-A twiddles, fetched in two batches
template <int RA, int RB, bool LDS>
struct FamTwA {
  cf w[RA];
  rsrc_t src;
  unsigned voff;
  const cf* tab;  // LDS column of this thread (entry p - 1 at (p - 1) RB)
  template <bool INV, int BATCH, bool STREAM = false>  // the twiddles of batch BATCH of the forward / inverse pass (rfx_fam_core.h::fam_tw_in_batch)
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int p = 1; p < RA; ++p) {
      if (!fam_tw_in_batch(RA, INV, STREAM, BATCH, p)) continue;
#if defined(RFX_FAM_ABL) && RFX_FAM_ABL >= 1  // timing ablation (wrong results): no pass-A twiddle fetches
      w[p] = cf{1.f, (float)p};
      continue;
#endif
      if (LDS) {
        w[p] = tab[(p - 1) * RB];
      } else {
        const v2f t = ld2(src, voff, (unsigned)(p - 1) * (RB * 8u));
        w[p] = cf{t.x, t.y};
      }
    }
  }
  template <bool INV, bool STREAM = false>
  __device__ __forceinline__ void load_batch(int b) {  // b >= 1 is a compile-time constant wherever this is called (unrolled stages)
    if (b == 1) load<INV, 1, STREAM>();
    if (fam_tw_batches(RA, INV, STREAM) > 2 && b == 2) load<INV, 2, STREAM>();
    if (fam_tw_batches(RA, INV, STREAM) > 3 && b == 3) load<INV, 3, STREAM>();
  }
};

// MODE 0: Z = S * angles0 (injected or drawn) -> synthesis; MODE 1 / 2: analysis of d = x_k - m x_{k-1} (x_0 in the first iteration;
// the two modes are the same code since the fold forms d)
// 128 VGPRs: two 512-thread workgroups per CU at 48 kHz.  Every table / HBM value is requested one barrier before its use:
//   before P1 | A : first half of the pass-A twiddles            before A | B : the frame's |S|
//   before B' | A': first half of the conjugate twiddles          before A'| P1': g(n')^k1 and the Hann samples (P1' and the NEXT
//   frame's P1 use the same values), and the next frame's ten input samples
// Phases are fenced for the compiler's scheduler (left alone it hoists the next phase's loads over the current butterfly and
// spills 70 registers).
template <int MODE, int RA, int RB, int NR = 40, int TU = RFX_FAM_TU>
__global__ void __launch_bounds__(fam_threads(RA, RB, NR)) __attribute__((amdgpu_waves_per_eu(4))) fam_gl_kernel(FamGlArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* cube = reinterpret_cast<cf*>(smem);
  constexpr int H = RA * RB;
  constexpr int NT = fam_threads(RA, RB, NR);
  constexpr int ROWS = NR / 2 + 1, WH = NR / 4;  // rows of the cube, window blocks of h samples (rfx_fam_core.h)
  const int tid = threadIdx.x;
  const int rs = a.g.rs;
  const bool act1 = tid < H;
  const int npr = act1 ? tid : H - 1;  // idle lanes shadow the last active one (loads only, never stores)
  const int col = NR == 40 ? npr : (npr + a.g.left) % H;  // cube column of this thread's window samples
  const bool actA = tid < ROWS * RB;
  const int tA = actA ? tid : ROWS * RB - 1;
  const int rowA = tA / RB, iA = tA - rowA * RB;
  const bool actB = tid < ROWS * RA;
  const int tB = actB ? tid : ROWS * RA - 1;
  const int rowB = tB / RA, pB = tB - rowB * RA;
  cf* const rowa = cube + rowA * rs + iA;
  cf* const rowb = cube + rowB * rs + pB * RB;
  const rsrc_t tw1 = make_rsrc(a.tw1, (size_t)ROWS * H * sizeof(cf));
  const rsrc_t win = make_rsrc(a.win, (size_t)WH * H * sizeof(float));
  const unsigned npr4 = (unsigned)npr * 4u, npr8 = (unsigned)npr * 8u, tB4 = (unsigned)tB * 4u;
  const float oscale = 2.0f / (float)a.g.n_fft;
  const long long nframes = (long long)a.B * a.T;

  constexpr bool VEC = RFX_FAM_VEC && RB % 2 == 0;  // the host picks an even row stride then (fam_row_stride_even)
  constexpr bool TWL = fam_twiddles_in_lds(RA, RB);
  __shared__ __attribute__((aligned(16))) cf twa_lds[TWL ? RB * (RA - 1) : 1];
  if (TWL)
    for (int i = tid; i < RB * (RA - 1); i += NT) twa_lds[i] = a.twa[i];
  __syncthreads();  // the first pass-A read of the frame loop may precede the loop's first barrier, and reads other waves' entries
  FamTwA<RA, RB, TWL> wa;
  wa.src = make_rsrc(a.twa, (size_t)RB * (RA - 1) * sizeof(cf));
  wa.voff = (unsigned)iA * 8u;
  wa.tab = twa_lds + iA;
  // g(n')^k1 for k1 = 1..10 and 20 only (fam_g_pow)
  cf w1[12];
  float wv[WH], u[WH];
  auto load_tables = [&] {
#if defined(RFX_FAM_ABL) && RFX_FAM_ABL >= 2  // timing ablation (wrong results): no g^k1 / Hann fetches either
#pragma unroll
    for (int k = 1; k <= (NR == 40 ? 11 : 10); ++k) w1[k] = cf{1.f, (float)k};
#pragma unroll
    for (int j = 0; j < WH; ++j) wv[j] = (float)j;
    return;
#endif
#pragma unroll
    for (int k = 1; k <= (NR == 40 ? 11 : 10); ++k) {
      const v2f t = ld2(tw1, npr8, (unsigned)(k <= 10 ? k : 20) * (H * 8u));
      w1[k] = cf{t.x, t.y};
    }
#pragma unroll
    for (int j = 0; j < WH; ++j) wv[j] = ld1(win, npr4, (unsigned)j * (H * 4u));
  };
  auto g1 = [&w1](int k) { return fam_g_pow(w1, k); };
  // frame fr is centred on sample hop * fr of the reflect-padded estimate (torch.stft center=True): the window covers
  // positions hop * fr + off .. hop * fr + off + win - 1, off = left - n_fft / 2 (-5 h in the 40 h family)
  // (x_cur holds d = x_k - m x_{k-1} since round 4: the fold of the previous iteration forms it, one load per window sample
  // here instead of two and ten registers less across P1')
  auto load_samples = [&](long long gf) {
    const int clip = (int)(gf / a.T), fr = (int)(gf - (long long)clip * a.T);
    const rsrc_t xc = make_rsrc(a.x_cur + (size_t)clip * a.audio_stride, (size_t)a.L * sizeof(float));
#pragma unroll
    for (int j = 0; j < WH; ++j) u[j] = ld1(xc, (unsigned)reflect_index(a.g.hop * fr + a.g.off + j * H + npr, a.L) * 4u, 0);
  };
  auto window_samples = [&] {
#pragma unroll
    for (int j = 0; j < WH; ++j) u[j] *= wv[j];
  };
  if (MODE != 0 && (long long)blockIdx.x < nframes) {
    load_tables();
    load_samples(blockIdx.x);
    window_samples();
  }

#ifdef RFX_FAM_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
  int nfr = 0;
#define FSTAMP(i) do { unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define FSTAMP(i) ((void)0)
#endif
  for (long long gf = blockIdx.x; gf < nframes; gf += gridDim.x) {
    const rsrc_t S = make_rsrc(a.S + (size_t)gf * a.g.fsf, (size_t)a.g.fsf * sizeof(float));
    cf R[RB];
    FSTAMP(0);
    if (MODE != 0) {
      if (act1) fam_p1_forward_store<NR>(u, g1, cube, col, rs);
      RFX_SCHED_FENCE();
      wa.template load<false, 0, RFX_FAM_GL_STREAM_FWD>();
#if RFX_FAM_TW_EARLY
      for (int b = 1; b < fam_tw_batches(RA, false, RFX_FAM_GL_STREAM_FWD); ++b) wa.template load_batch<false, RFX_FAM_GL_STREAM_FWD>(b);
#endif
      FSTAMP(1);
      __syncthreads();
      FSTAMP(2);
      RFX_SCHED_FENCE();
      if (actA)
        fam_pass_a_forward<RA, RB, RFX_FAM_GL_STREAM_FWD>(rowa, 0, [&wa](int p) { return wa.w[p]; }, [&wa](int batch) {
          if (!RFX_FAM_TW_EARLY) wa.template load_batch<false, RFX_FAM_GL_STREAM_FWD>(batch);
          RFX_SCHED_FENCE();
        });
      RFX_SCHED_FENCE();
      float Sv[RB];
#pragma unroll
      for (int s = 0; s < RB; ++s) Sv[s] = ld1<RFX_FAM_STREAM_AUX>(S, tB4, (unsigned)s * (NT * 4u));
      FSTAMP(3);
      __syncthreads();
      FSTAMP(2);
      RFX_SCHED_FENCE();
#if RFX_FAM_PRIO
      __builtin_amdgcn_s_setprio(2);
#endif
      fam_pass_b_forward<RA, RB, VEC>(rowb, 0, R);
      const float eps2 = a.row_scale ? a.row_scale[2 * (gf / a.T) + 1] : 1e-32f;  // (the fold has applied row_scale[2 row] to x_cur)
#pragma unroll
      for (int s = 0; s < RB; ++s) R[s] = gl_project(R[s], Sv[s], eps2);
    } else {
      const unsigned rng_key = rand_frame_key(a.seed, a.frame_base + (unsigned long long)gf);  // the generic engine's stream
#pragma unroll
      for (int s = 0; s < RB; ++s) {
        bool cj;
        const int bin = fam_slot_bin(a.g, rowB, pB, s, &cj);
        cf ang;
        if (a.angles0) ang = a.angles0[(size_t)gf * a.fs_plain + bin];
        else ang = rand_unit_pair(rng_key, bin);
        const float sv = ld1<RFX_FAM_STREAM_AUX>(S, tB4, (unsigned)s * (NT * 4u));
        R[s] = cf{sv * ang.re, cj ? -(sv * ang.im) : sv * ang.im};
      }
    }
    if (actB) fam_pass_b_inverse<RA, RB, VEC>(rowb, 0, R);
#if RFX_FAM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    RFX_SCHED_FENCE();
    wa.template load<true, 0, RFX_FAM_GL_STREAM_INV>();
    FSTAMP(4);
    __syncthreads();
    FSTAMP(2);
    RFX_SCHED_FENCE();
    if (actA)
      fam_pass_a_inverse<RA, RB, RFX_FAM_GL_STREAM_INV>(rowa, 0, [&wa](int p) { return wa.w[p]; }, [&wa](int batch) {
        wa.template load_batch<true, RFX_FAM_GL_STREAM_INV>(batch);
        RFX_SCHED_FENCE();
      });
    RFX_SCHED_FENCE();
    load_tables();
    FSTAMP(5);
    __syncthreads();
    FSTAMP(2);
    RFX_SCHED_FENCE();
    // the next frame's samples are requested here and arrive underneath P1' (requested before the barrier, together with the
    // tables, the 61 loads of this phase queue up behind each other: 4 us of issue time per frame, measured)
    const bool more = MODE != 0 && gf + gridDim.x < nframes;
    if (more) load_samples(gf + gridDim.x);
    RFX_SCHED_FENCE();
    {
      float y[WH];
      fam_p1_load_inverse<NR>(cube, g1, y, col, rs);
      if (act1) {
        const rsrc_t out = make_rsrc(a.frames + (size_t)gf * a.fpitch + a.fshift, (size_t)WH * H * sizeof(float));
#pragma unroll
        for (int j = 0; j < WH; ++j) st1<RFX_FAM_STORE_AUX>(y[j] * (wv[j] * oscale), out, npr4, (unsigned)j * (H * 4u));
      }
    }
    RFX_SCHED_FENCE();
    if (more) window_samples();
    RFX_SCHED_FENCE();
    FSTAMP(6);
    // no barrier here when a P1 follows: its stores go to column n' of the rows - exactly the elements this thread has just
    // read in P1' - and nobody else touches a column between these two phases (a wave's LDS operations execute in order)
    if (MODE == 0) __syncthreads();  // (mode 0 goes straight to the next frame's B', which writes whole rows)
#ifdef RFX_FAM_TIMING
    ++nfr;
#endif
  }
#ifdef RFX_FAM_TIMING
  if (MODE == 1 && (blockIdx.x == 7 || blockIdx.x == 300) && (threadIdx.x == 0 || threadIdx.x == 256 || threadIdx.x == 448))
    printf("fam_gl timing, block %d of %d, thread %d (100 MHz ticks per frame, %d frames): P1 %.1f | barriers %.1f | A %.1f | B+proj+B' %.1f | A' %.1f | P1' %.1f | loop top %.1f\n",
           (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, nfr, (double)tacc[1] / nfr, (double)tacc[2] / nfr, (double)tacc[3] / nfr, (double)tacc[4] / nfr, (double)tacc[5] / nfr,
           (double)tacc[6] / nfr, (double)tacc[0] / nfr);
#endif
}

#if RFX_FAM_TU == 0
// ---- forward: Spectrogram(power=None) [+ abs] of the family geometries (spectrogram_converter.py:47-59, :179-182).  Same P1 /
// A / B as above; the frame then changes places through LDS - once every row has been read the cube's memory becomes the
// bin-ordered frame, each bin written by its primary slot (the direct one where a bin has two) - and leaves in whole lines, in
// the plan's plain layout [B*T][fs] (what rfx_stft hands out and gen_mel_kernel reads).  MODE 0: |X|, MODE 1: X.
template <int MODE, int RA, int RB, int NR = 40>
__global__ void __launch_bounds__(fam_threads(RA, RB, NR)) __attribute__((amdgpu_waves_per_eu(4))) fam_fwd_kernel(FamFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* cube = reinterpret_cast<cf*>(smem);
  float* magl = reinterpret_cast<float*>(smem);
  constexpr int H = RA * RB;
  constexpr int NT = fam_threads(RA, RB, NR);
  constexpr int ROWS = NR / 2 + 1, WH = NR / 4;
  constexpr bool VEC = RFX_FAM_VEC && RB % 2 == 0;
  constexpr bool TWL = fam_twiddles_in_lds(RA, RB);
  const int tid = threadIdx.x;
  const int rs = a.g.rs;
  const bool act1 = tid < H;
  const int npr = act1 ? tid : H - 1;
  const int col = NR == 40 ? npr : (npr + a.g.left) % H;
  const bool actA = tid < ROWS * RB;
  const int tA = actA ? tid : ROWS * RB - 1;
  const int rowA = tA / RB, iA = tA - rowA * RB;
  const bool actB = tid < ROWS * RA;
  const int tB = actB ? tid : ROWS * RA - 1;
  const int rowB = tB / RA, pB = tB - rowB * RA;
  cf* const rowa = cube + rowA * rs + iA;
  cf* const rowb = cube + rowB * rs + pB * RB;
  const rsrc_t tw1 = make_rsrc(a.tw1, (size_t)ROWS * H * sizeof(cf));
  const rsrc_t win = make_rsrc(a.win, (size_t)WH * H * sizeof(float));
  const unsigned npr4 = (unsigned)npr * 4u, npr8 = (unsigned)npr * 8u;
  const long long nframes = (long long)a.B * a.T;
  const int fs = a.fs_plain, n_stft = a.g.n_stft;

  __shared__ __attribute__((aligned(16))) cf twa_lds[TWL ? RB * (RA - 1) : 1];
  if (TWL)
    for (int i = tid; i < RB * (RA - 1); i += NT) twa_lds[i] = a.twa[i];
  __syncthreads();  // as in fam_gl_kernel: the table must be complete before the first pass A
  FamTwA<RA, RB, TWL> wa;
  wa.src = make_rsrc(a.twa, (size_t)RB * (RA - 1) * sizeof(cf));
  wa.voff = (unsigned)iA * 8u;
  wa.tab = twa_lds + iA;
  cf w1[12];
  float wv[WH], u[WH];
  auto g1 = [&w1](int k) { return fam_g_pow(w1, k); };
  auto load_frame_inputs = [&](long long gf) {  // tables and the frame's samples (reflect-padded like torch.stft center=True)
#pragma unroll
    for (int k = 1; k <= (NR == 40 ? 11 : 10); ++k) {
      const v2f t = ld2(tw1, npr8, (unsigned)(k <= 10 ? k : 20) * (H * 8u));
      w1[k] = cf{t.x, t.y};
    }
    const int clip = (int)(gf / a.T), fr = (int)(gf - (long long)clip * a.T);
    const rsrc_t x = make_rsrc(a.wave + (size_t)clip * a.wave_stride, (size_t)a.Lw * sizeof(float));
#pragma unroll
    for (int j = 0; j < WH; ++j) {
      wv[j] = ld1(win, npr4, (unsigned)j * (H * 4u));
      u[j] = ld1(x, (unsigned)reflect_index(a.g.hop * fr + a.g.off + j * H + npr, a.Lw) * 4u, 0);
    }
  };
  if ((long long)blockIdx.x < nframes) load_frame_inputs(blockIdx.x);
  for (long long gf = blockIdx.x; gf < nframes; gf += gridDim.x) {
#pragma unroll
    for (int j = 0; j < WH; ++j) u[j] *= wv[j];
    if (act1) fam_p1_forward_store<NR>(u, g1, cube, col, rs);
    RFX_SCHED_FENCE();
    wa.template load<false, 0, true>();
    __syncthreads();
    RFX_SCHED_FENCE();
    if (actA)
      fam_pass_a_forward<RA, RB, true>(rowa, 0, [&wa](int p) { return wa.w[p]; }, [&wa](int batch) {
        wa.template load_batch<false, true>(batch);
        RFX_SCHED_FENCE();
      });
    RFX_SCHED_FENCE();
    __syncthreads();
    RFX_SCHED_FENCE();
    cf R[RB];
    fam_pass_b_forward<RA, RB, VEC>(rowb, 0, R);
    RFX_SCHED_FENCE();
    __syncthreads();  // every row has been read: the memory is now the bin-ordered frame
    if (actB) {
      // slot s of this thread holds k = k0 + 40 RA s.  k0 is frame-invariant, and the compiler would hoist all RB bin positions
      // and conjugate flags out of the frame loop - into scratch at the 128-register bound (68 spilled registers in round 3):
      // the empty asm makes k0 opaque per frame, two integer instructions per slot instead
      int k0 = rowB + NR * pB;
      asm volatile("" : "+v"(k0));
#pragma unroll
      for (int s = 0; s < RB; ++s) {
        const int k = k0 + NR * RA * s;
        const bool cj = k > a.g.n_fft / 2;
        const int bin = cj ? a.g.n_fft - k : k;
        // rows 0 and nrad / 2 hold their bins twice: the direct slot writes, the other one stores into a dump entry past the
        // frame (a select instead of a branch per slot: twenty s_and_saveexec / s_cbranch pairs per frame in round 3)
        const int at = fam_slot_is_primary(NR, rowB, cj) ? bin : n_stft + (tid & 31);
        if (MODE == 1) cube[at] = cf{R[s].re, cj ? -R[s].im : R[s].im};
        // v_sqrt_f32 (1 ulp) instead of the IEEE expansion: |X| carries ~1e-7 relative error from the transform anyway
        else magl[at] = __builtin_amdgcn_sqrtf(fmaf(R[s].re, R[s].re, R[s].im * R[s].im));
      }
    }
    RFX_SCHED_FENCE();
    __syncthreads();
    if (gf + gridDim.x < nframes) load_frame_inputs(gf + gridDim.x);  // in flight underneath the output loop
    if (MODE == 2) {
      // banded mel projection straight from the frame in LDS (the reference multiplies the dense filterbank,
      // spectrogram_converter.py:76-84, :185): a filter's weights come eight at a time from the L2-resident table, summed in
      // increasing bin order like gen_mel_kernel
      // (buffer-descriptor addressing: SRD + one 32-bit lane offset - the 64-bit per-lane pointers of the first version were what the
      // 48 kHz kernel spilled inside its frame loop)
      const rsrc_t bw = make_rsrc(a.band_wt, 0xFFFFFFFFull), blo = make_rsrc(a.band_lo, (size_t)a.Mpad * 4), bln = make_rsrc(a.band_len, (size_t)a.Mpad * 4);
      const rsrc_t mrow = make_rsrc(a.mel_tm + (size_t)gf * a.Mpad, (size_t)a.Mpad * 4);
      const unsigned mstride = (unsigned)a.Mpad * 4u;
      for (int m = tid; m < a.Mpad; m += NT) {
        float acc = 0.f;
#ifndef RFX_ABL_FAM_NOMEL  // (ablation: what the sum phase costs)
        if (m < a.M) {
          const int lo = (int)ld1u(blo, (unsigned)m * 4u, 0), n = (int)ld1u(bln, (unsigned)m * 4u, 0);
          for (int i = 0; i < n; i += 8) {
            float w[8], v[8];
            const unsigned woff = (unsigned)m * 4u + (unsigned)i * mstride;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              w[e] = ld1(bw, woff + (unsigned)e * mstride, 0);
              const int q = lo + i + e;
              v[e] = magl[q < n_stft ? q : n_stft - 1];  // (rows past the filter's end carry zero weights)
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(w[e], v[e], acc);
          }
        }
#else
        acc = magl[m];
#endif
        st1(acc, mrow, (unsigned)m * 4u, 0);
      }
    } else if (MODE == 1) {
      cf* __restrict__ out = a.spec + (size_t)gf * fs;
      for (int k = tid; k < fs; k += NT) out[k] = k < n_stft ? cube[k] : cf{0.f, 0.f};
    } else {
      float* __restrict__ out = a.mag + (size_t)gf * fs;
      for (int k = tid; k < fs; k += NT) out[k] = k < n_stft ? magl[k] : 0.f;  // (the padding of the frame stride stays zero)
    }
    RFX_SCHED_FENCE();
    __syncthreads();  // the next frame's P1 overwrites the frame
  }
}

// |S| (or any per-bin float array) from plain bin order [nframes][fs_plain] to slot order [nframes][fsf]: one workgroup per
// frame, the row staged in LDS (whole-line loads; the first version gathered 4-byte values 160 bytes apart straight from
// global memory: 1.39 ms per 64 x 512 frames at 48 kHz), whole-line stores
__global__ void __launch_bounds__(512) fam_repack_kernel(const float* __restrict__ plain, float* __restrict__ slots,
                                                         const int* __restrict__ bin_of, int fs_plain, int fsf, int n_stft) {
  extern __shared__ float row_s[];  // [n_stft]
  const size_t fr = blockIdx.x;
  const float* __restrict__ row = plain + fr * fs_plain;
  for (int i = threadIdx.x; i < n_stft; i += blockDim.x) row_s[i] = row[i];
  __syncthreads();
  float* __restrict__ out = slots + fr * fsf;
  for (int i = threadIdx.x; i < fsf; i += blockDim.x) {
    const int b = bin_of[i];
    out[i] = b >= 0 ? row_s[b] : 0.f;
  }
}

#endif  // RFX_FAM_TU == 0

using FamGlFn = void (*)(FamGlArgs);
template <int RA, int RB, int NR = 40>
static FamGlFn fam_fn(int mode) {
  return mode == 0 ? fam_gl_kernel<0, RA, RB, NR> : fam_gl_kernel<1, RA, RB, NR>;  // (modes 1 and 2 are one kernel since the fold forms d)
}
#if RFX_FAM_TU == 1
// unit 1: the packed Griffin-Lim kernels (every geometry but 48 kHz)
FamGlFn fam_gl_fn_packed(const FamGeom& g, int mode) {
  if (g.nrad == 20) return g.h == 441 ? fam_fn<21, 21, 20>(mode) : nullptr;  // 22.05 kHz
  switch (g.h) {
    case 80: return fam_fn<10, 8>(mode);
    case 160: return fam_fn<16, 10>(mode);
    case 240: return fam_fn<16, 15>(mode);
    case 320: return fam_fn<20, 16>(mode);
    case 441: return fam_fn<21, 21>(mode);
    default: return nullptr;
  }
}
#else
FamGlFn fam_gl_fn_packed(const FamGeom& g, int mode);  // rfx_fam_pk.hip
static FamGlFn fam_fn(const FamGeom& g, int mode) {
#if !defined(RFX_NO_PK) && !defined(RFX_FAM_PK)
  if (g.h != 480) return fam_gl_fn_packed(g, mode);
#else
  if (g.nrad == 20) return g.h == 441 ? fam_fn<21, 21, 20>(mode) : nullptr;  // 22.05 kHz
#endif
  switch (g.h) {
#if defined(RFX_NO_PK) || defined(RFX_FAM_PK)
    case 80: return fam_fn<10, 8>(mode);
    case 160: return fam_fn<16, 10>(mode);
    case 240: return fam_fn<16, 15>(mode);
    case 320: return fam_fn<20, 16>(mode);
    case 441: return fam_fn<21, 21>(mode);
#endif
    case 480: return fam_fn<24, 20>(mode);
    default: return nullptr;
  }
}

using FamFwdFn = void (*)(FamFwdArgs);
template <int RA, int RB, int NR = 40>
static FamFwdFn fam_fwd_fn(int mode) {
  return mode == 0 ? fam_fwd_kernel<0, RA, RB, NR> : mode == 1 ? fam_fwd_kernel<1, RA, RB, NR> : fam_fwd_kernel<2, RA, RB, NR>;
}
static FamFwdFn fam_fwd_fn(const FamGeom& g, int mode) {
  if (g.nrad == 20) return g.h == 441 ? fam_fwd_fn<21, 21, 20>(mode) : nullptr;  // 22.05 kHz
  switch (g.h) {
    case 80: return fam_fwd_fn<10, 8>(mode);
    case 160: return fam_fwd_fn<16, 10>(mode);
    case 240: return fam_fwd_fn<16, 15>(mode);
    case 320: return fam_fwd_fn<20, 16>(mode);
    case 441: return fam_fwd_fn<21, 21>(mode);
    case 480: return fam_fwd_fn<24, 20>(mode);
    default: return nullptr;
  }
}

hipError_t launch_fam_fwd(int mode, const FamFwdArgs& a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(fam_fwd_fn(a.g, mode), dim3(nblocks), dim3(a.g.nthr), fam_lds_bytes(a.g), stream, a);
  return hipGetLastError();
}

hipError_t prepare_fam_kernels(const FamGeom& g) {
  for (int mode = 0; mode < 3; ++mode) {
    const hipError_t e = hipFuncSetAttribute((const void*)fam_fwd_fn(g, mode), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fam_lds_bytes(g));
    if (e != hipSuccess) return e;
  }
  for (int mode = 0; mode < 2; ++mode) {
    const FamGlFn fn = fam_fn(g, mode);
    if (!fn) return hipErrorInvalidValue;
    const hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fam_lds_bytes(g));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

int fam_blocks_per_cu(const FamGeom& g) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)fam_fn(g, 1), g.nthr, fam_lds_bytes(g)) != hipSuccess || n < 1) n = 1;  // (counts the static LDS too)
  return n;
}

hipError_t launch_fam_gl(int mode, const FamGlArgs& a, int nblocks, hipStream_t stream) {
  hipLaunchKernelGGL(fam_fn(a.g, mode), dim3(nblocks), dim3(a.g.nthr), fam_lds_bytes(a.g), stream, a);
  return hipGetLastError();
}

hipError_t launch_fam_repack(const float* plain, float* slots, const int* bin_of, long long nframes, int fs_plain, int fsf, int n_stft,
                             hipStream_t stream) {
  hipLaunchKernelGGL(fam_repack_kernel, dim3((unsigned)nframes), dim3(512), sizeof(float) * (size_t)n_stft, stream, plain, slots, bin_of, fs_plain, fsf, n_stft);
  return hipGetLastError();
}

#endif  // RFX_FAM_TU

}  // namespace rfx
