// rfx_stft.hip - forward framed transform (replaces torchaudio.transforms.Spectrogram(power=None) +
// torch.abs, riffusion/spectrogram_converter.py:47-59, :179-182) and the layout converters between
// the reference's (B, n_stft, T) tensors and the slot-major frames the gfx950 kernels stream.
#define RFX_PK 1  // packed fp32 butterflies (rfx_core.h)
#include "rfx_frame.hip.h"
#include "rfx_kernels.h"

namespace rfx {

// one thread per slot position, 16 consecutive frames: reads 64 contiguous bytes of a bin row and
// writes position-contiguous (coalesced) floats into 16 frames
template <bool COMPLEX>
__global__ void __launch_bounds__(256) pack_kernel(const void* __restrict__ src_, void* __restrict__ dst_, int T) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= kFrameStride) return;
  const int tchunk = blockIdx.y * 16;
  const int clip = blockIdx.z;
  int q, kb;
  const bool real_slot = COMPLEX ? pos_c_to_slot(p, q, kb) : pos_f_to_slot(p, q, kb);
  if (!real_slot) {  // padding lane: keep it zero
    const int ntp = min(16, T - tchunk);
    for (int i = 0; i < ntp; ++i) {
      const size_t o = ((size_t)clip * T + tchunk + i) * kFrameStride + p;
      if (COMPLEX) reinterpret_cast<cf*>(dst_)[o] = cf{0.f, 0.f}; else reinterpret_cast<float*>(dst_)[o] = 0.f;
    }
    return;
  }
  bool cj;
  const int bin = slot_bin(q / 21, q % 21, kb, &cj);
  const int nt = min(16, T - tchunk);
  if (COMPLEX) {
    const cf* src = reinterpret_cast<const cf*>(src_) + ((size_t)clip * kBins + bin) * T + tchunk;
    cf* dst = reinterpret_cast<cf*>(dst_) + ((size_t)clip * T + tchunk) * kFrameStride + p;
    for (int i = 0; i < nt; ++i) {
      cf v = src[i];
      if (cj) v.im = -v.im;
      dst[(size_t)i * kFrameStride] = v;
    }
  } else {
    const float* src = reinterpret_cast<const float*>(src_) + ((size_t)clip * kBins + bin) * T + tchunk;
    float* dst = reinterpret_cast<float*>(dst_) + ((size_t)clip * T + tchunk) * kFrameStride + p;
    for (int i = 0; i < nt; ++i) dst[(size_t)i * kFrameStride] = src[i];
  }
}

// slots -> (B, n_stft, T) complex, reading each bin from its primary slot (test / debugging path)
__global__ void __launch_bounds__(256) unpack_complex_kernel(const cf* __restrict__ slots, cf* __restrict__ out, int T) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= kBins) return;
  const int tchunk = blockIdx.y * 16;
  const int clip = blockIdx.z;
  int k = bin;
  bool cj = false;
  if (k % 40 > 20) { k = kNfft - k; cj = true; }
  const int k1 = k % 40, kp = k / 40;
  const int ka = kp % 21, kb = kp / 21;
  const int p = slot_pos_c(k1 * 21 + ka, kb);
  const int nt = min(16, T - tchunk);
  for (int i = 0; i < nt; ++i) {
    cf v = slots[((size_t)clip * T + tchunk + i) * kFrameStride + p];
    if (cj) v.im = -v.im;
    out[((size_t)clip * kBins + bin) * T + tchunk + i] = v;
  }
}

// float slots -> (B, n_stft, T), reading each bin from its primary slot
__global__ void __launch_bounds__(256) unpack_mag_kernel(const float* __restrict__ slots, float* __restrict__ out, int T) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= kBins) return;
  const int tchunk = blockIdx.y * 16;
  const int clip = blockIdx.z;
  int k = bin;
  if (k % 40 > 20) k = kNfft - k;
  const int k1 = k % 40, kp = k / 40;
  const int p = slot_pos_f(k1 * 21 + kp % 21, kp / 21);
  const int nt = min(16, T - tchunk);
  for (int i = 0; i < nt; ++i)
    out[((size_t)clip * kBins + bin) * T + tchunk + i] = slots[((size_t)clip * T + tchunk + i) * kFrameStride + p];
}
hipError_t launch_unpack_mag(const float* slots, float* out_bft, int B, int T, hipStream_t stream) {
  dim3 grid((kBins + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(unpack_mag_kernel, grid, dim3(256), 0, stream, slots, out_bft, T);
  return hipGetLastError();
}

hipError_t launch_pack_mag(const float* lin_bft, float* S_slots, int B, int T, hipStream_t stream) {
  dim3 grid((kFrameStride + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(pack_kernel<false>, grid, dim3(256), 0, stream, (const void*)lin_bft, (void*)S_slots, T);
  return hipGetLastError();
}
hipError_t launch_pack_angles(const cf* ang_bft, cf* slots, int B, int T, hipStream_t stream) {
  dim3 grid((kFrameStride + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(pack_kernel<true>, grid, dim3(256), 0, stream, (const void*)ang_bft, (void*)slots, T);
  return hipGetLastError();
}
hipError_t launch_unpack_complex(const cf* slots, cf* out_bft, int B, int T, hipStream_t stream) {
  dim3 grid((kBins + 255) / 256, (T + 15) / 16, B);
  hipLaunchKernelGGL(unpack_complex_kernel, grid, dim3(256), 0, stream, slots, out_bft, T);
  return hipGetLastError();
}

// ---- forward STFT: frame t of clip b is centred on sample 441*t of the reflect-padded waveform
__global__ void __launch_bounds__(kThreads) stft_kernel(StftArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, a.tw1, a.tw2);
  const float* __restrict__ winp = a.win + t.npr;
  __syncthreads();

  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  const int clip = blockIdx.x / chunks;
  const int f0 = (blockIdx.x - clip * chunks) * a.frames_per_block;
  const int f1 = min(a.T, f0 + a.frames_per_block);
  const float* __restrict__ x = a.wave + (size_t)clip * a.Lw;

  for (int fr = f0; fr < f1; ++fr) {
    float u[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int p = reflect_index((fr + j - kHalfHops) * kHop + t.npr, a.Lw);
      u[j] = x[p] * winp[j * kHop];
    }
    cf R[21];
    frame_forward(u, R, f, t, [] {});
    const size_t fbase = ((size_t)clip * a.T + fr) * kFrameStride;
    // padded owner index of this thread's slots; seven idle lanes store zeros to the padding positions 63, 127, ...
    // (keeps the mel GEMM free of garbage)
    const bool store = t.active || t.pad >= 0;
    const int q = t.active ? slot_qp(t.npr) : 64 * t.pad + 63;
    if (a.mag && store) {
      float m[21];
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) m[kb] = t.active ? sqrtf(fmaf(R[kb].re, R[kb].re, R[kb].im * R[kb].im)) : 0.f;
      float4* d4 = reinterpret_cast<float4*>(a.mag + fbase);
#pragma unroll
      for (int i = 0; i < 5; ++i) d4[i * kQPad + q] = float4{m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]};
      a.mag[fbase + 20 * kQPad + q] = m[20];
    }
    if (a.spec && store) {
      if (!t.active) {
#pragma unroll
        for (int kb = 0; kb < 21; ++kb) R[kb] = cf{0.f, 0.f};
      }
      float4* d4 = reinterpret_cast<float4*>(a.spec + fbase);
#pragma unroll
      for (int i = 0; i < 10; ++i)
        d4[i * kQPad + q] = float4{R[2 * i].re, R[2 * i].im, R[2 * i + 1].re, R[2 * i + 1].im};
      a.spec[fbase + 20 * kQPad + q] = R[20];
    }
    __syncthreads();  // the next frame's P1 overwrites rows other waves may still be reading in P3
  }
}

// ---- fused forward path (replaces Spectrogram(power=None) -> abs -> MelScale, spectrogram_converter.py:165-185):
// the frame engine as above; the magnitudes of the frame are then parked in LDS (each thread's 21 in the cube
// elements it has just consumed) and every thread forms the mel amplitudes of one or two filters as the banded dot product
//     mel[m] = sum_{i < band_len[m]} fb[band_lo[m] + i][m] * |X[band_lo[m] + i]|
// i.e. the reference's `|X|^T @ fb` with the structural zeros of the triangular filterbank left out (7 976 of its
// 4.5 M products for the default bank), summed in increasing bin order.  The 1.2 GB magnitude stream and the dense
// GEMM of the unfused path disappear; only (B, M, T) floats are written.
__global__ void __launch_bounds__(kThreads, 4) stft_mel_kernel(StftMelArgs a) {  // 128 VGPRs: two workgroups per CU (it took 171 = one per CU without the bound)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, a.tw1, a.tw2);
  const float* __restrict__ winp = a.win + t.npr;
  float* magb = reinterpret_cast<float*>(smem);  // float view of the cube: slot (k1, ka, kb) -> magb[2 * cube_at(k1, ka, 0) + kb]
  __syncthreads();

  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  const int clip = blockIdx.x / chunks;
  const int f0 = (blockIdx.x - clip * chunks) * a.frames_per_block;
  const int f1 = min(a.T, f0 + a.frames_per_block);
  const float* __restrict__ x = a.wave + (size_t)clip * a.Lw;

  // the (up to two) filters of this thread: m0 = threadIdx, m1 = threadIdx + kThreads; waves walk bands of similar length
  const int m0 = threadIdx.x, m1 = threadIdx.x + kThreads;
  const bool has0 = m0 < a.M, has1 = m1 < a.M;
  const int n0 = has0 ? a.band_len[m0] : 0;
  const int n1 = has1 ? a.band_len[m1] : 0;
  float w10[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) w10[j] = winp[j * kHop];

  for (int fr = f0; fr < f1; ++fr) {
    float u[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int p = reflect_index((fr + j - kHalfHops) * kHop + t.npr, a.Lw);
      u[j] = x[p] * w10[j];
    }
    cf R[21];
    frame_forward(u, R, f, t, [] {});
    // P3 read exactly the 21 cube elements (k1, ka, 0..20) that only this thread touches: their memory takes the
    // thread's 21 magnitudes right away (no barrier, no bank-conflicted scatter); `band_addr` tells the mel threads where
    // the bins of their filter ended up
    if (t.active) {
      float* own = magb + 2 * cube_at(t.k1, t.idx, 0);
#pragma unroll
      for (int kb = 0; kb < 21; ++kb) own[kb] = sqrtf(fmaf(R[kb].re, R[kb].re, R[kb].im * R[kb].im));
    }
    __syncthreads();
    {
      // eight weights / addresses in flight per step (tables zero-padded to a multiple of eight rows: the tail multiplies
      // a finite magnitude by zero; padding filters have length 0); the next step's are requested before the current eight
      // are summed; bins are summed in increasing order
      // frame-major scratch (512 contiguous floats per frame: whole-line stores); a tiled transpose brings it into the
      // reference's (B, M, T) layout afterwards - 4-byte stores T floats apart cost 10x the bytes in HBM writes
      float* __restrict__ row = a.mel_tm + ((size_t)clip * a.T + fr) * a.Mpad;
      // one code path for the (up to) two filters of a thread, run one after the other: its register arrays exist once
#pragma unroll 1
      for (int which = 0; which < 2; ++which) {
        const int m = which ? m1 : m0;
        if (m >= a.Mpad) break;  // wave-uniform: only the first wave(s) carry a second filter
        const int n = which ? n1 : n0;
        const float* __restrict__ wt = a.band_wt + m;
        const int* __restrict__ ad = a.band_addr + m;
        float s = 0.f;
        float wn[8];
        int an[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          wn[j] = wt[(size_t)j * a.Mpad];
          an[j] = ad[(size_t)j * a.Mpad];
        }
#pragma unroll 1
        for (int i = 0; i < n; i += 8) {
          float w[8], v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            w[j] = wn[j];
            v[j] = magb[an[j]];
          }
          if (i + 8 < n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              wn[j] = wt[(size_t)(i + 8 + j) * a.Mpad];
              an[j] = ad[(size_t)(i + 8 + j) * a.Mpad];
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) s = fmaf(w[j], v[j], s);
        }
        row[m] = m < a.M ? s : 0.f;
      }
    }
    __syncthreads();  // the next frame's P1 overwrites the magnitudes
  }
}

// ---- fused forward path, product form.  What changed against stft_mel_kernel above, and why:
//  * the transform half follows the Griffin-Lim kernel's analysis half: a workgroup walks a RUN of consecutive frames, and
//    everything a frame's P1 needs - its ten input samples, the g(n')^k1 twiddles, the Hann samples - is requested during
//    the mel phase of the previous frame (stft_mel_kernel fetched ten samples and twenty twiddles per frame and waited for
//    them on the spot);
//  * the mel phase reads no tables: the thread that owns a bin's primary slot holds |X| in a register anyway, multiplies it
//    by the bin's (at most two) filterbank weights - frame-invariant, streamed from an L2-resident per-slot table under P3
//    like |S| in the Griffin-Lim kernel - and scatters the two products into LDS arrays prod0 / prod1 laid out GROUP by
//    group (group g = the bins whose first filter is g, in bin order, padded with zeros to a multiple of four; the cube is
//    dead by then).  Filter m is  sum(prod1 over group m-1) + sum(prod0 over group m): its rising part followed by its
//    falling part, i.e. still summed in increasing bin order - contiguous 16-byte LDS reads and plain adds, no masks per
//    product.  stft_mel_kernel fetched a weight and an address per product from L2 in dependent steps of eight (7 976 x 2
//    loads per frame and workgroup).
// Cost: two more workgroup barriers per frame (all waves must have left P3 before the cube is overwritten with products).
// Measured (B = 64, T = 512, round 3): 0.83 -> 0.69 ms.
// KBMASK: the kb (of a thread's 21 slots) that can contribute, as a compile-time set (rfx_kernels.h: kKbMaskLow / kKbMaskAll).
// Round 5: what the mel half costs is its TABLE TRAFFIC, not the exchange (profiles/r05_forward_ablation.txt): every frame each
// thread re-fetches 392 B of frame-invariant constants from L2 (no registers to keep them: 128 VGPRs), 152 B of them for the mel
// phase; without the mel tables the kernel takes 0.45 ms, with them and no exchange / no sums 0.535, complete 0.58.  PK (the
// default-bank form, KBMASK == kKbMaskLow and a plan that holds the packed tables): product positions and padding positions as
// 16-bit LDS byte addresses (two per dword), both segments of a filter in one 8-byte load, the second array at a compile-time
// offset (one address register serves both stores), the second filter's segments fetched by the first wave only; and for
// every form the g(n')^k1 twiddles of the rows 11..19 are derived from those of the rows 1..10 and 20 (g^(20-k) = g^20 conj(g^k),
// as the row-family kernels do: nine packed complex products instead of 72 B per thread and frame).  With the twiddle table
// shorter by 18 registers the ten-slot forms also keep their Hann samples in registers and slide their input window (one new sample
// per frame): 208 B in 29 loads per thread and frame instead of 392 B in 66, 0.554 -> 0.503 ms, 118 VGPRs, no scratch.
#ifndef RFX_FWD_ABL
#define RFX_FWD_ABL 0  // timing ablations of -DRFX_ABLATION builds (wrong results): 2 no product scatter, 3 no segment sums, 4 transform +
#endif                 // table fetches only, 5 no mel tables either, 7 no slot weights, 8 no twiddle fetches (profiles/r05_forward_ablation.txt)
#ifndef RFX_FWD_OPT
#define RFX_FWD_OPT 15  // what each bit buys is in profiles/r05_forward_ablation.txt; bit 4 (0.503 -> 0.500 ms) is off: it costs the nine
#endif                  // registers the running maximum of rfx_image_from_waveform needs
constexpr bool kFwdTwShort = (RFX_FWD_OPT & 2) != 0;   // eleven twiddles fetched, nine derived (bit 0: the packed tables, see launch_stft_mel)
constexpr bool kFwdWinRegs = (RFX_FWD_OPT & 4) != 0;   // ten-slot forms: the ten Hann samples stay in registers over the run
constexpr bool kFwdSlide = (RFX_FWD_OPT & 8) != 0;     // ten-slot forms: sliding input window, one new sample per frame
constexpr bool kFwdIdxRegs = (RFX_FWD_OPT & 16) != 0;  // PK: the packed positions / padding / segments stay in registers over the run
template <unsigned KBMASK, bool PK>
__global__ void __launch_bounds__(kThreads, 4) stft_mel2_kernel(StftMelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ThreadId t = thread_id();
  const FrameCtx f = frame_ctx(smem, t, a.tw1, a.tw2);
  float* prod = reinterpret_cast<float*>(smem);  // [0, kQPad): one dump float per lane, then w0 * |X|, group-padded; the same for w1 * |X| from prod_arr on
  constexpr int NKB = __builtin_popcount(KBMASK);
  constexpr bool TWS = kFwdTwShort;
  constexpr bool WIN_REGS = kFwdWinRegs && NKB <= 10, SLIDE = kFwdSlide && NKB <= 10;  // (the 21-slot form has no registers for them: 94 scratch instructions)
  static_assert(!PK || (KBMASK == kKbMaskLow && NKB == 10), "the packed tables hold ten slots per thread");

  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  int clip = blockIdx.x / chunks;
  int chunk = blockIdx.x - clip * chunks;  // index of this workgroup's run among the clip's `chunks` runs
  int f0 = chunk * a.frames_per_block;
  int f1 = min(a.T, f0 + a.frames_per_block);
  if (a.run_skew > 0) {
    // Round 6: frames are independent here, so the runs need not be equal.  With exactly two workgroups per CU the blocks of the
    // first half of the grid are the ones the dispatcher places first (one per CU); they win the CU's issue arbitration and finish
    // early (profiles/r05_wgclock_dispatch_order.txt: the same frame engine in the Griffin-Lim kernel, 659 against 746 us), so they
    // take `run_skew` frames more and their partners as many fewer: every clip's first chunks/2 runs are long ones (host: chunks
    // even, T == chunks * frames_per_block, gridDim.x == 2 x CUs).
    const int half = gridDim.x >> 1, hc = chunks >> 1;
    const bool lng = (int)blockIdx.x < half;
    const int idx = lng ? (int)blockIdx.x : (int)blockIdx.x - half;
    clip = idx / hc;
    const int c = idx - clip * hc, fl = a.frames_per_block + a.run_skew, fs = a.frames_per_block - a.run_skew;
    f0 = lng ? c * fl : hc * fl + c * fs;
    f1 = f0 + (lng ? fl : fs);
    chunk = lng ? c : hc + c;
  }
  const rsrc_t xin = make_rsrc(a.wave + (size_t)clip * a.Lw, (size_t)a.Lw * 4);
  const rsrc_t win = make_rsrc(a.win, kWin * 4);
  const rsrc_t slots = make_rsrc(a.slot_tab, 21 * kQPad * 8);
  // Frame-invariant per-thread constants that are only needed in the short mel phase (where the products go, the segments
  // of the thread's filters, its share of the zero padding) are re-fetched from their L2-resident tables every frame, in
  // flight across the barrier that precedes their use: held in registers they are spilled around the transform and reloaded
  // on the spot (seen in the ISA).
  const rsrc_t slotat = PK ? make_rsrc(a.pk_at, 5 * kQPad * 4) : make_rsrc(a.slot_at, 21 * kQPad * 4);
  const rsrc_t segsrc = PK ? make_rsrc(a.pk_seg, (size_t)2 * a.Mpad * 4) : make_rsrc(a.filt_seg, (size_t)2 * a.Mpad * 4);
  const rsrc_t padsrc = PK ? make_rsrc(a.pk_pad, 2 * kQPad * 4) : make_rsrc(a.pad_tab, (size_t)kMelPadsPerThread * kQPad * 4);
  const unsigned npr4 = (unsigned)t.npr * 4u;
  const unsigned qp4 = (unsigned)slot_qp(t.npr) * 4u;
  const bool second = kThreads < a.Mpad && threadIdx.x < 64;  // wave-uniform: the first wave carries the filters past kThreads

  // the thread's ten Hann samples: like the twiddles, fetched during the mel phase of the previous frame
  float w10[10];
  auto load_window = [&] {
#pragma unroll
    for (int j = 0; j < 10; ++j) w10[j] = ld1(win, npr4, (unsigned)j * (kHop * 4u));
  };
  load_window();
  auto load_x = [&](int blk) { return ld1(xin, (unsigned)reflect_index(blk * kHop + t.npr, a.Lw) * 4u, 0); };
  // the ten samples of a frame are fetched (L2) with the twiddles during the previous frame's mel phase.  (A register sliding
  // window with one new sample per frame, as in the Griffin-Lim kernel, costs ten registers across the transform: 11 spilled
  // VGPRs and 0.69 instead of 0.65 ms per 64 waveforms - measured, bit-identical results.)
  float d[10];
  auto load_frame = [&](int fr) {
#pragma unroll
    for (int j = 0; j < 10; ++j) d[j] = load_x(fr + j - kHalfHops);
  };
  auto next_frame = [&](int fr) {  // the samples of frame fr, given those of frame fr - 1
    if (SLIDE) {
#pragma unroll
      for (int j = 0; j < 9; ++j) d[j] = d[j + 1];
      d[9] = load_x(fr + 9 - kHalfHops);
    } else {
      load_frame(fr);
    }
  };
  load_frame(f0);
  Tw1 tw1;
  load_tw1<TWS>(tw1, f);
  // where the products go, this thread's zero-padding positions, and the segments of its (up to two) filters - threadIdx
  // and threadIdx + kThreads (the first wave carries the second filters: they are the LONGEST bands, its first filters
  // the shortest).  Plain form: float positions, one per dword; PK: byte addresses, two per dword (sat[i] = slots 2 i and
  // 2 i + 1 of the thread's ten).
  unsigned sat[PK ? 5 : 21], pad_at[PK ? 2 : kMelPadsPerThread], seg[2][2] = {{0u, 0u}, {0u, 0u}};
  auto load_pk_tables = [&] {
#pragma unroll
    for (int i = 0; i < 5; ++i) sat[i] = ld1u(slotat, qp4, (unsigned)i * (kQPad * 4u));
    const v2u pp = ld2u(padsrc, 2u * qp4, 0);
    pad_at[0] = pp.x;
    pad_at[1] = pp.y;
    const v2u s0 = ld2u(segsrc, threadIdx.x * 8u, 0);
    seg[0][0] = s0.x;
    seg[0][1] = s0.y;
    if (second) {
      const v2u s1 = ld2u(segsrc, (threadIdx.x + kThreads) * 8u, 0);
      seg[1][0] = s1.x;
      seg[1][1] = s1.y;
    }
  };
  if constexpr (PK && kFwdIdxRegs) load_pk_tables();
  unsigned runmax = 0;  // key of the largest mel amplitude this thread has formed (0: below every value)
  __syncthreads();  // tw2 table in LDS

  for (int fr = f0; fr < f1; ++fr) {
    float u[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) u[j] = d[j] * w10[j];
    cf R[21];
    float sw0[21], sw1[21];  // per contributing slot: weights of its bin on its first / second filter, then their products with |X|
    frame_forward_tw<TWS>(u, R, f, t, tw1,
                     NoHook(),
                     NoHook(),
                     [&] {  // before P3 (P2's registers are free): the slots' weights fly under P3
#pragma unroll
                       for (int kb = 0; kb < 21; ++kb)
                         if (RFX_FWD_ABL >= 5 && RFX_FWD_ABL != 8) {  // (7: no slot weights only; 8: no twiddle fetches only)
                           sw0[kb] = 1.f;
                           sw1[kb] = 2.f;
                         } else if ((KBMASK >> kb) & 1u) {
                           const v2f e = ld2(slots, 2u * qp4, (unsigned)kb * (kQPad * 8u));
                           sw0[kb] = e.x;
                           sw1[kb] = e.y;
                         }
                     });
#pragma unroll
    for (int kb = 0; kb < 21; ++kb)
      if ((KBMASK >> kb) & 1u) {
        // v_sqrt_f32 (1 ulp) instead of the IEEE expansion: |X| carries ~1e-7 relative error from the transform anyway
        const float mag = __builtin_amdgcn_sqrtf(fmaf(R[kb].re, R[kb].re, R[kb].im * R[kb].im));
        sw0[kb] *= mag;
        sw1[kb] *= mag;
      }
    // the mel phase's positions: needed right behind the next two barriers, in flight across them
    if constexpr (!(PK && kFwdIdxRegs)) {
#if RFX_FWD_ABL == 5 || RFX_FWD_ABL == 6
#pragma unroll
    for (int i = 0; i < (PK ? 5 : 21); ++i) sat[i] = 0;
#pragma unroll
    for (int i = 0; i < (PK ? 2 : kMelPadsPerThread); ++i) pad_at[i] = 0;
#else
    if constexpr (PK) {
      load_pk_tables();
    } else {
#pragma unroll
      for (int kb = 0; kb < 21; ++kb)
        if ((KBMASK >> kb) & 1u) sat[kb] = ld1u(slotat, qp4, (unsigned)kb * (kQPad * 4u));
#pragma unroll
      for (int i = 0; i < kMelPadsPerThread; ++i) pad_at[i] = ld1u(padsrc, qp4, (unsigned)i * (kQPad * 4u));
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          seg[w][h] = ld1u(segsrc, (threadIdx.x + w * kThreads) * 4u, (unsigned)h * (unsigned)a.Mpad * 4u);
    }
#endif
    }
#if RFX_FWD_ABL >= 4 && RFX_FWD_ABL <= 6  // ablation: the transform and the table fetches only (no exchange, no sums; one barrier before the next P1)
    {
      float s = 0.f;
#pragma unroll
      for (int kb = 0; kb < 21; ++kb)
        if ((KBMASK >> kb) & 1u) s += sw0[kb] + sw1[kb];
#pragma unroll
      for (int i = 0; i < (PK ? 5 : 21); ++i)
        if (PK || ((KBMASK >> i) & 1u)) s += __builtin_bit_cast(float, sat[i]);
#pragma unroll
      for (int i = 0; i < (PK ? 2 : kMelPadsPerThread); ++i) s += __builtin_bit_cast(float, pad_at[i]);
      s += __builtin_bit_cast(float, seg[0][0] + seg[0][1] + seg[1][0] + seg[1][1]);
      load_tw1<TWS>(tw1, f);
      if (!WIN_REGS) load_window();
      next_frame(fr + 1);
      if (threadIdx.x < a.Mpad) a.mel_tm[((size_t)clip * a.T + fr) * a.Mpad + threadIdx.x] = s;
      __syncthreads();
      continue;
    }
#endif
    __syncthreads();  // every wave has left P3: the cube may be overwritten
#ifndef RFX_NO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (t.active && RFX_FWD_ABL != 2) {  // slots that contribute nothing (duplicates, bins outside the bank) carry zero weights and the lane's dump position
      if constexpr (PK) {
        auto put = [&](unsigned byte_at, float p0, float p1) {
          float* at = reinterpret_cast<float*>(smem + byte_at);
          at[0] = p0;
          at[kMelProdArr] = p1;
        };
        int n = 0;
#pragma unroll
        for (int kb = 0; kb < 21; ++kb)
          if ((KBMASK >> kb) & 1u) {
            put((n & 1) ? sat[n >> 1] >> 16 : sat[n >> 1] & 0xffffu, sw0[kb], sw1[kb]);
            ++n;
          }
#pragma unroll
        for (int i = 0; i < kMelPadsPerThread; ++i) put((i & 1) ? pad_at[i >> 1] >> 16 : pad_at[i >> 1] & 0xffffu, 0.f, 0.f);
      } else {
#pragma unroll
        for (int kb = 0; kb < 21; ++kb)
          if ((KBMASK >> kb) & 1u) {
            prod[sat[kb]] = sw0[kb];
            prod[sat[kb] + a.prod_arr] = sw1[kb];
          }
#pragma unroll
        for (int i = 0; i < kMelPadsPerThread; ++i) {
          prod[pad_at[i]] = 0.f;
          prod[pad_at[i] + a.prod_arr] = 0.f;
        }
      }
    }
#if RFX_FWD_ABL == 2
    if (t.active) {  // (the registers stay alive through one dump store)
      float s = 0.f;
#pragma unroll
      for (int kb = 0; kb < 21; ++kb)
        if ((KBMASK >> kb) & 1u) s += sw0[kb] + sw1[kb];
#pragma unroll
      for (int i = 0; i < (PK ? 5 : 21); ++i)
        if (PK || ((KBMASK >> i) & 1u)) s += __builtin_bit_cast(float, sat[i]);
      prod[threadIdx.x & 63] = s;
    }
#endif
    load_tw1<TWS>(tw1, f);  // for the next frame's P1: in flight across the mel phase
    if (!WIN_REGS) load_window();
    next_frame(fr + 1);
    __syncthreads();
    {
      // Results go to a frame-major scratch (Mpad contiguous floats per frame: whole-line stores); a tiled transpose brings
      // it into the reference's (B, M, T) layout afterwards - 4-byte stores T floats apart cost 10x the bytes in HBM writes.
      float* __restrict__ row = a.mel_tm + ((size_t)clip * a.T + fr) * a.Mpad;
      const v4f* prod4 = reinterpret_cast<const v4f*>(prod);
      auto add4 = [](float s, v4f x) { return (((s + x.x) + x.y) + x.z) + x.w; };
      // the tail of a segment longer than 16 products (wave-uniform trip count: the lanes of a wave hold neighbouring filters)
      auto tail = [&](float s, const v4f* p, int n4) {
        for (int e0 = 4; __builtin_amdgcn_ballot_w64(e0 < n4) != 0; e0 += 4) {
          v4f x[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] = p[e0 + j];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (e0 + j < n4) s = add4(s, x[j]);
        }
        return s;
      };
      // One filter: the first 16 products of its rising AND of its falling segment are requested together (one LDS round
      // trip; every lane issues them all: a lane whose segment is shorter reads into its neighbours' products or the dump
      // area - always inside the cube - and does not add them), then the sums: rising part first, in bin order.
      auto filter_sum = [&](unsigned sr, unsigned sf) {
        const v4f* pa = prod4 + (sr >> 6);  // segments start on 16-byte boundaries
        const v4f* pb = prod4 + (sf >> 6);
        const int na = sr & 15, nb4 = sf & 15;
        v4f ra[4], rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ra[j] = pa[j];
          rb[j] = pb[j];
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < na) s = add4(s, ra[j]);
        s = tail(s, pa, na);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nb4) s = add4(s, rb[j]);
        return tail(s, pb, nb4);
      };
#if RFX_FWD_ABL == 3  // ablation: no segment sums (one LDS read per thread keeps the exchange alive)
      const float s0 = prod[seg[0][0] >> 4] + prod[seg[0][1] >> 4];
      if (threadIdx.x < a.Mpad) row[threadIdx.x] = s0;
      if (second && threadIdx.x + kThreads < a.Mpad) row[threadIdx.x + kThreads] = prod[seg[1][0] >> 4] + prod[seg[1][1] >> 4];
#else
      const float s0 = filter_sum(seg[0][0], seg[0][1]);
      if (threadIdx.x < a.Mpad) row[threadIdx.x] = s0;  // padding filters (m >= M): empty segments, 0
      if (a.max_keys && threadIdx.x < a.M) runmax = max(runmax, max_key(s0));
      if (second) {
        const float s1 = filter_sum(seg[1][0], seg[1][1]);
        if (threadIdx.x + kThreads < a.Mpad) row[threadIdx.x + kThreads] = s1;
        if (a.max_keys && threadIdx.x + kThreads < a.M) runmax = max(runmax, max_key(s1));
      }
#endif
    }
    __syncthreads();  // the next frame's P1 overwrites the products
  }
  if (a.max_keys) {  // image_from_spectrogram's maximum (image_util.py:41), see StftMelArgs::max_keys
    // Round 6: one word per WORKGROUP, written exactly once per launch - no atomics, and no memset dispatch in front of the kernel;
    // the encoder takes the maximum over an image's C * chunks words.
    __shared__ unsigned wave_max[kWaves];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) runmax = max(runmax, (unsigned)__shfl_xor((int)runmax, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = runmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned m = wave_max[0];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) m = max(m, wave_max[w]);
      a.max_keys[(size_t)clip * chunks + chunk] = m;
    }
  }
}

// (B, T, Mpad) frame-major mel amplitudes -> the reference's (B, M, T): 64 x 64 tiles through LDS, both sides coalesced
__global__ void __launch_bounds__(256) mel_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int M, int Mpad) {
  __shared__ float tile[64][65];
  const int t0 = blockIdx.x * 64, m0 = blockIdx.y * 64, b = blockIdx.z;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  for (int r = ly; r < 64; r += 4) {
    const int tt = t0 + r;
    tile[r][lx] = tt < T ? in[((size_t)b * T + tt) * Mpad + m0 + lx] : 0.f;  // m0 + lx < Mpad always (Mpad % 64 == 0)
  }
  __syncthreads();
  for (int r = ly; r < 64; r += 4) {
    const int m = m0 + r, tt = t0 + lx;
    if (m < M && tt < T) out[((size_t)b * M + m) * T + tt] = tile[lx][r];
  }
}

hipError_t launch_mel_transpose(const float* mel_tm, float* mel, int B, int T, int M, int Mpad, hipStream_t stream) {
  hipLaunchKernelGGL(mel_transpose_kernel, dim3((T + 63) / 64, Mpad / 64, B), dim3(256), 0, stream, mel_tm, mel, T, M, Mpad);
  return hipGetLastError();
}

hipError_t launch_stft_mel(const StftMelArgs& a, hipStream_t stream) {
  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  const dim3 grid(a.B * chunks), block(kThreads);
  if (a.slot_tab && (a.kb_mask & ~kKbMaskLow) == 0 && a.pk_at && (RFX_FWD_OPT & 1))
    hipLaunchKernelGGL((stft_mel2_kernel<kKbMaskLow, true>), grid, block, kFrameDynLdsBytes, stream, a);
  else if (a.slot_tab && (a.kb_mask & ~kKbMaskLow) == 0)
    hipLaunchKernelGGL((stft_mel2_kernel<kKbMaskLow, false>), grid, block, kFrameDynLdsBytes, stream, a);
  else if (a.slot_tab)
    hipLaunchKernelGGL((stft_mel2_kernel<kKbMaskAll, false>), grid, block, kFrameDynLdsBytes, stream, a);
  else hipLaunchKernelGGL(stft_mel_kernel, grid, block, kFrameDynLdsBytes, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess || !a.mel) return e;  // (no (B, M, T) copy wanted: rfx_image_from_waveform encodes from the frame-major scratch)
  return launch_mel_transpose(a.mel_tm, a.mel, a.B, a.T, a.M, a.Mpad, stream);
}

hipError_t prepare_frame_kernels() {
  for (const void* fn : {(const void*)stft_kernel, (const void*)stft_mel_kernel, (const void*)stft_mel2_kernel<kKbMaskLow, true>,
                         (const void*)stft_mel2_kernel<kKbMaskLow, false>, (const void*)stft_mel2_kernel<kKbMaskAll, false>}) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kFrameDynLdsBytes);
    if (e != hipSuccess) return e;
  }
  return prepare_gl_kernels();
}

hipError_t launch_stft(const StftArgs& a, hipStream_t stream) {
  const size_t lds = kFrameDynLdsBytes;
  const int chunks = (a.T + a.frames_per_block - 1) / a.frames_per_block;
  hipLaunchKernelGGL(stft_kernel, dim3(a.B * chunks), dim3(kThreads), lds, stream, a);
  return hipGetLastError();
}

}  // namespace rfx
