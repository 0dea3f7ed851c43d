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
#include <stdlib.h>

namespace rfx {

constexpr int kImelThreads = 256;

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
  const int n_stft = a.n_stft;

  for (int i = tid; i < tb.nnz; i += kImelThreads) w_s[i] = tb.csr_w[i];
  if (tid == 0) diff_s[a.M] = 0.f;

  // ---- phase-B ownership: bins f = f_lo + tid + 256*j
  float spec[BPT], buf[BPT], w0[BPT], w1[BPT];
  int m0[BPT];
  const unsigned rbase = rand_frame_key(a.seed, a.frame_base + (unsigned long long)frame);
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    const int f = tb.f_lo + tid + kImelThreads * j;
    const bool ok = f < tb.f_hi;
    m0[j] = ok ? tb.bin_m0[f] : -1;
    w0[j] = ok ? tb.bin_w0[f] : 0.f;
    w1[j] = ok ? tb.bin_w1[f] : 0.f;
    spec[j] = ok ? (a.spec0 ? a.spec0[(size_t)frame * n_stft + f] : rand_unit(rbase, f)) : 0.f;
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
  int mlist[kMaxMelPerThread], pbeg[kMaxMelPerThread], pend[kMaxMelPerThread], soff[kMaxMelPerThread];
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
    pbeg[r] = m >= 0 ? tb.csr_ptr[m] : 0;
    pend[r] = m >= 0 ? tb.csr_ptr[m + 1] : 0;
    soff[r] = m >= 0 ? (tb.band_lo[m] - tb.f_lo) - pbeg[r] : 0;  // spec_s[soff + p] pairs with w_s[p]
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
      const int p1 = pend[r];
      const float* sp = spec_s + soff[r];
      // eight LDS pairs in flight per trip (the band length varies per mel: clamp + zero weight)
      float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
      for (int p = pbeg[r]; p < p1; p += 8) {
        float w[8], sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int q = min(p + i, p1 - 1);
          w[i] = (p + i < p1) ? w_s[q] : 0.f;
          sv[i] = sp[q];
        }
        acc0 = fmaf(w[0], sv[0], acc0); acc1 = fmaf(w[1], sv[1], acc1);
        acc2 = fmaf(w[2], sv[2], acc2); acc3 = fmaf(w[3], sv[3], acc3);
        acc0 = fmaf(w[4], sv[4], acc0); acc1 = fmaf(w[5], sv[5], acc1);
        acc2 = fmaf(w[6], sv[6], acc2); acc3 = fmaf(w[7], sv[7], acc3);
      }
      const float acc = (acc0 + acc1) + (acc2 + acc3);
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
  float* out = a.out_slots + (size_t)frame * a.out_stride;
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
    const float v = a.spec0 ? a.spec0[(size_t)frame * n_stft + f] : rand_unit(rbase, f);
    out[tb.bin_pos[f]] = v;
    const int p2 = tb.bin_pos2[f];
    if (p2 >= 0) out[p2] = v;
  }
  // padding positions are zeroed so that later consumers never see garbage
  if (a.plain) {
    for (int p = n_stft + tid; p < a.out_stride; p += kImelThreads) out[p] = 0.f;
  } else {
    for (int p = tid; p < kFrameStride; p += kImelThreads) {
      int q, kb;
      if (!pos_f_to_slot(p, q, kb)) out[p] = 0.f;
    }
  }
  if (a.loss_hist && !a.it_limit)
    for (int i = tid; i < a.max_iter; i += kImelThreads) a.loss_hist[(size_t)frame * a.max_iter + i] = i < steps ? hist_s[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// Fast path: group formulation.  Bins whose first filter is g form group g (contiguous, disjoint):
//   A_g = sum w0*spec (into filter g)      B_g = sum w1*spec (into filter g+1)      pred_m = A_m + B_{m-1}
// A thread owns one short low-frequency group and one long high-frequency group with all of their
// state (spec, momentum buffer, both weights) in registers; per step it publishes A/B (4 LDS
// writes), crosses ONE barrier, reads its neighbours' B_{g-1} / A_{g+1} (4 LDS reads) and forms the
// two residuals it needs itself.  Nothing else touches memory inside the 200-step loop.
// ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true);
  return x + __builtin_bit_cast(float, y);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_rows(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false);
  return x + __builtin_bit_cast(float, y);
}
// sum over the 64 lanes of a wave (wave-uniform result): 4 DPP adds inside the rows of 16, then the row sums travel up
// through lane 15 (row_bcast:15 into rows 1 and 3) and lane 31 (row_bcast:31 into rows 2 and 3); lane 63 holds the total
__device__ __forceinline__ float wave_sum(float x) {
  x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
  x = dpp_add<0x141>(x);  // row_half_mirror
  x = dpp_add<0x140>(x);  // row_mirror
#ifndef RFX_IMEL_READLANE_SUM
  x = dpp_add_rows<0x142, 0xA>(x);  // row_bcast:15
  x = dpp_add_rows<0x143, 0xC>(x);  // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
#else
  const int xi = __builtin_bit_cast(int, x);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48)));
#endif
}

// LDS of one frame of the group formulation: A and B double-buffered ([2][M + 4] each: entry m at index m + 1, zero pads
// at m = -1 and m = M, a dump entry for absent groups at m = M + 1 and its right neighbour), per-wave partial losses
// [max_iter][4]
RFX_HD size_t imel_group_lds_bytes(int M, int max_iter) { return sizeof(float) * (size_t)(4 * (M + 4) + 4 * max_iter); }
// ... followed by the epilogue's stage: the frame's active bins in bin order (round 5, see imel_emit_frame)
RFX_HD size_t imel_frame_lds_bytes(int M, int max_iter, int band) { return imel_group_lds_bytes(M, max_iter) + sizeof(float) * (size_t)((band + 3) & ~3); }

// The frame leaves in POSITION order, 16 bytes per lane, whole lines per wave-wide store (round 5).  Until then every kernel
// stored bin by bin: 9408 four-byte stores per frame to slot positions 336 B apart, 64 cache lines per wave-wide store - alone
// (max_iter = 1) the wave kernel took 2.05 ms per 64 tiles for 1.23 GB, and 1.2 ms of it stayed exposed behind the 200 steps
// (profiles/r05_imel_epilogue.txt).  Now the threads park the active bins in LDS in bin order (`stage`, entry f - f_lo; the caller
// synchronises between the two halves), then walk the frame's positions: pos_bin says which bin a position holds - from the stage
// if a filter reaches it, its initial value (passed through bit for bit) if none does, zero for padding.
__device__ __forceinline__ void imel_emit_frame(const ImelArgs& a, const float* stage, int frame, unsigned rbase, int tid, int nthr) {
  const ImelTables& tb = a.tb;
  float* out = a.out_slots + (size_t)frame * a.out_stride;
  auto value_at = [&](int bin) {
    if (bin < 0) return 0.f;
    if (bin >= tb.f_lo && bin < tb.f_hi) return stage[bin - tb.f_lo];
    return a.spec0 ? a.spec0[(size_t)frame * a.n_stft + bin] : rand_unit(rbase, bin);
  };
  if ((a.out_stride & 3) == 0) {
    const int4* __restrict__ pb4 = reinterpret_cast<const int4*>(tb.pos_bin);
    float4* __restrict__ out4 = reinterpret_cast<float4*>(out);
    for (int p4 = tid; p4 < (a.out_stride >> 2); p4 += nthr) {
      const int4 b = pb4[p4];
      out4[p4] = float4{value_at(b.x), value_at(b.y), value_at(b.z), value_at(b.w)};
    }
  } else {
    for (int p = tid; p < a.out_stride; p += nthr) out[p] = value_at(tb.pos_bin[p]);
  }
}

// Two things keep the per-bin cost of a step at five instructions for the long groups (seven in round 2):
//  * scaled state: spec, buf and the mel targets are held multiplied by kImelScale = 2^-60.  Every operation of the step
//    is linear except the clamp at zero, and a power-of-two factor commutes with fp32 rounding, so the scaled iteration is
//    the unscaled one bit for bit (as long as nothing leaves the normal range: values below 1.4e-20 in the reference's units
//    would, they sit 23+ orders of magnitude under a spectrogram's scale) - and `max(0, x)` becomes the VALU's free output
//    clamp to [0, 1] on the FMA that produces x (the upper bound is 1.15e18 in the reference's units);
//  * unit form (UF): between two filter centres the falling weight of filter g and the rising weight of filter g+1 sum to
//    one (torchaudio's melscale_fbanks, norm=None; checked to 1e-6 per bin at plan creation), so the gradient
//    d0 w0 + d1 w1 = d1 + (d0 - d1) w0: one FMA per bin less, and momentum folds into the first (`fma(mom, buf, d1)`).
//    The sums A and B keep both weights (a thread's unused register slots carry w0 = w1 = 0 and must stay out of them; their
//    spec values drift inside [0, 1] and touch nothing).  Unlike the power-of-two scaling above this is NOT bit-identical to
//    the two-weight form: the weights sum to one only to 1e-6 and the gradient is rounded differently; emulated on the CPU
//    against the oracle the two forms sit at the same distance (rel-L2 2.1e-7 both).
// Round 6: the exponent is per CLIP, chosen from the clip's largest mel amplitude (or the caller's magnitude_hint) by
// imel_range_kernel below so that the largest target sits near 2^-35 whatever the units are - 2^-60 for the reference's default
// max_value = 30e6, as in rounds 2-5, and the same bits for ANY power of two (the scale commutes with rounding).  A fixed 2^-60
// saturated silently above 1.15e18 and flushed below 1.4e-20 in the reference's units; now the supported range is the one
// include/rfx.h states ("Numeric range").
constexpr float kImelScale = 8.673617379884035e-19f;    // 2^-60: without a per-clip table (ImelArgs::clip_scale == nullptr)
constexpr float kImelUnscale = 1152921504606846976.0f;  // 2^60
// sets a.sc / a.un (scale into the state's units / back) for the frame's clip
__device__ __forceinline__ void imel_set_scale(ImelArgs& a, int clip) {
  a.sc = a.clip_scale ? a.clip_scale[2 * clip] : kImelScale;
  a.un = a.clip_scale ? a.clip_scale[2 * clip + 1] : kImelUnscale;
}
#ifndef RFX_IMEL_CLAMP
#define RFX_IMEL_CLAMP 1
#endif
#ifndef RFX_IMEL_UFORM
#define RFX_IMEL_UFORM 1
#endif

__device__ __forceinline__ float clamp_step(float x) {
#if RFX_IMEL_CLAMP
  return __builtin_amdgcn_fmed3f(x, 0.f, 1.f);  // folds into the producing instruction's clamp modifier
#else
  return fmaxf(0.f, x);
#endif
}

#ifndef RFX_IMEL_PK
#define RFX_IMEL_PK 1
#endif
#if RFX_IMEL_PK
// Round 4: the per-bin state lives in register PAIRS (bins 2i and 2i+1 of the group) and every operation of the step is one
// v_pk_*_f32: the step is bound by the issue slots of the frame's heaviest wave, and a packed instruction does
// the work of two plain ones in one slot.  The arithmetic per bin is the plain form's, operation for operation: the group
// sums were already accumulated as even / odd partial sums, now the two halves of one accumulator.  A padding half (odd
// bin counts) carries w0 = w1 = 0 like every unused slot.
using c2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c2 bc2(float x) { return c2{x, x}; }
// spec = clamp(spec + nl * buf, 0, 1): the output clamp of the packed FMA (the compiler does not fold it into v_pk_fma_f32: two
// v_max per pair); nl = -lr * gradient scale sits in both halves of an SGPR pair
__device__ __forceinline__ c2 pk_step_clamp(c2 spec, unsigned long long nl2, c2 buf) {
#if RFX_IMEL_CLAMP
  asm("v_pk_fma_f32 %0, %1, %2, %0 clamp" : "+v"(spec) : "s"(nl2), "v"(buf));
  return spec;
#else
  const float nl = __builtin_bit_cast(float, (unsigned)nl2);
  const c2 r = __builtin_elementwise_fma(bc2(nl), buf, spec);
  return c2{fmaxf(0.f, r.x), fmaxf(0.f, r.y)};
#endif
}

template <int N, bool UF>
struct GroupState {
  static constexpr int NP = (N + 1) / 2;
  c2 spec[NP], buf[NP], w0[NP], w1[NP];
  int f0, n;  // first bin, bin count
};

template <int N, bool UF>
__device__ __forceinline__ void group_load(GroupState<N, UF>& g, int grp, const ImelArgs& a, int frame, unsigned rbase, float scale) {
  const ImelTables& tb = a.tb;
  g.f0 = grp >= 0 ? tb.grp_start[grp] : 0;
  g.n = grp >= 0 ? tb.grp_start[grp + 1] - g.f0 : 0;
#pragma unroll
  for (int i = 0; i < 2 * g.NP; ++i) {
    const bool ok = i < g.n;
    const int f = g.f0 + (ok ? i : 0);
    const float w0 = ok ? tb.bin_w0[f] : 0.f, w1 = ok ? tb.bin_w1[f] : 0.f;
    const float sp = ok ? scale * (a.spec0 ? a.spec0[(size_t)frame * a.n_stft + f] : rand_unit(rbase, f)) : 0.f;
    if (i & 1) { g.w0[i >> 1].y = w0; g.w1[i >> 1].y = w1; g.spec[i >> 1].y = sp; g.buf[i >> 1].y = 0.f; }
    else       { g.w0[i >> 1].x = w0; g.w1[i >> 1].x = w1; g.spec[i >> 1].x = sp; g.buf[i >> 1].x = 0.f; }
  }
}
template <int N, bool UF>
__device__ __forceinline__ void group_ab(const GroupState<N, UF>& g, float& A, float& B) {
  c2 sa = bc2(0.f), sb = bc2(0.f);
#pragma unroll
  for (int i = 0; i < g.NP; ++i) {
    sa = __builtin_elementwise_fma(g.w0[i], g.spec[i], sa);
    sb = __builtin_elementwise_fma(g.w1[i], g.spec[i], sb);
  }
  A = sa.x + sa.y;
  B = sb.x + sb.y;
}
template <int N, bool UF>
__device__ __forceinline__ void group_step(GroupState<N, UF>& g, float d0, float d1, float mom, unsigned long long nl2) {
  const c2 vm = bc2(mom), v0 = bc2(d0), v1 = bc2(d1), vd = bc2(d0 - d1);
#pragma unroll
  for (int i = 0; i < g.NP; ++i) {
    // torch.optim.SGD: buf.mul_(momentum).add_(grad), in place (see the plain form below)
    c2 bnew;
    if (UF) {
      bnew = __builtin_elementwise_fma(vm, g.buf[i], v1);
      bnew = __builtin_elementwise_fma(vd, g.w0[i], bnew);
    } else {
      bnew = vm * g.buf[i];
      bnew = __builtin_elementwise_fma(v0, g.w0[i], bnew);
      bnew = __builtin_elementwise_fma(v1, g.w1[i], bnew);
    }
    g.buf[i] = bnew;
  }
#pragma unroll
  for (int i = 0; i < g.NP; ++i) g.spec[i] = pk_step_clamp(g.spec[i], nl2, g.buf[i]);
}
template <int N, bool UF>
__device__ __forceinline__ void group_stage(const GroupState<N, UF>& g, const ImelTables& tb, float* stage, float unscale) {
#pragma unroll
  for (int i = 0; i < 2 * g.NP; ++i)
    if (i < g.n) stage[g.f0 + i - tb.f_lo] = unscale * ((i & 1) ? g.spec[i >> 1].y : g.spec[i >> 1].x);
}
#else
template <int N, bool UF>
struct GroupState {
  float spec[N], buf[N], w0[N], w1[N];
  int f0, n;  // first bin, bin count
};

template <int N, bool UF>
__device__ __forceinline__ void group_load(GroupState<N, UF>& g, int grp, const ImelArgs& a, int frame, unsigned rbase, float scale) {
  const ImelTables& tb = a.tb;
  g.f0 = grp >= 0 ? tb.grp_start[grp] : 0;
  g.n = grp >= 0 ? tb.grp_start[grp + 1] - g.f0 : 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const bool ok = i < g.n;
    const int f = g.f0 + (ok ? i : 0);
    g.w0[i] = ok ? tb.bin_w0[f] : 0.f;
    g.w1[i] = ok ? tb.bin_w1[f] : 0.f;
    g.spec[i] = ok ? scale * (a.spec0 ? a.spec0[(size_t)frame * a.n_stft + f] : rand_unit(rbase, f)) : 0.f;
    g.buf[i] = 0.f;
  }
}
template <int N, bool UF>
__device__ __forceinline__ void group_ab(const GroupState<N, UF>& g, float& A, float& B) {
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
  for (int i = 0; i < N; i += 2) {
    a0 = fmaf(g.w0[i], g.spec[i], a0);
    b0 = fmaf(g.w1[i], g.spec[i], b0);
    if (i + 1 < N) {
      a1 = fmaf(g.w0[i + 1], g.spec[i + 1], a1);
      b1 = fmaf(g.w1[i + 1], g.spec[i + 1], b1);
    }
  }
  A = a0 + a1;
  B = b0 + b1;
}
template <int N, bool UF>
__device__ __forceinline__ void group_step(GroupState<N, UF>& g, float d0, float d1, float mom, unsigned long long nl2) {
  const float lrg = -__builtin_bit_cast(float, (unsigned)nl2);
  const float dd = d0 - d1;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // torch.optim.SGD: buf.mul_(momentum).add_(grad); accumulating in place keeps buf in its register
    // (a separate gradient temporary costs a v_mov per bin and step across the loop back-edge).  The first step's
    // buf = grad needs no special case: buf starts at +0 and momentum * 0 is +0
    float bnew;
    if (UF) {
      bnew = fmaf(mom, g.buf[i], d1);
      bnew = fmaf(dd, g.w0[i], bnew);
    } else {
      bnew = mom * g.buf[i];
      bnew = fmaf(d0, g.w0[i], bnew);
      bnew = fmaf(d1, g.w1[i], bnew);
    }
    g.buf[i] = bnew;
    g.spec[i] = clamp_step(fmaf(-lrg, bnew, g.spec[i]));
  }
}
template <int N, bool UF>
__device__ __forceinline__ void group_stage(const GroupState<N, UF>& g, const ImelTables& tb, float* stage, float unscale) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < g.n) stage[g.f0 + i - tb.f_lo] = unscale * g.spec[i];
}

#endif

// `tid` is the thread's ROLE (0..255: which two groups it owns; roles 64c..64c+63 form size class c), not its hardware
// index: the kernels below deal the four classes to the waves of a workgroup in different orders.  `frame` is the frame
// this wave's workgroup slot works on, `live` false for a slot past the last frame (it runs the same barriers on a copy of
// the last frame's data and stores nothing).
template <int NLO, int NHI, bool UF>
__device__ __forceinline__ void imel_group_body(ImelArgs a, char* smem, int tid, int frame, bool live) {
  const ImelTables& tb = a.tb;
  const int M = a.M;
  float* Ab = reinterpret_cast<float*>(smem);  // [2][M + 4], entry m at index m + 1
  float* Bb = Ab + 2 * (M + 4);                // [2][M + 4]
  float* part = Bb + 2 * (M + 4);              // [max_iter][4] per-wave partial sums of diff^2

  const int b = frame / a.T, t = frame - b * a.T;
  const int clip = b / a.C;
  const int steps = a.it_limit ? a.it_limit[clip] : a.max_iter;
  if (a.it_limit && steps >= a.max_iter) return;  // fix-up pass (one frame per workgroup): this clip never stopped early
  const unsigned rbase = rand_frame_key(a.seed, a.frame_base + (unsigned long long)frame);
  imel_set_scale(a, clip);

  const int gH = (M - 1 - tid >= 0) ? M - 1 - tid : -1;           // long groups, counted down from the top
  const int gL = (tid < M - kImelThreads) ? tid : -1;             // short groups, counted up from 0
  // the short groups keep both weights (the lowest bins sit below the first filter's centre and feed one filter only);
  // the long groups run in unit form when the plan found the bank fit for it (UF)
  const float kScale = RFX_IMEL_CLAMP ? a.sc : 1.f, kUnscale = RFX_IMEL_CLAMP ? a.un : 1.f;
  GroupState<NLO, false> lo;
  GroupState<NHI, UF> hi;
  group_load(lo, gL, a, frame, rbase, kScale);
  group_load(hi, gH, a, frame, rbase, kScale);
  auto melat = [&](int m) { return (m >= 0 && m < M) ? kScale * a.mel[((size_t)b * M + m) * a.T + t] : 0.f; };
  const float mL0 = gL >= 0 ? melat(gL) : 0.f, mL1 = gL >= 0 ? melat(gL + 1) : 0.f;
  const float mH0 = gH >= 0 ? melat(gH) : 0.f, mH1 = gH >= 0 ? melat(gH + 1) : 0.f;
  for (int i = tid; i < 4 * (M + 4); i += kImelThreads) Ab[i] = 0.f;  // Ab and Bb are contiguous: zero both incl. pads
  // The momentum buffer is kept in units of the gradient scale g = -2/(C T) of the loss mean (buf = g buf''): the step
  // spec -= lr buf becomes spec = fma(-lr g, buf'', spec) and the four products g * residual per step disappear
  const float lrg = a.lr * (-2.0f / (float)(a.C * a.T));
  // -lr g in both halves of an SGPR pair (wave-uniform: from kernel arguments only)
  const unsigned nlb = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, -lrg));
  const unsigned long long nl2 = ((unsigned long long)nlb << 32) | nlb;
  // an absent group (n_mels < 512) publishes zeros to the dump entry and reads the pads around it: the loop below has no
  // branches, and the four neighbour reads of a step go out together (one LDS round trip, not four)
  const int xL = (gL >= 0 ? gL : M + 1) + 1, xH = (gH >= 0 ? gH : M + 1) + 1;
  // unit form: the last group's second filter does not exist (its bins carry w1 == 0 in the bank): its residual is forced to 0
  const bool noH1 = UF && gH == M - 1;
  const int wave = tid >> 6;
  __syncthreads();

  // one SGD step; Ap / Bp = this step's half of the double buffers (the parity is a compile-time matter of the caller)
  auto sgd_step = [&](int it, float* __restrict__ Ap, float* __restrict__ Bp) {
    float AL, BL, AH, BH;
    group_ab(lo, AL, BL);
    group_ab(hi, AH, BH);
    Ap[xL] = AL; Bp[xL] = BL;
    Ap[xH] = AH; Bp[xH] = BH;
    __syncthreads();
    const float bLm = Bp[xL - 1], aLp = Ap[xL + 1], bHm = Bp[xH - 1], aHp = Ap[xH + 1];
    // residuals of the two filters each group feeds: d0 = diff[g], d1 = diff[g+1].  An absent group needs no special case:
    // its targets and sums are zero and the entries next to the dump are never written
    const float dL0 = mL0 - AL - bLm;
    const float dL1 = mL1 - aLp - BL;
    const float dH0 = mH0 - AH - bHm;
    const float dH1 = noH1 ? 0.f : mH1 - aHp - BH;
    // every filter's residual is owned exactly once; the loss history is kept in the reference's units
#ifndef RFX_ABL_IMEL_NO_LOSS  // (ablation: what the per-step loss history costs)
    const float uL = kUnscale * dL0, uH = kUnscale * dH0;
    const float sq = wave_sum(fmaf(uL, uL, uH * uH));
    if ((tid & 63) == 0) part[4 * it + wave] = sq;
#endif
    // (without the unit form the last filter needs nothing either: it has no successor and its d1 multiplies w1 == 0)
    group_step(lo, dL0, dL1, a.momentum, nl2);
    group_step(hi, dH0, dH1, a.momentum, nl2);
  };
  float* const A0 = Ab, * const A1 = Ab + (M + 4), * const B0 = Bb, * const B1 = Bb + (M + 4);
  int it = 0;
  for (; it + 1 < steps; it += 2) {
    sgd_step(it, A0, B0);
    sgd_step(it + 1, A1, B1);
  }
  if (it < steps) sgd_step(it, A0, B0);
  __syncthreads();

  float* stage = reinterpret_cast<float*>(smem + imel_group_lds_bytes(M, a.max_iter));  // behind the loss partials (imel_frame_lds_bytes)
  group_stage(lo, tb, stage, kUnscale);
  group_stage(hi, tb, stage, kUnscale);
  __syncthreads();  // (every frame of the workgroup: a slot without a frame computes a copy of the last one and stores nothing)
  if (!live) return;
  imel_emit_frame(a, stage, frame, rbase, tid, kImelThreads);
  if (a.loss_hist && !a.it_limit)
    for (int i = tid; i < a.max_iter; i += kImelThreads)
      a.loss_hist[(size_t)frame * a.max_iter + i] = i < steps ? (part[4 * i] + part[4 * i + 1]) + (part[4 * i + 2] + part[4 * i + 3]) : 0.f;
}

// uniform register budget for every wave
template <int NLO, int NHI>
__global__ void __launch_bounds__(kImelThreads) imel_group_kernel(ImelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  imel_group_body<NLO, NHI, false>(a, smem, threadIdx.x, blockIdx.x, true);
}
// Group sizes fall with the role index (mel spacing is logarithmic): roles 64c..64c+63 form size class c, and each wave
// runs the body compiled for its class's maximum, so the long-group class no longer sets everybody's instruction count.
// All bodies execute the same sequence of barriers.
//
// FPW frames per workgroup (4 FPW waves).  The waves of a workgroup land on the CU's four SIMDs round-robin
// (profiles/r01_wave_placement_ubench.txt: wave i and wave i+4 share a SIMD), and a class-0 wave issues 25/15 of a class-3
// wave's instructions per SGD step.  With one frame per workgroup, which SIMD gets the heavy wave is left to the order
// workgroups happen to arrive in; with FPW = 2 or 4 the classes are dealt so that the waves sharing a SIMD carry different
// classes (FPW = 2: c and 3-c, FPW = 4: a Latin square, every SIMD gets one wave of each class).  Measured (rounds 2, 3): one
// frame per workgroup wins; dealing its classes by HW_REG_HW_ID (class = (SIMD + wave slot) mod 4, a Latin square over the
// resident workgroups) or by blockIdx changes nothing (4.92 - 5.02 ms either way: the dispatcher already spreads the heavy
// waves), and s_setprio by class costs 7 %.
#ifndef RFX_IMEL_WAVES_PER_EU
#define RFX_IMEL_WAVES_PER_EU 4  // 128 VGPRs: 16 waves per CU (7.6 ms vs 8.6 ms at 3, measured)
#endif
#ifndef RFX_IMEL_WAVES_PER_EU_UF
#define RFX_IMEL_WAVES_PER_EU_UF 4
#endif
#ifndef RFX_IMEL_FPW
#define RFX_IMEL_FPW 1
#endif

template <int FPW, int WPE, bool UF, int L0, int H0, int L1, int H1, int L2, int H2, int L3, int H3>
__global__ void __launch_bounds__(kImelThreads * FPW) __attribute__((amdgpu_waves_per_eu(WPE)))
imel_group_kernel_perwave(ImelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = wave >> 2, w4 = wave & 3;
  const int cls = FPW == 1 ? w4 : FPW == 2 ? (slot ? 3 - w4 : w4) : ((w4 + slot) & 3);
  const int tid = cls * 64 + (threadIdx.x & 63);
  const int nframes = a.B * a.T;
  int frame = blockIdx.x * FPW + slot;
  const bool live = frame < nframes;
  if (!live) frame = nframes - 1;
  char* my = smem + (size_t)slot * imel_frame_lds_bytes(a.M, a.max_iter, a.tb.f_hi - a.tb.f_lo);
  switch (cls) {
    case 0: imel_group_body<L0, H0, UF>(a, my, tid, frame, live); break;
    case 1: imel_group_body<L1, H1, UF>(a, my, tid, frame, live); break;
    case 2: imel_group_body<L2, H2, UF>(a, my, tid, frame, live); break;
    default: imel_group_body<L3, H3, UF>(a, my, tid, frame, live); break;
  }
}

// (Round 3 also measured a two-frames-per-workgroup kernel with complementary size classes per wave - every wave within 4 % of
// the mean instruction count, twelve balanced waves per CU instead of sixteen unbalanced ones: 5.39 ms against 4.75 ms for the
// kernel above on the same box, profiles/r03b_imel_pair_kernel_experiment.txt.  The code was removed in round 4.)

// ---------------------------------------------------------------------------------------------------
// Wave kernel (round 4): ONE wave per frame, no barrier and no LDS exchange inside the SGD loop.
//
// The group kernels above cross a workgroup barrier per step with four waves of unequal length; 39 % of their wave-cycles
// are spent parked (profiles/r04_imel_pmc.json).  Here a frame is one wave: lane l owns eight groups, one per chunk of 64
// consecutive groups - group 64 c + l in the even chunks, 64 c + 63 - l in the odd ones (rfx_kernels.h::imel_wave_group) - so
//  * every lane carries 60 - 67 of the 4000 active bins although a group grows from 1 to 23 bins over the bank,
//  * the neighbours g - 1 and g + 1 of a lane's group sit in the adjacent lane (one DPP wave shift each) or, where two chunks
//    meet, in the lane itself (the `old` operand of the same DPP instruction: the shift leaves the end lane untouched),
//  * all state of the frame stays in the wave's VGPRs at two waves per SIMD.
// Two observations make the state small and the step cheap:
//  1. on a uniform bin grid a triangular filter's weight is LINEAR in the bin index inside a group, w0 = a0 + s0 i,
//     w1 = a1 + s1 i (least-squares line in double, checked against the table to 1e-6 per bin at plan creation,
//     ImelTables::lin), so with S = sum x_i and Q = sum i x_i
//        A = a0 S + s0 Q,  B = a1 S + s1 Q  (unit form, chunks 4 - 7 of a bank without area normalisation: B = S - A)
//        gradient of bin i = (d0 a0 + d1 a1) + (d0 s0 + d1 s1) i  =: cc + st i        - no weight registers;
//  2. the gradient is a line in i, the momentum buffer starts at zero and torch.optim.SGD updates it linearly
//     (buf <- momentum buf + grad, whatever the clamp does to x afterwards), so the buffer of a group's bin i IS the line
//     C + G i with C <- momentum C + cc, G <- momentum G + st: two scalars per group instead of a register per bin.
// A PAIR of bins (2p, 2p + 1) then costs four packed instructions per step - S += x; Q += p x (Q = 2 (Qx + Qy) + Sy);
// v = fma(p, (2 h, 2 h), (-lr g C, -lr g C + h)) with h = -lr g G; x = clamp(x + v) - minus the p = 0 and p = 1 terms that need
// no arithmetic: 132 packed instructions per frame and step where the group kernels issue 209.  The state is the scaled one
// of the group kernels (2^-60, output clamp), the residuals are formed in the same order.
// Padding slots (a lane's group is shorter than its chunk's budget) hold x = 0 and their step is multiplied by a per-lane
// 0 / 1 mask (x = clamp(fma(v, mask, x))) - only the pairs behind kImelWaveFullPairs can be padding and carry one.
// The per-step loss (sum of the squared residuals over the frame's filters, read by imel_scan_kernel) is summed by the LDS
// unit (ds_add_f32 of all lanes into one word: the unit is otherwise idle here), not by six DPP steps on the VALU.
// Numerics: not bit-identical to the group kernels (weights and buffer from lines: within one ulp of the GROUP's largest weight -
// the plan admits this kernel only if every bin's weight is within 4e-7 of that maximum of its fitted line, rfx_api.hip - sums in
// another order); emulated in numpy against the oracle (tests/test_imel_wave_form.py) rel-L2 3.1e-7 after 120 steps (table weights:
// 2.1e-7), on the device 8.9e-8 against the group kernels at T = 512, gate 1e-3.
// Measured per VALU instruction and SIMD at two waves per SIMD (tools/ubench/valu_rate.hip): v_pk_fma_f32 2.4 ns, v_fma_f32 1.5,
// v_mov_b32_dpp wave_shr 2.1: the kernel runs at the sum of its instructions' costs, i.e. the count is what is left to cut.
// ---------------------------------------------------------------------------------------------------
#if RFX_IMEL_PK
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;
// lane i receives `src` of lane i - 1 (SHR) or i + 1 (SHL); the lane at the end keeps `old`
template <int CTRL>
__device__ __forceinline__ float wave_shift(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
// x = clamp(x + v, 0, 1) / x = clamp(x + v m, 0, 1): the packed instruction's output clamp (see pk_step_clamp)
__device__ __forceinline__ c2 pk_add_clamp(c2 x, c2 v) {
  asm("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(x) : "v"(v));
  return x;
}
__device__ __forceinline__ c2 pk_fma_clamp(c2 x, c2 v, c2 m) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 clamp" : "+v"(x) : "v"(v), "v"(m));
  return x;
}

// x = m x + c written over x (hipcc picks v_fmac, whose result lands in c's register, and copies it back every trip of the loop)
__device__ __forceinline__ void fma_in_place(float& x, unsigned m_sgpr, float c) { asm("v_fma_f32 %0, %1, %0, %2" : "+v"(x) : "s"(m_sgpr), "v"(c)); }

template <int NP, int NF, bool UF>  // NP pairs of slots, the first NF of them full in every lane
struct WvChunk {
  static constexpr int NT = NP - NF;
  c2 spec[NP];
  c2 mask[NT > 0 ? NT : 1];
  float a0, s0, a1, s1;  // a1, s1 unused in unit form
  float m0;              // scaled mel target of filter g
  float C, G;            // the momentum buffer of the group's bin i is C + G i, in units of the STEP (-lr x gradient scale folded in)
};

template <int NP, int NF, bool UF>
__device__ __forceinline__ void wv_load(WvChunk<NP, NF, UF>& k, int g, const ImelArgs& a, int frame, int b, int t, unsigned rbase) {
  const ImelTables& tb = a.tb;
  const int f0 = tb.grp_start[g], n = tb.grp_start[g + 1] - f0;
  k.a0 = tb.lin[g];
  k.s0 = tb.lin[a.M + g];
  k.a1 = tb.lin[2 * a.M + g];
  k.s1 = tb.lin[3 * a.M + g];
  k.m0 = a.sc * a.mel[((size_t)b * a.M + g) * a.T + t];
  k.C = 0.f;
  k.G = 0.f;
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) {
    const bool ok = i < n;
    const int f = f0 + (ok ? i : 0);
    const float sp = ok ? a.sc * (a.spec0 ? a.spec0[(size_t)frame * a.n_stft + f] : rand_unit(rbase, f)) : 0.f;
    if (i & 1) k.spec[i >> 1].y = sp; else k.spec[i >> 1].x = sp;
    if (i >= 2 * NF) {
      if (i & 1) k.mask[(i >> 1) - NF].y = ok ? 1.f : 0.f; else k.mask[(i >> 1) - NF].x = ok ? 1.f : 0.f;
    }
  }
}
// The step is written phase by phase ACROSS chunks - the compiler keeps the source order of independent instructions: the
// packed sums of two chunks advance together (four accumulator chains), the scalar tails of four chunks at a time, and the
// update forms all of a chunk's step pairs before it applies them.
template <int NPA, int NFA, int NPB, int NFB, bool UFA, bool UFB>
__device__ __forceinline__ void wv_sums2(const WvChunk<NPA, NFA, UFA>& ka, const WvChunk<NPB, NFB, UFB>& kb, c2& SA, c2& QA, c2& SB, c2& QB) {
  static_assert(NPA >= 2 && NPB >= 2, "every chunk holds at least two pairs");
  SA = ka.spec[0] + ka.spec[1];
  SB = kb.spec[0] + kb.spec[1];
  QA = ka.spec[1];
  QB = kb.spec[1];
#pragma unroll
  for (int p = 2; p < (NPA > NPB ? NPA : NPB); ++p) {
    if (p < NPA) SA = SA + ka.spec[p];
    if (p < NPB) SB = SB + kb.spec[p];
    if (p < NPA) QA = __builtin_elementwise_fma(bc2((float)p), ka.spec[p], QA);
    if (p < NPB) QB = __builtin_elementwise_fma(bc2((float)p), kb.spec[p], QB);
  }
}
// step of the pair p: v_p = (vx, vy) + p (w2, w2) with vx = C, vy = C + G, w2 = 2 G
template <int NP, int NF, bool UF>
__device__ __forceinline__ void wv_update(WvChunk<NP, NF, UF>& k, float vx, float vy, float w2) {
  const c2 base = c2{vx, vy}, s2 = bc2(w2);
  c2 v[NP];
  v[0] = base;
#pragma unroll
  for (int p = 1; p < NP; ++p) v[p] = __builtin_elementwise_fma(bc2((float)p), s2, base);
#pragma unroll
  for (int p = 0; p < NP; ++p) k.spec[p] = p < NF ? pk_add_clamp(k.spec[p], v[p]) : pk_fma_clamp(k.spec[p], v[p], k.mask[p < NF ? 0 : p - NF]);
}
// the chunk's bins, unscaled, into the frame's LDS stage (bin order: entry f - f_lo)
template <int NP, int NF, bool UF>
__device__ __forceinline__ void wv_stage(const WvChunk<NP, NF, UF>& k, int g, const ImelTables& tb, float* stage, float unscale) {
  const int f0 = tb.grp_start[g], n = tb.grp_start[g + 1] - f0;
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i)
    if (i < n) stage[f0 + i - tb.f_lo] = unscale * ((i & 1) ? k.spec[i >> 1].y : k.spec[i >> 1].x);
}

// ---------------------------------------------------------------------------------------------------
// Line form inside the group-kernel layout (round 5).  The wave kernel above serves ONE bank shape (512 groups whose sizes fit its
// chunk budgets: the default 0 - 10 kHz bank); every bank with longer groups - 512 filters up to 16 / 20 / 22.05 kHz (the
// reference's own round-trip test runs 20 Hz .. 20 kHz, test/spectrogram_converter_test.py:46-53), 384 filters - fell through to
// the general LDS kernel: 169 ms per 64 tiles against 4.5.  The group kernels cannot take them either: with four registers per
// bin (spec, buffer, two weights) a 62-bin group does not fit a thread.  Here a thread keeps the group kernels' roles and LDS
// exchange (short group t, long group M-1-t; A and B published, one barrier per step, residuals formed from the neighbours'
// sums) but holds its LONG group in the wave kernel's line form - weights a0 + s0 i, momentum buffer C + G i, ONE register per
// bin plus a 0 / 1 mask per slot (group sizes vary inside a wave's class, and in unit form B = S - A must not see a padding slot) -
// while the short group stays in table form (group 0 of a bank may hold its first filter's rising edge and is not a line).
// The plan admits the kernel when the long groups M-256 .. M-1 are lines (rfx_api.hip, code 5) and the budgets kImelLoCapLine /
// kImelHiCapLine hold every group.  Two waves per SIMD (the heaviest class holds 31 spec pairs + 31 masks).
template <int NP, bool UF>
struct LineGroup {
  c2 spec[NP], mask[NP];
  float a0, s0, a1, s1;  // a1, s1 unused in unit form
  float C, G;            // the momentum buffer of the group's bin i is C + G i, in units of the STEP
};
template <int NP, bool UF>
__device__ __forceinline__ void line_load(LineGroup<NP, UF>& k, int g, const ImelArgs& a, int frame, unsigned rbase) {
  const ImelTables& tb = a.tb;
  const int f0 = g >= 0 ? tb.grp_start[g] : 0, n = g >= 0 ? tb.grp_start[g + 1] - f0 : 0;
  k.a0 = g >= 0 ? tb.lin[g] : 0.f;
  k.s0 = g >= 0 ? tb.lin[a.M + g] : 0.f;
  k.a1 = g >= 0 ? tb.lin[2 * a.M + g] : 0.f;
  k.s1 = g >= 0 ? tb.lin[3 * a.M + g] : 0.f;
  k.C = 0.f;
  k.G = 0.f;
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i) {
    const bool ok = i < n;
    const int f = f0 + (ok ? i : 0);
    const float sp = ok ? a.sc * (a.spec0 ? a.spec0[(size_t)frame * a.n_stft + f] : rand_unit(rbase, f)) : 0.f;
    if (i & 1) { k.spec[i >> 1].y = sp; k.mask[i >> 1].y = ok ? 1.f : 0.f; }
    else       { k.spec[i >> 1].x = sp; k.mask[i >> 1].x = ok ? 1.f : 0.f; }
  }
}
// A = sum w0 x = a0 S + s0 Q with S = sum x_i, Q = sum i x_i (four accumulator chains); B likewise, or S - A in unit form
template <int NP, bool UF>
__device__ __forceinline__ void line_ab(const LineGroup<NP, UF>& k, float& A, float& B) {
  static_assert(NP >= 2, "a line group holds at least two pairs");
  c2 Sa = k.spec[0], Sb = k.spec[1], Qa = bc2(0.f), Qb = k.spec[1];
#pragma unroll
  for (int p = 2; p < NP; ++p) {
    if (p & 1) { Sb = Sb + k.spec[p]; Qb = __builtin_elementwise_fma(bc2((float)p), k.spec[p], Qb); }
    else       { Sa = Sa + k.spec[p]; Qa = __builtin_elementwise_fma(bc2((float)p), k.spec[p], Qa); }
  }
  const c2 S = Sa + Sb, Q = Qa + Qb;
  const float s = S.x + S.y, h = Q.x + Q.y;
  const float q = fmaf(2.f, h, S.y);  // sum i x_i over the slots (2p, 2p + 1) = 2 sum p (x_2p + x_2p+1) + sum x_2p+1
  A = fmaf(k.s0, q, k.a0 * s);
  B = UF ? s - A : fmaf(k.s1, q, k.a1 * s);
}
// n0, n1: the residuals of the group's two filters times the step factor -lr g.  Gradient line cc + st i, buffer line (C, G) updated
// like torch.optim.SGD's buf.mul_(momentum).add_(grad), then x = clamp(x + mask (C + G i)) pair by pair
template <int NP, bool UF>
__device__ __forceinline__ void line_step(LineGroup<NP, UF>& k, float n0, float n1, float mom) {
  float cc, st;
  if (UF) {
    const float dd = n0 - n1;
    cc = fmaf(dd, k.a0, n1);
    st = dd * k.s0;
  } else {
    cc = fmaf(n1, k.a1, n0 * k.a0);
    st = fmaf(n1, k.s1, n0 * k.s0);
  }
  k.C = fmaf(mom, k.C, cc);
  k.G = fmaf(mom, k.G, st);
  const c2 base = c2{k.C, k.C + k.G}, s2 = bc2(k.G + k.G);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const c2 v = p == 0 ? base : __builtin_elementwise_fma(bc2((float)p), s2, base);
    k.spec[p] = pk_fma_clamp(k.spec[p], v, k.mask[p]);
  }
}
template <int NP, bool UF>
__device__ __forceinline__ void line_stage(const LineGroup<NP, UF>& k, int g, const ImelTables& tb, float* stage, float unscale) {
  if (g < 0) return;
  const int f0 = tb.grp_start[g], n = tb.grp_start[g + 1] - f0;
#pragma unroll
  for (int i = 0; i < 2 * NP; ++i)
    if (i < n) stage[f0 + i - tb.f_lo] = unscale * ((i & 1) ? k.spec[i >> 1].y : k.spec[i >> 1].x);
}

// the group body (imel_group_body) with the long group in line form; same roles, same LDS layout, same barriers
template <int NLO, int NHI, bool UF>
__device__ __forceinline__ void imel_line_body(ImelArgs a, char* smem, int tid, int frame) {
  const ImelTables& tb = a.tb;
  const int M = a.M;
  float* Ab = reinterpret_cast<float*>(smem);  // [2][M + 4], entry m at index m + 1
  float* Bb = Ab + 2 * (M + 4);                // [2][M + 4]
  float* part = Bb + 2 * (M + 4);              // [max_iter][4] per-wave partial sums of diff^2
  const int b = frame / a.T, t = frame - b * a.T;
  const int clip = b / a.C;
  const int steps = a.it_limit ? a.it_limit[clip] : a.max_iter;
  if (a.it_limit && steps >= a.max_iter) return;  // fix-up pass: this clip never stopped early
  const unsigned rbase = rand_frame_key(a.seed, a.frame_base + (unsigned long long)frame);
  imel_set_scale(a, clip);
  int gH = (M - 1 - tid >= 0) ? M - 1 - tid : -1;
  int gL = (tid < M - kImelThreads) ? tid : -1;
  if (gH >= 0 && gH < tb.line_from) {  // a long group that is not a line: into the (free: the plan checked) table-form slot
    gL = gH;
    gH = -1;
  }
  GroupState<NLO, false> lo;
  LineGroup<(NHI + 1) / 2, UF> hi;
  group_load(lo, gL, a, frame, rbase, a.sc);
  line_load(hi, gH, a, frame, rbase);
  auto melat = [&](int m) { return (m >= 0 && m < M) ? a.sc * a.mel[((size_t)b * M + m) * a.T + t] : 0.f; };
  const float mL0 = gL >= 0 ? melat(gL) : 0.f, mL1 = gL >= 0 ? melat(gL + 1) : 0.f;
  const float mH0 = gH >= 0 ? melat(gH) : 0.f, mH1 = gH >= 0 ? melat(gH + 1) : 0.f;
  for (int i = tid; i < 4 * (M + 4); i += kImelThreads) Ab[i] = 0.f;
  const float lrg = a.lr * (-2.0f / (float)(a.C * a.T));
  const float nl = -lrg;  // the step in units of the gradient scale (see imel_group_body)
  const unsigned nlb = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, nl));
  const unsigned long long nl2 = ((unsigned long long)nlb << 32) | nlb;
  const int xL = (gL >= 0 ? gL : M + 1) + 1, xH = (gH >= 0 ? gH : M + 1) + 1;
  const bool noH1 = UF && gH == M - 1;
  const int wave = tid >> 6;
  __syncthreads();

  auto sgd_step = [&](int it, float* __restrict__ Ap, float* __restrict__ Bp) {
    float AL, BL, AH, BH;
    group_ab(lo, AL, BL);
    line_ab(hi, AH, BH);
    Ap[xL] = AL; Bp[xL] = BL;
    Ap[xH] = AH; Bp[xH] = BH;
    __syncthreads();
    const float bLm = Bp[xL - 1], aLp = Ap[xL + 1], bHm = Bp[xH - 1], aHp = Ap[xH + 1];
    const float dL0 = mL0 - AL - bLm;
    const float dL1 = mL1 - aLp - BL;
    const float dH0 = mH0 - AH - bHm;
    const float dH1 = noH1 ? 0.f : mH1 - aHp - BH;
    const float uL = a.un * dL0, uH = a.un * dH0;
    const float sq = wave_sum(fmaf(uL, uL, uH * uH));
    if ((tid & 63) == 0) part[4 * it + wave] = sq;
    group_step(lo, dL0, dL1, a.momentum, nl2);
    line_step(hi, nl * dH0, nl * dH1, a.momentum);
  };
  float* const A0 = Ab, * const A1 = Ab + (M + 4), * const B0 = Bb, * const B1 = Bb + (M + 4);
  int it = 0;
  for (; it + 1 < steps; it += 2) {
    sgd_step(it, A0, B0);
    sgd_step(it + 1, A1, B1);
  }
  if (it < steps) sgd_step(it, A0, B0);
  __syncthreads();

  float* stage = reinterpret_cast<float*>(smem + imel_group_lds_bytes(M, a.max_iter));
  group_stage(lo, tb, stage, a.un);
  line_stage(hi, gH, tb, stage, a.un);
  __syncthreads();
  imel_emit_frame(a, stage, frame, rbase, tid, kImelThreads);
  if (a.loss_hist && !a.it_limit)
    for (int i = tid; i < a.max_iter; i += kImelThreads)
      a.loss_hist[(size_t)frame * a.max_iter + i] = i < steps ? (part[4 * i] + part[4 * i + 1]) + (part[4 * i + 2] + part[4 * i + 3]) : 0.f;
}

template <bool UF, int L0, int H0, int L1, int H1, int L2, int H2, int L3, int H3>
__global__ void __launch_bounds__(kImelThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) imel_line_kernel_perwave(ImelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cls = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = threadIdx.x;
  const int frame = blockIdx.x;
  switch (cls) {
    case 0: imel_line_body<L0, H0, UF>(a, smem, tid, frame); break;
    case 1: imel_line_body<L1, H1, UF>(a, smem, tid, frame); break;
    case 2: imel_line_body<L2, H2, UF>(a, smem, tid, frame); break;
    default: imel_line_body<L3, H3, UF>(a, smem, tid, frame); break;
  }
}

#define RFX_WV_CHUNKS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define RFX_WV_LO(X) X(0) X(1) X(2) X(3)
#define RFX_WV_HI(X) X(4) X(5) X(6) X(7)

template <bool UFH>  // unit form in the upper four chunks (triangles without area normalisation); false: both weights everywhere
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) imel_wave_kernel(ImelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = reinterpret_cast<float*>(smem);  // [max_iter] sum of diff^2 over the frame's filters, in the reference's units
  const ImelTables& tb = a.tb;
  const int lane = threadIdx.x, frame = blockIdx.x;
  const int b = frame / a.T, t = frame - b * a.T;
  const int clip = b / a.C;
  const int steps = a.it_limit ? a.it_limit[clip] : a.max_iter;
  if (a.it_limit && steps >= a.max_iter) return;  // fix-up pass: this clip never stopped early
  const unsigned rbase = rand_frame_key(a.seed, a.frame_base + (unsigned long long)frame);
  imel_set_scale(a, clip);
  for (int i = lane; i < a.max_iter; i += 64) part[i] = 0.f;
  const unsigned part_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of part[0]

#define RFX_WV_DECL(c, UF) WvChunk<kImelWavePairs[c], kImelWaveFullPairs[c], UF> k##c;
  RFX_WV_DECL(0, false) RFX_WV_DECL(1, false) RFX_WV_DECL(2, false) RFX_WV_DECL(3, false)
  RFX_WV_DECL(4, UFH) RFX_WV_DECL(5, UFH) RFX_WV_DECL(6, UFH) RFX_WV_DECL(7, UFH)
#undef RFX_WV_DECL
#define RFX_WV_LOAD(c) wv_load(k##c, imel_wave_group(c, lane), a, frame, b, t, rbase);
  RFX_WV_CHUNKS(RFX_WV_LOAD)
#undef RFX_WV_LOAD

  const float nl = -(a.lr * (-2.0f / (float)(a.C * a.T)));  // the step in units of the gradient scale -2 / (C T), see imel_group_body
  const unsigned mom_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.momentum));
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the zeroed loss words (one wave per workgroup: no barrier needed)

  for (int it = 0; it < steps; ++it) {
    float A0, A1, A2, A3, A4, A5, A6, A7, B0, B1, B2, B3, B4, B5, B6, B7;
    {  // group sums of chunks 0 - 3 (both weights), then 4 - 7 (unit form: B = S - A)
      c2 S0, Q0, S1, Q1, S2, Q2, S3, Q3;
      wv_sums2(k0, k1, S0, Q0, S1, Q1);
      wv_sums2(k2, k3, S2, Q2, S3, Q3);
#define RFX_WV_T1(c) const float s##c = S##c.x + S##c.y, h##c = Q##c.x + Q##c.y;
#define RFX_WV_T2(c) const float q##c = fmaf(2.f, h##c, S##c.y), e##c = k##c.a0 * s##c, g##c = k##c.a1 * s##c;
#define RFX_WV_T3(c) A##c = fmaf(k##c.s0, q##c, e##c); B##c = fmaf(k##c.s1, q##c, g##c);
      RFX_WV_LO(RFX_WV_T1) RFX_WV_LO(RFX_WV_T2) RFX_WV_LO(RFX_WV_T3)
#undef RFX_WV_T2
#undef RFX_WV_T3
    }
    {
      c2 S4, Q4, S5, Q5, S6, Q6, S7, Q7;
      wv_sums2(k4, k5, S4, Q4, S5, Q5);
      wv_sums2(k6, k7, S6, Q6, S7, Q7);
#define RFX_WV_T2(c) const float q##c = fmaf(2.f, h##c, S##c.y), e##c = k##c.a0 * s##c;
#define RFX_WV_T3(c) A##c = fmaf(k##c.s0, q##c, e##c);
#define RFX_WV_T4(c) B##c = UFH ? s##c - A##c : fmaf(k##c.s1, q##c, k##c.a1 * s##c);
      RFX_WV_HI(RFX_WV_T1) RFX_WV_HI(RFX_WV_T2) RFX_WV_HI(RFX_WV_T3) RFX_WV_HI(RFX_WV_T4)
#undef RFX_WV_T1
#undef RFX_WV_T2
#undef RFX_WV_T3
#undef RFX_WV_T4
    }
    // B of group g - 1: the previous lane of an even chunk (wave_shr), the next lane of an odd one (wave_shl); the end lane's
    // predecessor is the previous chunk's group in the lane itself.  Residual of filter g: d0 = (mel_g - A_g) - B_{g-1}
#define RFX_WV_R1(c) const float r##c = k##c.m0 - A##c;
    RFX_WV_CHUNKS(RFX_WV_R1)
#undef RFX_WV_R1
    const float p0 = wave_shift<kDppWaveShr1>(0.f, B0), p1 = wave_shift<kDppWaveShl1>(B0, B1), p2 = wave_shift<kDppWaveShr1>(B1, B2),
                p3 = wave_shift<kDppWaveShl1>(B2, B3), p4 = wave_shift<kDppWaveShr1>(B3, B4), p5 = wave_shift<kDppWaveShl1>(B4, B5),
                p6 = wave_shift<kDppWaveShr1>(B5, B6), p7 = wave_shift<kDppWaveShl1>(B6, B7);
#define RFX_WV_R2(c) const float d0##c = r##c - p##c;
    RFX_WV_CHUNKS(RFX_WV_R2)
#undef RFX_WV_R2
#ifndef RFX_ABL_IMEL_NO_LOSS
    {  // every filter's residual is owned exactly once; the loss history is kept in the reference's units
#define RFX_WV_L1(c) const float u##c = a.un * d0##c;
      RFX_WV_CHUNKS(RFX_WV_L1)
#undef RFX_WV_L1
      const float sqa = fmaf(u6, u6, fmaf(u4, u4, fmaf(u2, u2, u0 * u0))), sqb = fmaf(u7, u7, fmaf(u5, u5, fmaf(u3, u3, u1 * u1)));
      // one ds_add_f32 of all 64 lanes into the step's word: the LDS unit adds them (written as asm: the compiler's atomic
      // optimizer would replace a uniform-address atomic by a 64-trip v_readlane loop on the VALU)
      asm volatile("ds_add_f32 %0, %1" ::"v"(part_lds + 4u * (unsigned)it), "v"(sqa + sqb) : "memory");
    }
#endif
    // From here on the residuals carry the step factor -lr g (n = -lr g d): everything below is linear in them, so the buffer
    // line (C, G) is kept in step units and needs no further scaling
#define RFX_WV_N0(c) const float n0##c = nl * d0##c;
    RFX_WV_CHUNKS(RFX_WV_N0)
#undef RFX_WV_N0
    // residual of filter g + 1 = that of the next group: the next lane of an even chunk, the previous lane of an odd one, the
    // following chunk's in the end lane; filter 512 does not exist (chunk 7, lane 0: zero)
    const float n10 = wave_shift<kDppWaveShl1>(n01, n00), n11 = wave_shift<kDppWaveShr1>(n02, n01), n12 = wave_shift<kDppWaveShl1>(n03, n02),
                n13 = wave_shift<kDppWaveShr1>(n04, n03), n14 = wave_shift<kDppWaveShl1>(n05, n04), n15 = wave_shift<kDppWaveShr1>(n06, n05),
                n16 = wave_shift<kDppWaveShl1>(n07, n06), n17 = wave_shift<kDppWaveShr1>(0.f, n07);
    // gradient line of every chunk: bin i of the group gets cc + st i (both weights: chunks 0 - 3; unit form: 4 - 7)
    const float dd4 = n04 - n14, dd5 = n05 - n15, dd6 = n06 - n16, dd7 = n07 - n17;
    const float x0 = n00 * k0.a0, x1 = n01 * k1.a0, x2 = n02 * k2.a0, x3 = n03 * k3.a0;
    const float y0 = n00 * k0.s0, y1 = n01 * k1.s0, y2 = n02 * k2.s0, y3 = n03 * k3.s0;
    const float cc0 = fmaf(n10, k0.a1, x0), cc1 = fmaf(n11, k1.a1, x1), cc2 = fmaf(n12, k2.a1, x2), cc3 = fmaf(n13, k3.a1, x3);
    const float st0 = fmaf(n10, k0.s1, y0), st1 = fmaf(n11, k1.s1, y1), st2 = fmaf(n12, k2.s1, y2), st3 = fmaf(n13, k3.s1, y3);
#define RFX_WV_G0(c) const float cc##c = UFH ? fmaf(dd##c, k##c.a0, n1##c) : fmaf(n1##c, k##c.a1, n0##c * k##c.a0), \
                                 st##c = UFH ? dd##c * k##c.s0 : fmaf(n1##c, k##c.s1, n0##c * k##c.s0);
    RFX_WV_HI(RFX_WV_G0)
#undef RFX_WV_G0
    // torch.optim.SGD: buf.mul_(momentum).add_(grad) for every bin of the group at once - the buffer line (C, G), in place -
    // then the step of the pair p: (C, C + G) + p (2 G, 2 G)
#define RFX_WV_G1(c) fma_in_place(k##c.C, mom_s, cc##c); fma_in_place(k##c.G, mom_s, st##c);
#define RFX_WV_G3(c) const float vx##c = k##c.C, vy##c = k##c.C + k##c.G, w2##c = k##c.G + k##c.G;
    RFX_WV_CHUNKS(RFX_WV_G1) RFX_WV_CHUNKS(RFX_WV_G3)
#undef RFX_WV_G1
#undef RFX_WV_G3
#define RFX_WV_UPDATE(c) wv_update(k##c, vx##c, vy##c, w2##c);
    RFX_WV_CHUNKS(RFX_WV_UPDATE)
#undef RFX_WV_UPDATE
  }

  // the frame leaves through the LDS stage (imel_emit_frame): active bins parked in bin order, then one walk over the positions
  float* stage = part + a.max_iter;  // [f_hi - f_lo]
#define RFX_WV_STAGE(c) wv_stage(k##c, imel_wave_group(c, lane), tb, stage, a.un);
  RFX_WV_CHUNKS(RFX_WV_STAGE)
#undef RFX_WV_STAGE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // one wave: its LDS operations execute in order, the compiler must keep them so
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  imel_emit_frame(a, stage, frame, rbase, lane, 64);
  if (a.loss_hist && !a.it_limit) {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS atomics
    for (int i = lane; i < a.max_iter; i += 64) a.loss_hist[(size_t)frame * a.max_iter + i] = i < steps ? part[i] : 0.f;
  }
}
#undef RFX_WV_CHUNKS
#undef RFX_WV_LO
#undef RFX_WV_HI
#endif  // RFX_IMEL_PK

// one workgroup per clip: replays the reference's stopping rule on the clip-mean loss.  Thread (g, i) sums
// iteration i over every fourth frame (loads coalesce across i), the four partial sums meet in LDS, then one
// thread walks the max_iter means.
__global__ void __launch_bounds__(1024) imel_scan_kernel(const float* __restrict__ loss_hist, int* __restrict__ it_stop,
                                                        int* __restrict__ any_early, int nclips, int C, int T, int max_iter,
                                                        float tol_loss, float tol_change) {
  extern __shared__ float scan_smem[];  // [4][256] partial sums, then [max_iter] means
  float* part = scan_smem;
  float* mean = scan_smem + 1024;
  const int clip = blockIdx.x;
  if (clip >= nclips) return;
  const int nframes = C * T;
  const int i = threadIdx.x & 255, g = threadIdx.x >> 8;
  const float* base = loss_hist + (size_t)clip * nframes * max_iter;
  for (int it0 = 0; it0 < max_iter; it0 += 256) {
    const int it = it0 + i;
    float s0 = 0.f, s1 = 0.f;
    if (it < max_iter) {
      int f = g;
      for (; f + 4 < nframes; f += 8) {
        s0 += base[(size_t)f * max_iter + it];
        s1 += base[(size_t)(f + 4) * max_iter + it];
      }
      if (f < nframes) s0 += base[(size_t)f * max_iter + it];
    }
    part[g * 256 + i] = s0 + s1;
    __syncthreads();
    if (g == 0 && it < max_iter) mean[it] = ((part[i] + part[256 + i]) + (part[512 + i] + part[768 + i])) / (float)nframes;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float prev = __builtin_inff();
    int stop = max_iter;
    for (int it = 0; it < max_iter; ++it) {
      const float loss = mean[it];
      if (loss < tol_loss || fabsf(prev - loss) < tol_change) { stop = it + 1; break; }
      prev = loss;
    }
    it_stop[clip] = stop;
    if (stop < max_iter) atomicExch(any_early, 1);
  }
}

template <int FPW, int SET, bool UF>
static void launch_perwave(const ImelArgs& a, hipStream_t stream) {
  const int nframes = a.B * a.T;
  const size_t lds = FPW * imel_frame_lds_bytes(a.M, a.max_iter, a.tb.f_hi - a.tb.f_lo);
  constexpr const int* lo = SET == 0 ? kImelLoCap : kImelLoCapWide;
  constexpr const int* hi = SET == 0 ? kImelHiCap : kImelHiCapWide;
  // the wide set's class 0 holds 31 bins per thread (124 state registers): three waves per SIMD (168 VGPRs) instead of four
  constexpr int wpe = SET == 0 ? (UF ? RFX_IMEL_WAVES_PER_EU_UF : RFX_IMEL_WAVES_PER_EU) : 3;
  hipLaunchKernelGGL((imel_group_kernel_perwave<FPW, wpe, UF, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]>), dim3((nframes + FPW - 1) / FPW),
                     dim3(kImelThreads * FPW), lds, stream, a);
}

// Which kernel launch_imel runs for a bank / variant / step count: 4 wave, 5 line-form groups, 2 / 3 per-wave group budgets, 1 uniform
// groups, 0 the general LDS kernel.  ONE place decides (round 6): the launcher below, rfx_plan_imel_kernel and
// imel_can_emit_fam_slots all ask here, so a build without the packed kernels (RFX_IMEL_PK = 0) or a bank whose frame does not
// fit the 64 KB of LDS a kernel gets without the opt-in attribute falls back to the general kernel everywhere at once.
int imel_kernel_choice(const ImelTables& tb, int M, int max_iter, int variant) {
  const int band = tb.f_hi - tb.f_lo;
  constexpr size_t kPlainLds = 64 * 1024;  // dynamic LDS a kernel may ask for without hipFuncAttributeMaxDynamicSharedMemorySize
#if RFX_IMEL_PK && RFX_IMEL_WAVE
  if (tb.wave_ok && variant == 0 && sizeof(float) * ((size_t)max_iter + (size_t)band) <= kPlainLds) return 4;
#endif
  const size_t frame_lds = imel_frame_lds_bytes(M, max_iter, band);
#if RFX_IMEL_PK
  if (tb.fast_ok == 5 && variant == 0 && frame_lds <= kPlainLds) return 5;
#endif
  if (tb.fast_ok && tb.fast_ok != 5 && variant != 2 && !(variant == 1 && tb.fast_ok == 3) && frame_lds <= kPlainLds) {
    if (tb.fast_ok == 2 && variant != 1) return 2;
    if (tb.fast_ok == 3 && variant != 1) return 3;
    return 1;
  }
  return 0;
}

hipError_t launch_imel(const ImelArgs& a, int variant, hipStream_t stream) {
  const int choice = imel_kernel_choice(a.tb, a.M, a.max_iter, variant);
#if RFX_IMEL_PK && RFX_IMEL_WAVE
  if (choice == 4) {  // one wave per frame (the fix-up pass too: its frames are independent of each other)
    const size_t lds = sizeof(float) * ((size_t)a.max_iter + (size_t)(a.tb.f_hi - a.tb.f_lo));  // loss words + the epilogue's stage (16.8 KB: eight waves per CU)
    if (a.tb.unit_form) hipLaunchKernelGGL(imel_wave_kernel<true>, dim3(a.B * a.T), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL(imel_wave_kernel<false>, dim3(a.B * a.T), dim3(64), lds, stream, a);
    return hipGetLastError();
  }
#endif
#if RFX_IMEL_PK
  if (choice == 5) {  // long groups in line form (full-band banks, 384 filters ...)
    constexpr const int* lo = kImelLoCapLine;
    constexpr const int* hi = kImelHiCapLine;
    const size_t lds = imel_frame_lds_bytes(a.M, a.max_iter, a.tb.f_hi - a.tb.f_lo);
    if (a.tb.unit_form) hipLaunchKernelGGL((imel_line_kernel_perwave<true, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]>), dim3(a.B * a.T), dim3(kImelThreads), lds, stream, a);
    else hipLaunchKernelGGL((imel_line_kernel_perwave<false, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]>), dim3(a.B * a.T), dim3(kImelThreads), lds, stream, a);
    return hipGetLastError();
  }
#endif
  if (choice >= 1 && choice <= 3) {  // (the wide and line sets have no uniform fallback: general kernel)
    // the fix-up pass runs a different number of steps (and barriers) per clip: one frame per workgroup there
    const bool fixup = a.it_limit != nullptr;
    if (choice == 2) {
      if (a.tb.unit_form && RFX_IMEL_UFORM) {
        if (fixup) launch_perwave<1, 0, true>(a, stream); else launch_perwave<RFX_IMEL_FPW, 0, true>(a, stream);
      } else {
        if (fixup) launch_perwave<1, 0, false>(a, stream); else launch_perwave<RFX_IMEL_FPW, 0, false>(a, stream);
      }
    } else if (choice == 3) {
      if (a.tb.unit_form && RFX_IMEL_UFORM) {
        if (fixup) launch_perwave<1, 1, true>(a, stream); else launch_perwave<RFX_IMEL_FPW, 1, true>(a, stream);
      } else {
        if (fixup) launch_perwave<1, 1, false>(a, stream); else launch_perwave<RFX_IMEL_FPW, 1, false>(a, stream);
      }
    } else {
      hipLaunchKernelGGL((imel_group_kernel<8, 24>), dim3(a.B * a.T), dim3(kImelThreads), imel_frame_lds_bytes(a.M, a.max_iter, a.tb.f_hi - a.tb.f_lo), stream, a);
    }
    return hipGetLastError();
  }
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

// ---- numeric range (round 6): the powers of two the SGD and Griffin-Lim kernels work in, per clip / per row ----------------------
// max |x| per group as an integer key (the bits of a non-negative float order like the float; NaN is skipped, as fmaxf does)
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}
__global__ void __launch_bounds__(256) range_max_kernel(const float* __restrict__ x, size_t count, unsigned* __restrict__ keys) {
  const float* p = x + (size_t)blockIdx.y * count;
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(p[i]));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&keys[blockIdx.y], __float_as_uint(mx));
}
__global__ void __launch_bounds__(64) range_finish_kernel(const unsigned* __restrict__ keys, int groups, float hint, float* __restrict__ imel_scale,
                                                          float* __restrict__ gl_scale, int rows, int mel_units) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= groups) return;
  const float mx = hint > 0.f ? hint : __uint_as_float(keys[g]);
  int k = 0;  // mx in [2^(k-1), 2^k); an all-zero (or all-NaN) group works in the default units
  if (mx > 0.f) {
    if (mx < __builtin_inff()) (void)frexpf(mx, &k);
    else k = 129;
  }
  int e, j;
  range_exponents(k, mel_units, &e, &j);
  if (imel_scale) {
    // e = max(k + 35, 30): the clip's largest target near 2^-35 of the clamp's upper bound (2^-60 for max_value = 30e6, as in rounds
    // 2-5); never above 2^-30: the untouched bins start at U[0, 1) in the reference's units whatever the targets are
    imel_scale[2 * g] = ldexpf(1.f, -e);
    imel_scale[2 * g + 1] = ldexpf(1.f, e);
  }
  if (gl_scale) {
    // j = ks - 26: the row's largest magnitude near 2^25, what 30e6 gives unscaled (j = 0); ks = k, or max(k, 0) + 1 in mel units
    const float eps2 = fmaxf(ldexpf(1e-32f, -2 * j), 1.17549435e-38f);  // never zero: 0 * rsq(0 + 0) would be NaN where the reference gives 0
    for (int r = 0; r < rows; ++r) {
      gl_scale[2 * ((size_t)g * rows + r)] = ldexpf(1.f, -j);
      gl_scale[2 * ((size_t)g * rows + r) + 1] = eps2;
    }
  }
}
hipError_t launch_range_scale(const float* x, size_t count, int groups, float hint, unsigned* keys, float* imel_scale, float* gl_scale, int rows,
                              int mel_units, hipStream_t stream) {
  if (!(hint > 0.f)) {
    hipError_t e = hipMemsetAsync(keys, 0, sizeof(unsigned) * (size_t)groups, stream);
    if (e != hipSuccess) return e;
    size_t chunks = (count + 256 * 64 - 1) / (256 * 64);  // ~64 values per thread
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    for (int g0 = 0; g0 < groups; g0 += 65535) {  // (grid y is 16 bits wide)
      const int n = groups - g0 < 65535 ? groups - g0 : 65535;
      hipLaunchKernelGGL(range_max_kernel, dim3((unsigned)chunks, (unsigned)n), dim3(256), 0, stream, x + (size_t)g0 * count, count, keys + g0);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
  }
  hipLaunchKernelGGL(range_finish_kernel, dim3((groups + 63) / 64), dim3(64), 0, stream, keys, groups, hint, imel_scale, gl_scale, rows, mel_units);
  return hipGetLastError();
}

hipError_t launch_imel_scan(const float* loss_hist, int* it_stop, int* any_early, int nclips, int C, int T, int max_iter,
                            float tol_loss, float tol_change, hipStream_t stream) {
  hipLaunchKernelGGL(imel_scan_kernel, dim3(nclips), dim3(1024), sizeof(float) * (1024 + (size_t)max_iter), stream, loss_hist, it_stop, any_early, nclips, C, T,
                     max_iter, tol_loss, tol_change);
  return hipGetLastError();
}

}  // namespace rfx
