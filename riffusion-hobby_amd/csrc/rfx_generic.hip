// rfx_generic.hip - the framed transform and Griffin-Lim for ANY STFT geometry (replaces the same torchaudio
// modules as rfx_stft.hip / rfx_gl.hip: Spectrogram(power=None) and GriffinLim, riffusion/spectrogram_converter.py:47-73).
//
// The 44.1 kHz geometry of the reference's defaults runs on the specialised engine (rfx_core.h: hard-wired 40 x 21 x 21
// factorisation, one fused launch per Griffin-Lim iteration).  Every other sample rate / window / padding the reference
// accepts (cli.py:43 takes the rate from the input file; spectrogram_params.py:62-81) runs here: an in-place FFT over a
// runtime radix list in LDS (rfx_gen_core.h), one workgroup per frame.  Griffin-Lim is two launches per iteration:
//     gen_gl_kernel    per frame: analysis of x_k - m x_{k-1} (the momentum applied in the time domain: the STFT is linear,
//                      see rfx_gl.hip), forward passes, Z = S * a / (|a| + 1e-16) pairwise IN PLACE on the packed spectrum,
//                      inverse passes, windowed synthesis frame -> HBM
//     gen_fold_kernel  overlap-add of the frames and division by the window envelope -> x_{k+1}
// No spectrum ever reaches HBM: per bin and iteration only |S| (4 B) is read, plus the windowed frames (win floats per frame
// written and read once) and the two L2-resident audio estimates.  (Round 2 ran the reference's op order with `rebuilt`,
// `tprev` and Z in HBM - 36 B per bin and iteration, three launches.)  Bit-reproducible (no atomics), same entry points.
#include <hip/hip_runtime.h>

#include "rfx_gen_core.h"
#include "rfx_kernels.h"

namespace rfx {

// RFX_GEN_INPLACE=1: in-place passes (rfx_gen_core.h) in ONE LDS buffer, so two 512-thread workgroups share a CU at 48 kHz
// and cover each other's barrier / LDS waits; 0: Stockham autosort between two buffers, one 1024-thread workgroup per CU.
#ifndef RFX_GEN_INPLACE
#define RFX_GEN_INPLACE 1
#endif
#ifndef RFX_GEN_THREADS
#define RFX_GEN_THREADS (RFX_GEN_INPLACE ? 512 : 1024)
#endif
constexpr int kGenThreads = RFX_GEN_THREADS;

struct GenLds {
  cf* a;
  cf* b;
  cf* lo;   // [128]  exp(-2 pi i t / nc)
  cf* hi;   // [nhi]  exp(-2 pi i 128 t / nc)
  cf* lo2;  // [128]  exp(-2 pi i t / n_fft)
  cf* hi2;  // [nhi2] exp(-2 pi i 128 t / n_fft)
};

__device__ __forceinline__ GenLds gen_lds(char* smem, const GenGeom& g, const GenTables& tb) {
  GenLds l;
  l.a = reinterpret_cast<cf*>(smem);
  l.b = l.a + (RFX_GEN_INPLACE ? 0 : gen_buf_elems(g.nc));
  l.lo = l.b + (RFX_GEN_INPLACE ? gen_ibuf_elems(g.nc, g.pad_shift) : gen_buf_elems(g.nc));
  l.hi = l.lo + kGenTwLo;
  l.lo2 = l.hi + g.nhi;
  l.hi2 = l.lo2 + kGenTwLo;
  for (int i = threadIdx.x; i < kGenTwLo; i += blockDim.x) {
    l.lo[i] = tb.lo[i];
    l.lo2[i] = tb.lo2[i];
  }
  for (int i = threadIdx.x; i < g.nhi; i += blockDim.x) l.hi[i] = tb.hi[i];
  for (int i = threadIdx.x; i < g.nhi2; i += blockDim.x) l.hi2[i] = tb.hi2[i];
  return l;
}

size_t gen_lds_bytes(const GenGeom& g) {
  return sizeof(cf) * ((RFX_GEN_INPLACE ? (size_t)gen_ibuf_elems(g.nc, g.pad_shift) : 2 * (size_t)gen_buf_elems(g.nc)) + 2 * kGenTwLo + g.nhi + g.nhi2);
}

// all passes of the nc-point FFT; data starts in l.a, the result's buffer is returned.  Barriers inside.
template <bool INV, int MAXR>
__device__ __forceinline__ cf* gen_fft(const GenGeom& g, const GenLds& l, const cf* __restrict__ tw) {
#if RFX_GEN_INPLACE
  int L = INV ? 1 : g.nc;  // forward: blocks shrink from nc; inverse: they grow from the last radix
  int off = INV ? gen_tw_table_elems(g) : 0;  // this pass's slice of the exact twiddle tables
  for (int i = 0; i < g.nstages; ++i) {
    const int R = g.radix[INV ? g.nstages - 1 - i : i];
    if (INV) {
      L *= R;
      off -= (L / R) * (R - 1);
    }
    __syncthreads();
    gen_ip_stage<INV, MAXR>(l.a, g.nc, L, R, l.lo, l.hi, (int)threadIdx.x, (int)blockDim.x, g.pad_shift, tw + off);
    if (!INV) {
      off += (L / R) * (R - 1);
      L /= R;
    }
  }
  __syncthreads();
  return l.a;
#endif
  cf* in = l.a;
  cf* out = l.b;
  int Ns = 1;
  for (int s = 0; s < g.nstages; ++s) {
    __syncthreads();
    gen_stage<INV, MAXR>(in, out, g.nc, Ns, g.radix[s], l.lo, l.hi, (int)threadIdx.x, (int)blockDim.x);
    Ns *= g.radix[s];
    cf* t = in;
    in = out;
    out = t;
  }
  __syncthreads();
  return in;
}

// ---- forward: frame fr of clip b is centred on sample hop*fr of the reflect-padded waveform (torch.stft center=True)
enum GenStftMode { kGenMag = 0, kGenSpec = 1 };

// (Prefetching the next frame's samples and the epilogue's |S| / tprev into register slots across the passes was tried:
// 229 -> 245-257 ms per 64 tiles at 48 kHz - the slot arrays spill.  The simple loops below stay.)
// (four waves per SIMD = 128 VGPRs, so that two 512-thread workgroups share a CU: without the bound the compiler took 169
// registers for the batched butterflies and the second workgroup no longer fitted - 121 -> 182 ms per 64 tiles at 48 kHz;
// the O(R^2) radix-11 / 13 class keeps its larger budget)
template <int MODE, int MAXR>
__global__ void __launch_bounds__(kGenThreads) __attribute__((amdgpu_waves_per_eu(MAXR <= 7 ? 4 : 2)))
gen_stft_kernel(GenStftArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GenGeom& g = a.g;
  const GenLds l = gen_lds(smem, g, a.tb);
  const long long nframes = (long long)a.B * a.T;
  const int half = g.n_fft / 2;
  for (long long fr = blockIdx.x; fr < nframes; fr += gridDim.x) {
    const int clip = (int)(fr / a.T), t = (int)(fr - (long long)clip * a.T);
    const float* __restrict__ x = a.wave + (size_t)clip * a.wave_stride;
    __syncthreads();  // previous frame's epilogue is done with the buffers
    // windowed, zero-padded frame, packed two reals per complex when n_fft is even
    for (int n = threadIdx.x; n < g.nc; n += blockDim.x) {
      float v[2] = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!g.even && e == 1) break;
        const int i = g.even ? 2 * n + e : n;  // position inside the padded frame
        const int j = i - g.left;              // position inside the window
        if (j >= 0 && j < g.win) v[e] = x[reflect_index(g.hop * t + i - half, a.Lw)] * a.tb.win[j];
      }
      l.a[RFX_GEN_INPLACE ? gen_ipad(n, g.pad_shift) : gen_pad(n)] = cf{v[0], v[1]};
    }
    const cf* Z = gen_fft<false, MAXR>(g, l, a.tb.tw);
    const size_t base = (size_t)fr * g.fs;
    for (int k = threadIdx.x; k < g.fs; k += blockDim.x) {
      if (k >= g.n_stft) {  // padding of the frame stride: keep it zero
        if (MODE == kGenMag) a.mag[base + k] = 0.f;
        if (MODE == kGenSpec) a.spec[base + k] = cf{0.f, 0.f};
        continue;
      }
      const cf X = gen_split_forward(g, Z, l.lo2, l.hi2, k, RFX_GEN_INPLACE ? a.tb.rev : nullptr);
      if (MODE == kGenMag) a.mag[base + k] = sqrtf(fmaf(X.re, X.re, X.im * X.im));
      if (MODE == kGenSpec) a.spec[base + k] = X;
    }
  }
}

// ---- Griffin-Lim, one iteration for one frame per trip.  MODE 0: Z = S * angles0 (injected or drawn) -> synthesis;
// MODE 1: analysis of x_0 (no momentum term yet); MODE 2: analysis of x_k - m x_{k-1}.
template <int MODE, int MAXR>
__global__ void __launch_bounds__(kGenThreads) __attribute__((amdgpu_waves_per_eu(MAXR <= 7 ? 4 : 2)))
gen_gl_kernel(GenGlArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const GenGeom& g = a.g;
  const GenLds l = gen_lds(smem, g, a.tb);
  const long long nframes = (long long)a.B * a.T;
  const int half = g.n_fft / 2;
  const float scale = 1.0f / (float)g.nc;  // even: z = IFFT_nc(Z) ; odd: x = Re IFFT_n(Z)
  const int npairs = gen_pair_count(g);
#ifdef RFX_GEN_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
  int nfr = 0;
#define GSTAMP(i) do { unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define GSTAMP(i) ((void)0)
#endif
  for (long long fr = blockIdx.x; fr < nframes; fr += gridDim.x) {
    const int clip = (int)(fr / a.T), t = (int)(fr - (long long)clip * a.T);
    const float eps2 = a.row_scale ? a.row_scale[2 * clip + 1] : 1e-32f;
    (void)eps2;
    const size_t base = (size_t)fr * g.fs;
    const float* __restrict__ S = a.S + base;
    __syncthreads();  // the previous frame's output loop is done with the buffer
    GSTAMP(0);
    if (MODE == 0) {
      auto X = [&](int k) {
        cf ang;
        if (a.angles0) ang = a.angles0[base + k];
        else ang = rand_unit_pair(rand_frame_key(a.seed, a.frame_base + (unsigned long long)fr), k);
        const float s = S[k];
        return cf{s * ang.re, s * ang.im};
      };
      for (int k = threadIdx.x; k < g.nc; k += blockDim.x) l.a[a.tb.rev[k]] = gen_split_inverse(g, X, l.lo2, l.hi2, k);
    } else {
      // windowed, zero-padded frame of x_k - m x_{k-1} (reflect-padded like torch.stft center=True), packed two reals per
      // complex when n_fft is even.  Only the elements the window covers need loads ([n_lo, n_hi): a quarter of the frame at
      // the reference's 100 / 400 ms); the rest is zeroed.  Loads are batched - all of a batch's global loads, then its LDS
      // stores - so that a thread waits for HBM / L2 once per batch, not once per element.
      const float* __restrict__ xc = a.x_cur + (size_t)clip * a.audio_stride;
      const float* __restrict__ xp = a.x_prev + (size_t)clip * a.audio_stride;
      const int nthr = (int)blockDim.x;
      const int per = g.even ? 2 : 1;
      const int n_lo = g.left / per, n_hi = (g.left + g.win + per - 1) / per;
      for (int n = threadIdx.x; n < g.nc; n += nthr)
        if (n < n_lo || n >= n_hi) l.a[gen_ipad(n, g.pad_shift)] = cf{0.f, 0.f};
      constexpr int UL = 5;
      for (int n0 = n_lo + (int)threadIdx.x; n0 < n_hi; n0 += UL * nthr) {
        float xs[UL][2], ps[UL][2], ws[UL][2];
#pragma unroll
        for (int u = 0; u < UL; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            xs[u][e] = ps[u][e] = ws[u][e] = 0.f;
            const int n = n0 + u * nthr;
            if (e < per && n < n_hi) {
              const int i = per * n + e;  // position inside the padded frame
              const int j = i - g.left;   // position inside the window
              if (j >= 0 && j < g.win) {
                const int p = reflect_index(g.hop * t + i - half, a.L);
                xs[u][e] = xc[p];
                if (MODE == 2) ps[u][e] = xp[p];
                ws[u][e] = a.tb.win[j];
              }
            }
          }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int n = n0 + u * nthr;
          if (n < n_hi)
            l.a[gen_ipad(n, g.pad_shift)] = cf{fmaf(-a.mom, ps[u][0], xs[u][0]) * ws[u][0], fmaf(-a.mom, ps[u][1], xs[u][1]) * ws[u][1]};
        }
      }
      __syncthreads();
      GSTAMP(1);
      gen_fft<false, MAXR>(g, l, a.tb.tw);  // ends with a barrier; spectrum digit-reversed in l.a
      GSTAMP(2);
      {
        // split / projection / merge, pairwise in place (gen_pair_compute).  Batches of UP pairs per thread: first every
        // global load of the batch (LDS positions from the digit-reversal table, |S| from HBM), then the LDS reads that depend
        // on them, the arithmetic, the stores - one global round trip and one LDS round trip per batch.
        constexpr int UP = 5;
        const int kc_of0 = g.even ? g.nc : 0;
        for (int k0 = threadIdx.x; k0 < npairs; k0 += UP * nthr) {
          GenPair pr[UP];
          int pk[UP], pc[UP];
#pragma unroll
          for (int u = 0; u < UP; ++u) {
            const int k = k0 + u * nthr;
            const bool ok = k < npairs;
            const int kk = ok ? k : 0;
            const int kc = g.even ? g.nc - kk : (kk == 0 ? 0 : g.n_fft - kk);  // partner ELEMENT (odd n_fft: the mirror element)
            pr[u].k = kk;
            pk[u] = a.tb.rev[kk];
            pc[u] = a.tb.rev[kc == kc_of0 && g.even ? 0 : kc];
            pr[u].sk = S[kk];
            pr[u].sc = g.even ? S[kc] : 0.f;  // even, k == 0: bin nc
          }
#pragma unroll
          for (int u = 0; u < UP; ++u) {
            pr[u].zk = l.a[pk[u]];
            pr[u].zc = g.even ? l.a[pc[u]] : pr[u].zk;
          }
#pragma unroll
          for (int u = 0; u < UP; ++u) gen_pair_compute(pr[u], g, l.lo2, l.hi2, eps2);
#pragma unroll
          for (int u = 0; u < UP; ++u) {
            const int k = k0 + u * nthr;
            if (k < npairs) {
              l.a[pk[u]] = pr[u].zk;
              const bool partner = g.even ? (k != 0 && k != g.nc - k) : k != 0;
              if (partner) l.a[pc[u]] = pr[u].zc;
            }
          }
        }
      }
    }
#ifdef RFX_GEN_TIMING
    __syncthreads();
#endif
    GSTAMP(3);
    const cf* z = gen_fft<true, MAXR>(g, l, a.tb.tw);  // starts with a barrier
    GSTAMP(4);
    float* __restrict__ out = a.frames + (size_t)fr * g.fpitch + g.fshift;
    {
      constexpr int UO = 5;  // window samples fetched per batch before the stores
      const int nthr = (int)blockDim.x;
      for (int j0 = threadIdx.x; j0 < g.win; j0 += UO * nthr) {
        float wv[UO], zv[UO];
#pragma unroll
        for (int u = 0; u < UO; ++u) {
          const int j = j0 + u * nthr;
          const int jj = j < g.win ? j : 0;
          const int i = jj + g.left;
          wv[u] = a.tb.win[jj];
          const cf zz = z[gen_ipad(g.even ? i >> 1 : i, g.pad_shift)];
          zv[u] = (g.even && (i & 1)) ? zz.im : zz.re;
        }
#pragma unroll
        for (int u = 0; u < UO; ++u) {
          const int j = j0 + u * nthr;
          if (j < g.win) out[j] = zv[u] * scale * wv[u];
        }
      }
    }
    GSTAMP(5);
#ifdef RFX_GEN_TIMING
    ++nfr;
#endif
  }
#ifdef RFX_GEN_TIMING
  if (MODE == 2 && blockIdx.x == 7 && threadIdx.x == 0)
    printf("gen_gl timing (100 MHz ticks per frame, %d frames): barrier-wait %.1f load %.1f fwd %.1f pair %.1f inv %.1f out %.1f\n", nfr,
           (double)tacc[0] / nfr, (double)tacc[1] / nfr, (double)tacc[2] / nfr, (double)tacc[3] / nfr, (double)tacc[4] / nfr, (double)tacc[5] / nfr);
#endif
}

// overlap-add of the windowed frames and division by the window envelope (torch.istft center=True, length = hop*(T-1)):
// sample p of the output sits at P = p + n_fft/2 of the padded signal; frame t contributes its window sample
// j = P - hop*t - left.  Fixed summation order (t ascending): bit-reproducible.  The envelope sum_t w[j]^2 depends on p only and
// is computed once per call (gen_env_kernel; the fold recomputed it per sample and iteration: ten window loads and FMAs next to
// the ten frame loads); where the geometry allows it (window, hop, left margin and lengths multiples of four: 48 kHz, 16 kHz ...)
// a thread folds four consecutive samples with 16-byte accesses - they share their frame range.  Same arithmetic, same bits.
__device__ __forceinline__ void gen_fold_range(const GenGeom& g, int q, int T, int& tlo, int& thi) {
  tlo = q - (g.win - 1) <= 0 ? 0 : (q - (g.win - 1) + g.hop - 1) / g.hop;
  thi = q / g.hop;
  if (thi > T - 1) thi = T - 1;
}
__global__ void __launch_bounds__(256) gen_env_kernel(const float* __restrict__ win, float* __restrict__ env, GenGeom g, int T, int L) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= L) return;
  const int q = p + g.n_fft / 2 - g.left;  // j = q - hop*t
  int tlo, thi;
  gen_fold_range(g, q, T, tlo, thi);
  float e = 0.f;
  for (int t = tlo; t <= thi; ++t) {
    const float w = win[q - g.hop * t];
    e = fmaf(w, w, e);
  }
  env[p] = e;
}
// Optional second output (row-family Griffin-Lim, round 4): the signal the NEXT iteration analyses, d = x_{k+1} - m x_k (x_k =
// `prev`; d = x_{k+1} when prev is null: the first iteration has no momentum term).  The Griffin-Lim kernel then fetches one
// value per window sample instead of two and needs ten registers less across its last phase; the fold has x_{k+1} in a register
// anyway.  Same fma as the kernel used to do: same bits.
__global__ void __launch_bounds__(256) gen_fold_kernel(const float* __restrict__ frames, const float* __restrict__ env,
                                                       float* __restrict__ out, GenGeom g, int B, int T, int L, size_t out_stride,
                                                       const float* __restrict__ prev, float* __restrict__ dout, float mom,
                                                       const float* __restrict__ row_scale) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= L) return;
  const int q = p + g.n_fft / 2 - g.left;
  int tlo, thi;
  gen_fold_range(g, q, T, tlo, thi);
  float acc = 0.f;
  for (int t = tlo; t <= thi; ++t) acc += frames[((size_t)b * T + t) * g.fpitch + g.fshift + (q - g.hop * t)];
  const float x = acc / env[p];
  out[(size_t)b * out_stride + p] = x;
  const float ks = row_scale ? row_scale[2 * b] : 1.f;  // a power of two (GlArgs::row_scale)
  if (dout) dout[(size_t)b * out_stride + p] = ks * (prev ? fmaf(-mom, prev[(size_t)b * out_stride + p], x) : x);
}
__global__ void __launch_bounds__(256) gen_fold4_kernel(const float* __restrict__ frames, const float* __restrict__ env,
                                                        float* __restrict__ out, GenGeom g, int B, int T, int L, size_t out_stride,
                                                        const float* __restrict__ prev, float* __restrict__ dout, float mom,
                                                        const float* __restrict__ row_scale) {
  using v4 = float __attribute__((ext_vector_type(4)));
  const int p = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  const int b = blockIdx.y;
  if (p >= L) return;
  // q + fshift is a multiple of four, like hop and fpitch: every frame is read 16 aligned bytes at a time.  Plain layout: q .. q + 3
  // share their frame range.  Padded layout (gen_frame_layout): the window's ends may fall inside the group - the frames of q's
  // first and (q + 3)'s last are all read, and a sample outside a frame's window reads the zero padding of its row (x + 0 = x:
  // the same sums as the scalar fold, term for term)
  const int q = p + g.n_fft / 2 - g.left;
  int tlo, thi, tmp;
  gen_fold_range(g, q, T, tlo, tmp);
  gen_fold_range(g, q + 3, T, tmp, thi);
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = tlo; t <= thi; ++t) acc += *reinterpret_cast<const v4*>(frames + ((size_t)b * T + t) * g.fpitch + g.fshift + (q - g.hop * t));
  const v4 e = *reinterpret_cast<const v4*>(env + p);
  const v4 x = v4{acc.x / e.x, acc.y / e.y, acc.z / e.z, acc.w / e.w};
  *reinterpret_cast<v4*>(out + (size_t)b * out_stride + p) = x;
  if (dout) {
    v4 d = x;
    if (prev) {
      const v4 xp = *reinterpret_cast<const v4*>(prev + (size_t)b * out_stride + p);
      d = v4{fmaf(-mom, xp.x, x.x), fmaf(-mom, xp.y, x.y), fmaf(-mom, xp.z, x.z), fmaf(-mom, xp.w, x.w)};
    }
    const float ks = row_scale ? row_scale[2 * b] : 1.f;
    *reinterpret_cast<v4*>(dout + (size_t)b * out_stride + p) = v4{ks * d.x, ks * d.y, ks * d.z, ks * d.w};
  }
}

// ---- (B, F, T) <-> [B*T][fs] layout conversion (tiled transposes; the padding of the frame stride is zeroed)
template <class V>
__global__ void __launch_bounds__(256) gen_pack_kernel(const V* __restrict__ bft, V* __restrict__ frames, int F, int T, int fs) {
  __shared__ V tile[32][33];
  const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32, b = blockIdx.z;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  for (int r = ly; r < 32; r += 8) {
    const int f = f0 + r, t = t0 + lx;
    tile[r][lx] = (f < F && t < T) ? bft[((size_t)b * F + f) * T + t] : V{};
  }
  __syncthreads();
  for (int r = ly; r < 32; r += 8) {
    const int t = t0 + r, f = f0 + lx;
    if (t < T && f < fs) frames[((size_t)b * T + t) * fs + f] = tile[lx][r];
  }
}
template <class V>
__global__ void __launch_bounds__(256) gen_unpack_kernel(const V* __restrict__ frames, V* __restrict__ bft, int F, int T, int fs) {
  __shared__ V tile[32][33];
  const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32, b = blockIdx.z;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  for (int r = ly; r < 32; r += 8) {
    const int t = t0 + r, f = f0 + lx;
    tile[r][lx] = (t < T && f < F) ? frames[((size_t)b * T + t) * fs + f] : V{};
  }
  __syncthreads();
  for (int r = ly; r < 32; r += 8) {
    const int f = f0 + r, t = t0 + lx;
    if (f < F && t < T) bft[((size_t)b * F + f) * T + t] = tile[lx][r];
  }
}

// ---- banded mel projection of magnitude frames [B*T][fs] -> frame-major (B, T, Mpad), one workgroup per frame.  The frame's
// active bins [f_lo, f_lo + nb) are staged in LDS (whole-line loads), and a filter's weights (column m of band_wt: zero past
// the filter's end, the row count a multiple of eight) are requested eight at a time before they are summed - in increasing
// bin order, like the reference's matmul row.  (The first version read one weight and one magnitude per trip from L2, each
// trip waiting for the last: 1.2 ms per 64 x 512 frames at 48 kHz, more than the transform itself.)
__global__ void __launch_bounds__(256) gen_mel_kernel(const float* __restrict__ mag, float* __restrict__ mel_tm,
                                                      const float* __restrict__ band_wt, const int* __restrict__ band_lo,
                                                      const int* __restrict__ band_len, int fs, int M, int Mpad, int f_lo, int nb) {
  extern __shared__ float row_s[];  // [nb]
  const size_t fr = blockIdx.x;
  const float* __restrict__ row = mag + fr * fs + f_lo;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) row_s[i] = row[i];
  __syncthreads();
  for (int m = threadIdx.x; m < Mpad; m += blockDim.x) {
    float s = 0.f;
    if (m < M) {
      const int lo = band_lo[m] - f_lo, n = band_len[m];
      const float* __restrict__ wcol = band_wt + m;
      for (int i = 0; i < n; i += 8) {
        float w[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          w[u] = wcol[(size_t)(i + u) * Mpad];
          const int q = lo + i + u;
          v[u] = row_s[q < nb ? q : nb - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(w[u], v[u], s);
      }
    }
    mel_tm[fr * Mpad + m] = s;
  }
}

// --------------------------------------------------------------------------------------------------------------------
static int gen_grid(const GenGeom& g, int num_cus, long long nframes) {
  // resident workgroups: LDS bound (160 KiB per CU) and register bound (128 VGPRs: 16 waves per CU)
  const size_t lds = gen_lds_bytes(g);
  int per_cu = (int)((160u * 1024u) / (lds + 512));
  if (per_cu < 1) per_cu = 1;
  const int by_waves = 1024 / g.nthr;
  if (per_cu > by_waves) per_cu = by_waves < 1 ? 1 : by_waves;
  long long n = (long long)num_cus * per_cu;
  return (int)(n < nframes ? n : nframes);
}

// kernels by (mode, radix class)
using GenStftFn = void (*)(GenStftArgs);
using GenGlFn = void (*)(GenGlArgs);
template <int MAXR>
static GenStftFn gen_stft_fn(int mode) {
  return mode == kGenMag ? gen_stft_kernel<kGenMag, MAXR> : gen_stft_kernel<kGenSpec, MAXR>;
}
static GenStftFn gen_stft_fn(const GenGeom& g, int mode) {
  const int c = gen_radix_class(g.radix, g.nstages);
  return c == 5 ? gen_stft_fn<5>(mode) : c == 7 ? gen_stft_fn<7>(mode) : gen_stft_fn<13>(mode);
}
template <int MAXR>
static GenGlFn gen_gl_fn(int mode) {
  return mode == 0 ? gen_gl_kernel<0, MAXR> : mode == 1 ? gen_gl_kernel<1, MAXR> : gen_gl_kernel<2, MAXR>;
}
static GenGlFn gen_gl_fn(const GenGeom& g, int mode) {
  const int c = gen_radix_class(g.radix, g.nstages);
  return c == 5 ? gen_gl_fn<5>(mode) : c == 7 ? gen_gl_fn<7>(mode) : gen_gl_fn<13>(mode);
}

hipError_t prepare_generic_kernels(const GenGeom& g) {
  const int lds = (int)gen_lds_bytes(g);
  hipError_t e;
  for (int mode = 0; mode < 2; ++mode)
    if ((e = hipFuncSetAttribute((const void*)gen_stft_fn(g, mode), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  for (int mode = 0; mode < 3; ++mode)
    if ((e = hipFuncSetAttribute((const void*)gen_gl_fn(g, mode), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  return hipSuccess;
}

hipError_t launch_gen_stft(int mode, const GenStftArgs& a, int num_cus, hipStream_t stream) {
  const int grid = gen_grid(a.g, num_cus, (long long)a.B * a.T);
  hipLaunchKernelGGL(gen_stft_fn(a.g, mode), dim3(grid), dim3(a.g.nthr), gen_lds_bytes(a.g), stream, a);
  return hipGetLastError();
}

hipError_t launch_gen_gl(int mode, const GenGlArgs& a, int num_cus, hipStream_t stream) {
  const int grid = gen_grid(a.g, num_cus, (long long)a.B * a.T);
  hipLaunchKernelGGL(gen_gl_fn(a.g, mode), dim3(grid), dim3(a.g.nthr), gen_lds_bytes(a.g), stream, a);
  return hipGetLastError();
}

hipError_t launch_gen_env(const float* win, float* env, const GenGeom& g, int T, int L, hipStream_t stream) {
  hipLaunchKernelGGL(gen_env_kernel, dim3((L + 255) / 256), dim3(256), 0, stream, win, env, g, T, L);
  return hipGetLastError();
}
hipError_t launch_gen_fold(const float* frames, const float* env, float* out, const GenGeom& g, int B, int T, int L, size_t out_stride,
                           hipStream_t stream, const float* prev, float* dout, float mom, const float* row_scale) {
  const bool vec = g.fpitch % 4 == 0 && g.hop % 4 == 0 && (g.n_fft / 2 - g.left + g.fshift) % 4 == 0 && L % 4 == 0 && out_stride % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(frames) & 15) == 0 && (reinterpret_cast<uintptr_t>(env) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(prev) & 15) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0;
  if (vec) hipLaunchKernelGGL(gen_fold4_kernel, dim3((L / 4 + 255) / 256, B), dim3(256), 0, stream, frames, env, out, g, B, T, L, out_stride, prev, dout, mom, row_scale);
  else hipLaunchKernelGGL(gen_fold_kernel, dim3((L + 255) / 256, B), dim3(256), 0, stream, frames, env, out, g, B, T, L, out_stride, prev, dout, mom, row_scale);
  return hipGetLastError();
}

hipError_t launch_gen_pack(const void* bft, void* frames, bool complex_, int B, int F, int T, int fs, hipStream_t stream) {
  dim3 grid((fs + 31) / 32, (T + 31) / 32, B);
  if (complex_) hipLaunchKernelGGL(gen_pack_kernel<float2>, grid, dim3(256), 0, stream, (const float2*)bft, (float2*)frames, F, T, fs);
  else hipLaunchKernelGGL(gen_pack_kernel<float>, grid, dim3(256), 0, stream, (const float*)bft, (float*)frames, F, T, fs);
  return hipGetLastError();
}
hipError_t launch_gen_unpack(const void* frames, void* bft, bool complex_, int B, int F, int T, int fs, hipStream_t stream) {
  dim3 grid((F + 31) / 32, (T + 31) / 32, B);
  if (complex_) hipLaunchKernelGGL(gen_unpack_kernel<float2>, grid, dim3(256), 0, stream, (const float2*)frames, (float2*)bft, F, T, fs);
  else hipLaunchKernelGGL(gen_unpack_kernel<float>, grid, dim3(256), 0, stream, (const float*)frames, (float*)bft, F, T, fs);
  return hipGetLastError();
}

hipError_t launch_gen_mel(const float* mag, float* mel_tm, const float* band_wt, const int* band_lo, const int* band_len, long long nframes,
                          int fs, int M, int Mpad, int f_lo, int f_hi, hipStream_t stream) {
  const int nb = f_hi - f_lo;
  hipLaunchKernelGGL(gen_mel_kernel, dim3((unsigned)nframes), dim3(256), sizeof(float) * (size_t)nb, stream, mag, mel_tm, band_wt, band_lo, band_len, fs, M, Mpad,
                     f_lo, nb);
  return hipGetLastError();
}

}  // namespace rfx
