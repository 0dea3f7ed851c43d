// rfx_gen_core.h - per-thread arithmetic of the GENERIC framed transform: any STFT geometry whose n_fft factors
// into the radices below (every common sample rate: 48 kHz -> n_fft 19200, 32 kHz -> 12800, 22.05 kHz -> 8820,
// 16 kHz -> 6400, ... at the reference's default 400 / 100 / 10 ms; riffusion/spectrogram_params.py:62-81 derives the
// three lengths from the sample rate with int() truncation).  Written once for the gfx950 kernels (hipcc) and for the
// host-side emulator of the CPU tests (g++), like rfx_core.h for the specialised 44.1 kHz engine.
//
// A frame transform is a real FFT of length n_fft of the win_length windowed samples (centred, zero padded):
//   * n_fft even: the n_fft reals are packed as nc = n_fft/2 complex numbers z[n] = x[2n] + i x[2n+1], one complex
//     FFT of length nc, then the classic split  X[k] = (Z[k] + conj Z[nc-k])/2 - i w^k (Z[k] - conj Z[nc-k])/2,
//     w = exp(-2 pi i / n_fft);  the inverse retraces it;
//   * n_fft odd: a complex FFT of length nc = n_fft on (x, 0).
// The complex FFT runs one pass per radix R over the nc points in LDS, in one of two forms:
//   * in place (the kernels' default): forward = decimation in frequency - pass s splits blocks of length L into R
//     sub-blocks of length m = L / R:  y = DFT_R(buf[base + q m]);  buf[base + p m] = y[p] W_L^{i p}  (i = position inside
//     the sub-block) - leaving the spectrum digit-reversed; inverse = decimation in time over the same (L, m) pairs in
//     reverse order on digit-reversed input.  One buffer: two workgroups per CU at 48 kHz.
//   * Stockham autosort between two buffers, pass s combining sub-transforms of length Ns = prod of the earlier radices:
//       v[q] = in[j + q nc/R] * W_{Ns R}^{q (j mod Ns)},  y = DFT_R(v),  out[(j div Ns) Ns R + (j mod Ns) + q Ns] = y[q]
// Twiddles W_nc^t come from a two-level table (W^t = hi[t >> 7] * lo[t & 127], both small enough to sit in LDS next
// to the two buffers), the R-th roots of a pass are looked up once per thread and pass.
#pragma once
#include "rfx_core.h"

namespace rfx {

constexpr int kGenMaxStages = 16;
constexpr int kGenTwLo = 128;       // entries of the low twiddle table
constexpr int kGenMaxNc = 10000;    // two LDS buffers of nc complex numbers + tables must fit 160 KiB

// LDS position of element i of an FFT buffer.  (A pad element after every 32 - i + (i >> 5) - removes the 4- to 16-way
// bank conflicts of the early Stockham passes, 54 % of all LDS cycles by PMC; measured: no gain at 48 kHz, 7 % slower at
// 22.05 kHz - the kernels wait on barriers and dependent LDS round trips, not on LDS bandwidth.  Identity kept.)
RFX_HD int gen_pad(int i) { return i; }
RFX_HD int gen_buf_elems(int nc) { return nc; }

struct GenGeom {
  int n_fft, win, hop, n_stft;
  int nc;       // length of the complex FFT: n_fft/2 (even n_fft) or n_fft (odd)
  int even;     // 1: packed-real split
  int left;     // (n_fft - win) / 2: position of the first windowed sample inside the padded frame
  int fs;       // elements between consecutive frames of a [B*T][fs] array (n_stft rounded up to 64)
  int nhi;      // entries of the high twiddle table: ceil(nc / 128) (+1)
  int nhi2;     // entries of the high table of the split twiddles exp(-2 pi i k / n_fft), k <= nc
  int nstages;
  int radix[kGenMaxStages];
};

// radices the butterfly below implements; 4 first (fewest passes), then the primes
RFX_HD bool gen_factor(int n, int* radix, int* nstages) {
  const int cand[7] = {4, 2, 3, 5, 7, 11, 13};
  int ns = 0;
  for (int c = 0; c < 7; ++c)
    while (n % cand[c] == 0) {
      if (ns == kGenMaxStages) return false;
      radix[ns++] = cand[c];
      n /= cand[c];
    }
  *nstages = ns;
  return n == 1;
}

RFX_HD cf gen_tw(const cf* lo, const cf* hi, int t) { return cmul(hi[t >> 7], lo[t & (kGenTwLo - 1)]); }

// j div Ns for 0 <= j < 2^23 without an integer division (the GPU has none: ~40 instructions): float reciprocal,
// then one correction step either way.  Exact: j and Ns are exactly representable, the estimate is off by at most one.
RFX_HD int gen_div(int j, int Ns, float inv_ns) {
  int q = (int)((float)j * inv_ns);
  const int r = j - q * Ns;
  q += (r >= Ns) - (r < 0);
  return q;
}

// the R-point DFT of v into y (INV: exp(+i ...) kernels); root = exp(-2 pi i t / R), used by the O(R^2) radices only
template <int R, bool INV>
RFX_HD void gen_dft(const cf (&v)[R], cf (&y)[R], const cf (&root)[R]) {
  if (R == 2) {
    y[0] = cf{v[0].re + v[1].re, v[0].im + v[1].im};
    y[1] = cf{v[0].re - v[1].re, v[0].im - v[1].im};
  } else if (R == 4) {
    const cf a{v[0].re + v[2].re, v[0].im + v[2].im}, b{v[0].re - v[2].re, v[0].im - v[2].im};
    const cf c{v[1].re + v[3].re, v[1].im + v[3].im}, d{v[1].re - v[3].re, v[1].im - v[3].im};
    // forward: y1 = b - i d, y3 = b + i d ; inverse swaps them
    const cf md = INV ? cf{-d.im, d.re} : cf{d.im, -d.re};  // (-i d) forward, (+i d) inverse
    y[0] = cf{a.re + c.re, a.im + c.im};
    y[2] = cf{a.re - c.re, a.im - c.im};
    y[1] = cf{b.re + md.re, b.im + md.im};
    y[3] = cf{b.re - md.re, b.im - md.im};
  } else if (R == 3) {
    cf a = v[0], b = v[1], c = v[2];
    dft3<INV>(a, b, c);
    y[0] = a;
    y[1] = b;
    y[2] = c;
  } else if (R == 5) {
    // X0 = x0 + s1 + s2;  X1,4 = a1 -+ i b1;  X2,3 = a2 -+ i b2  (forward; the inverse swaps the signs)
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;  // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;   // sin(2 pi / 5), sin(4 pi / 5)
    const cf p1{v[1].re + v[4].re, v[1].im + v[4].im}, m1{v[1].re - v[4].re, v[1].im - v[4].im};
    const cf p2{v[2].re + v[3].re, v[2].im + v[3].im}, m2{v[2].re - v[3].re, v[2].im - v[3].im};
    const cf a1{fmaf(c2, p2.re, fmaf(c1, p1.re, v[0].re)), fmaf(c2, p2.im, fmaf(c1, p1.im, v[0].im))};
    const cf a2{fmaf(c1, p2.re, fmaf(c2, p1.re, v[0].re)), fmaf(c1, p2.im, fmaf(c2, p1.im, v[0].im))};
    const cf b1{fmaf(s2, m2.re, s1 * m1.re), fmaf(s2, m2.im, s1 * m1.im)};
    const cf b2{fmaf(-s1, m2.re, s2 * m1.re), fmaf(-s1, m2.im, s2 * m1.im)};
    y[0] = cf{v[0].re + p1.re + p2.re, v[0].im + p1.im + p2.im};
    const cf lo1{a1.re + b1.im, a1.im - b1.re}, hi1{a1.re - b1.im, a1.im + b1.re};  // a - i b, a + i b
    const cf lo2{a2.re + b2.im, a2.im - b2.re}, hi2{a2.re - b2.im, a2.im + b2.re};
    y[1] = INV ? hi1 : lo1;
    y[4] = INV ? lo1 : hi1;
    y[2] = INV ? hi2 : lo2;
    y[3] = INV ? lo2 : hi2;
  } else if (R == 7) {
    cf x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3], x4 = v[4], x5 = v[5], x6 = v[6];
    dft7<INV>(x0, x1, x2, x3, x4, x5, x6);
    y[0] = x0; y[1] = x1; y[2] = x2; y[3] = x3; y[4] = x4; y[5] = x5; y[6] = x6;
  } else {
#pragma unroll
    for (int p = 0; p < R; ++p) {
      cf acc = v[0];
#pragma unroll
      for (int q = 1; q < R; ++q) {
        const cf w = root[(q * p) % R];
        const cf t = INV ? cmulc(v[q], w) : cmul(v[q], w);
        acc.re += t.re;
        acc.im += t.im;
      }
      y[p] = acc;
    }
  }
}

template <int R, bool INV>
RFX_HD void gen_butterfly(const cf* in, cf* out, int j, int m, int Ns, float inv_ns, int tstep, const cf* lo, const cf* hi,
                          const cf (&root)[R]) {
  const int jd = gen_div(j, Ns, inv_ns);
  const int k = j - jd * Ns;
  cf v[R];
  v[0] = in[gen_pad(j)];
#pragma unroll
  for (int q = 1; q < R; ++q) {
    const cf w = gen_tw(lo, hi, q * k * tstep);  // q * k * tstep < nc
    const cf x = in[gen_pad(j + q * m)];
    v[q] = INV ? cmulc(x, w) : cmul(x, w);
  }
  cf y[R];
  gen_dft<R, INV>(v, y, root);
  const int j0 = jd * Ns * R + k;
#pragma unroll
  for (int p = 0; p < R; ++p) out[gen_pad(j0 + p * Ns)] = y[p];
}

// ---- in-place alternative (half the LDS: two workgroups per CU at 48 kHz).  Forward = decimation in frequency: pass s
// splits blocks of length L into R sub-blocks of length m = L / R (butterfly, then twiddle W_L^{i p}); the spectrum comes
// out digit-reversed (bin k at rev[k]).  Inverse = decimation in time over the same (L, m) pairs in reverse order (conjugate
// twiddle, then butterfly) on digit-reversed input, natural order out.
template <int R, bool INV>
RFX_HD void gen_ip_butterfly(cf* buf, int j, int m, float inv_m, int L, int tstep, const cf* lo, const cf* hi, const cf (&root)[R]) {
  const int blk = gen_div(j, m, inv_m);
  const int i = j - blk * m;
  const int base = blk * L + i;
  cf v[R], y[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const cf x = buf[base + q * m];
    v[q] = (INV && q > 0) ? cmulc(x, gen_tw(lo, hi, q * i * tstep)) : x;  // q * i * tstep < nc
  }
  gen_dft<R, INV>(v, y, root);
#pragma unroll
  for (int p = 0; p < R; ++p) buf[base + p * m] = (!INV && p > 0) ? cmul(y[p], gen_tw(lo, hi, p * i * tstep)) : y[p];
}
template <int R, bool INV>
RFX_HD void gen_ip_stage_r(cf* buf, int nc, int L, const cf* lo, const cf* hi, int tid, int nthr) {
  const int m = L / R, tstep = nc / L, nbf = nc / R;
  const float inv_m = 1.0f / (float)m;
  cf root[R];
#pragma unroll
  for (int t = 0; t < R; ++t) root[t] = gen_tw(lo, hi, t * (nc / R));
  for (int j = tid; j < nbf; j += nthr) gen_ip_butterfly<R, INV>(buf, j, m, inv_m, L, tstep, lo, hi, root);
}
template <bool INV, int MAXR = 13>
RFX_HD void gen_ip_stage(cf* buf, int nc, int L, int R, const cf* lo, const cf* hi, int tid, int nthr) {
  switch (R) {
    case 2: gen_ip_stage_r<2, INV>(buf, nc, L, lo, hi, tid, nthr); break;
    case 3: gen_ip_stage_r<3, INV>(buf, nc, L, lo, hi, tid, nthr); break;
    case 4: gen_ip_stage_r<4, INV>(buf, nc, L, lo, hi, tid, nthr); break;
    case 5: gen_ip_stage_r<5, INV>(buf, nc, L, lo, hi, tid, nthr); break;
    case 7: if (MAXR >= 7) gen_ip_stage_r<7, INV>(buf, nc, L, lo, hi, tid, nthr); break;
    case 11: if (MAXR >= 11) gen_ip_stage_r<11, INV>(buf, nc, L, lo, hi, tid, nthr); break;
    default: if (MAXR >= 13) gen_ip_stage_r<13, INV>(buf, nc, L, lo, hi, tid, nthr); break;
  }
}
// position of bin k after the forward in-place passes (and where the inverse expects it)
RFX_HD int gen_digit_reverse(const GenGeom& g, int k) {
  int pos = 0, len = g.nc;
  for (int s = 0; s < g.nstages; ++s) {
    len /= g.radix[s];
    pos += (k % g.radix[s]) * len;
    k /= g.radix[s];
  }
  return pos;
}

template <int R, bool INV>
RFX_HD void gen_stage_r(const cf* in, cf* out, int nc, int Ns, const cf* lo, const cf* hi, int tid, int nthr) {
  const int m = nc / R, tstep = nc / (Ns * R);
  const float inv_ns = 1.0f / (float)Ns;
  cf root[R];
#pragma unroll
  for (int t = 0; t < R; ++t) root[t] = gen_tw(lo, hi, t * m);  // exp(-2 pi i t / R)
  for (int j = tid; j < m; j += nthr) gen_butterfly<R, INV>(in, out, j, m, Ns, inv_ns, tstep, lo, hi, root);
}

// one Stockham pass, the share of thread `tid` of `nthr`.  MAXR is the largest radix the caller's geometry uses: the
// kernels are instantiated per class (5, 7, 13) so that a 48 kHz plan (radices 4, 2, 3, 5) is not compiled with the
// register footprint of the O(R^2) radix-13 butterfly (it spilled 131 VGPRs when every radix shared one kernel).
template <bool INV, int MAXR = 13>
RFX_HD void gen_stage(const cf* in, cf* out, int nc, int Ns, int R, const cf* lo, const cf* hi, int tid, int nthr) {
  switch (R) {
    case 2: gen_stage_r<2, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 3: gen_stage_r<3, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 4: gen_stage_r<4, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 5: gen_stage_r<5, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 7: if (MAXR >= 7) gen_stage_r<7, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 11: if (MAXR >= 11) gen_stage_r<11, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    default: if (MAXR >= 13) gen_stage_r<13, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
  }
}
// the class a radix list needs
RFX_HD int gen_radix_class(const int* radix, int nstages) {
  int mx = 0;
  for (int i = 0; i < nstages; ++i) mx = radix[i] > mx ? radix[i] : mx;
  return mx <= 5 ? 5 : mx <= 7 ? 7 : 13;
}

// ---- real <-> packed-complex split.  Z: the nc-point complex spectrum (LDS), lo2/hi2: two-level table of
// exp(-2 pi i k / n_fft).  Returns bin k (0 <= k <= n_fft/2) of the real FFT.
// `rev` (nullable): position of element k inside Z (digit-reversed after the in-place passes)
RFX_HD cf gen_split_forward(const GenGeom& g, const cf* Z, const cf* lo2, const cf* hi2, int k, const int* rev = nullptr) {
  auto at = [&](int i) { return Z[rev ? rev[i] : gen_pad(i)]; };
  if (!g.even) return at(k);
  const cf zk = at(k == g.nc ? 0 : k), zc = at(k == 0 ? 0 : g.nc - k);
  const cf s{zk.re + zc.re, zk.im - zc.im}, d{zk.re - zc.re, zk.im + zc.im};  // Z[k] +- conj Z[nc-k]
  const cf p = cmul(gen_tw(lo2, hi2, k), d);
  return cf{0.5f * (s.re + p.im), 0.5f * (s.im - p.re)};  // (s - i p) / 2
}
// element k (0 <= k < nc) of the complex spectrum whose inverse FFT yields the packed real signal, from the one-sided
// bins it is made of: even n_fft: xa = X[k], xb = X[nc - k]; odd n_fft: xa = X[k] (k <= (n_fft-1)/2) or X[n_fft - k], xb unused.
// Like torch.istft's irfft (pocketfft c2r) the imaginary parts of bins 0 and n_fft/2 are ignored.
RFX_HD cf gen_split_inverse_vals(const GenGeom& g, cf xa, cf xb, const cf* lo2, const cf* hi2, int k) {
  if (!g.even) {
    if (k == 0) return cf{xa.re, 0.f};
    return k <= (g.n_fft - 1) / 2 ? xa : cf{xa.re, -xa.im};
  }
  if (k == 0) {
    xa.im = 0.f;
    xb.im = 0.f;
  }
  const cf s{xa.re + xb.re, xa.im - xb.im}, d{xa.re - xb.re, xa.im + xb.im};  // X[k] +- conj X[nc-k]
  const cf p = cmulc(d, gen_tw(lo2, hi2, k));  // d * exp(+2 pi i k / n_fft)
  return cf{0.5f * (s.re - p.im), 0.5f * (s.im + p.re)};  // (s + i p) / 2
}
// which one-sided bins element k needs
RFX_HD int gen_split_bin_a(const GenGeom& g, int k) { return (g.even || k <= (g.n_fft - 1) / 2) ? k : g.n_fft - k; }
RFX_HD int gen_split_bin_b(const GenGeom& g, int k) { return g.even ? g.nc - k : 0; }
template <class XF>
RFX_HD cf gen_split_inverse(const GenGeom& g, XF X, const cf* lo2, const cf* hi2, int k) {
  return gen_split_inverse_vals(g, X(gen_split_bin_a(g, k)), X(gen_split_bin_b(g, k)), lo2, hi2, k);
}

// Griffin-Lim per-bin update of the generic path, in the reference's op order (torchaudio functional.griffinlim,
// SURVEY App. A.5): a = rebuilt - m * tprev ; angles = a / (|a| + 1e-16) ; next = S * angles.  The normalisation is
// gl_project of rfx_core.h (one v_rsq_f32 on the device: the IEEE sqrt + two divisions it replaces were 30 % of this
// kernel's instructions; exact sqrt / divide on the host).
RFX_HD cf gen_gl_update(cf rebuilt, cf tprev, float mom, float S) {
  return gl_project(cf{rebuilt.re - tprev.re * mom, rebuilt.im - tprev.im * mom}, S);
}

}  // namespace rfx
