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
constexpr int kGenMaxNc = 20000;    // hard bound on the complex FFT length; the real limit is LDS: the in-place buffer of nc complex numbers
                                    // (+ padding) and the twiddle tables must fit the 160 KiB of a CU (checked at plan creation)

// LDS position of element i of an FFT buffer.  (A pad element after every 32 - i + (i >> 5) - removes the 4- to 16-way
// bank conflicts of the early Stockham passes, 54 % of all LDS cycles by PMC; measured: no gain at 48 kHz, 7 % slower at
// 22.05 kHz - the kernels wait on barriers and dependent LDS round trips, not on LDS bandwidth.  Identity kept.)
RFX_HD int gen_pad(int i) { return i; }
RFX_HD int gen_buf_elems(int nc) { return nc; }
// The in-place passes (the kernels' form) DO pad, by a per-geometry shift chosen at plan creation (gen_pick_pad): element i
// sits at i + (i >> ps), ps = 0 meaning no padding.  What it buys: after the forward passes bin k sits at the digit-reversed
// position rev[k], and consecutive k - the lanes of a wave in the split / projection step - are nc / R0 elements apart: at
// 48 kHz 600 elements = 1200 dwords = 48 mod 64 banks, i.e. FOUR distinct banks for 64 lanes (16-way conflicts on every LDS
// access of that step: 14.5 of a frame's 45 us).  With ps = 6 the stride becomes 609 elements = 2 mod 64 banks: conflict-free.
// The late passes (sub-blocks of fewer than 64 elements: strides of 4, 40, ... elements) gain the same way.
RFX_HD int gen_ipad(int i, int ps) { return ps ? i + (i >> ps) : i; }
RFX_HD int gen_ibuf_elems(int nc, int ps) { return ps ? nc + (nc >> ps) + 1 : nc; }

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
  int nthr;     // threads per workgroup the kernels are launched with (gen_pick_threads)
  int pad_shift;  // LDS padding of the in-place buffer (gen_ipad), 0 = none
  // synthesis-frame buffer [B*T][fpitch]: window sample j of a frame sits at fshift + j.  Plain (fpitch = win, fshift = 0) unless
  // gen_frame_layout found a shift that lets the fold read 16 bytes at a time (hop a multiple of four but the window length or its
  // offset in the padded signal not: 22.05 kHz, 2205 / 1103): then the row is padded with >= 3 zeros on either side
  int fpitch, fshift;
};
// Frame-buffer layout for the overlap-add (gen_fold4_kernel): output sample p reads frame t at j = p + n_fft / 2 - left - hop t.  Four
// consecutive p (p % 4 == 0) read 16 aligned bytes from every frame iff hop % 4 == 0 and (n_fft / 2 - left + fshift) % 4 == 0 and
// fpitch % 4 == 0; where the window's ends fall inside such a group the neighbours must read zeros: three floats of padding
RFX_HD void gen_frame_layout(GenGeom& g) {
  const int off = g.n_fft / 2 - g.left;
  g.fshift = 0;
  g.fpitch = g.win;
  if (g.hop % 4 != 0 || (g.win % 4 == 0 && off % 4 == 0)) return;
  int sh = (4 - off % 4) % 4;
  while (sh < 3) sh += 4;
  g.fshift = sh;
  g.fpitch = (sh + g.win + 3 + 3) / 4 * 4;
}

// radices the butterfly below implements
RFX_HD bool gen_factor(int n, int* radix, int* nstages) {
  // largest digits first: composite radices (computed in registers, gen_dft_ct) mean fewer passes through LDS
  const int cand[15] = {16, 15, 14, 12, 10, 9, 8, 6, 4, 2, 3, 5, 7, 11, 13};
  int ns = 0;
  for (int c = 0; c < 15; ++c)
    while (n % cand[c] == 0) {
      if (ns == kGenMaxStages) return false;
      radix[ns++] = cand[c];
      n /= cand[c];
    }
  *nstages = ns;
  return n == 1;
}

// Threads per workgroup.  A pass of radix R has nc / R butterflies; dealt to `nthr` threads it takes ceil(nc / R / nthr)
// rounds, and with the large composite radices the butterfly count comes close to the thread count: 9600 points / radix 16 =
// 600 butterflies keep 512 threads busy for 59 % of two rounds, 320 threads for 94 %.  Picks the multiple of 64 in
// [192, max_threads] that wastes the fewest thread-rounds over all passes (ties: more threads).
RFX_HD int gen_pick_threads(const GenGeom& g, int max_threads) {
  int best = max_threads;
  long long best_cost = -1;
  for (int nthr = max_threads; nthr >= 192; nthr -= 64) {
    long long cost = 0;
    for (int s = 0; s < g.nstages; ++s) {
      const int nbf = g.nc / g.radix[s];
      cost += (long long)((nbf + nthr - 1) / nthr) * nthr * g.radix[s];  // element-slots issued by the pass
    }
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nthr; }
  }
  return best;
}

// LDS padding shift of the in-place buffer: the candidate (0 = none, 4..7) with the fewest bank conflicts in the split /
// projection step (64 consecutive bins k at positions rev[k]: a stride of nc / R0 elements, 8 bytes each, over 64 four-byte
// banks) and, second, in the passes whose sub-blocks are shorter than a wave; `max_elems` bounds the padded buffer (LDS left
// when as many workgroups share a CU as without padding).
RFX_HD int gen_conflict_degree(const int (&pos)[64], int ps) {
  int hits[64];
  for (int b = 0; b < 64; ++b) hits[b] = 0;
  int worst = 0;
  for (int l = 0; l < 64; ++l) {
    const int bank = (2 * gen_ipad(pos[l], ps)) & 63;  // first dword of the 8-byte element
    if (++hits[bank] > worst) worst = hits[bank];
  }
  return worst;  // 2 = ideal (64 lanes x 8 bytes over 64 banks take two cycles anyway)
}
RFX_HD int gen_pick_pad(const GenGeom& g, int max_elems) {
  int best = 0;
  long long best_score = -1;
  const int cand[5] = {0, 6, 5, 7, 4};
  for (int c = 0; c < 5; ++c) {
    const int ps = cand[c];
    if (gen_ibuf_elems(g.nc, ps) > max_elems) continue;
    int pos[64];
    // the projection step: lane l handles bin k0 + l at rev[k0 + l] = (k0 + l) % R0 * (nc / R0) + ...; four accesses per pair
    for (int l = 0; l < 64; ++l) pos[l] = (l % g.radix[0]) * (g.nc / g.radix[0]) + (l / g.radix[0]) * (g.nstages > 1 ? g.nc / g.radix[0] / g.radix[1] : 0);
    long long score = 4LL * gen_conflict_degree(pos, ps);
    int L = g.nc;
    for (int s2 = 0; s2 < g.nstages; ++s2) {  // passes: lane l = butterfly j0 + l reads blk * L + i + q * m (q fixed per access)
      const int R = g.radix[s2], m = L / R;
      for (int l = 0; l < 64; ++l) pos[l] = (l / m) * L + (l % m);
      score += 2LL * gen_conflict_degree(pos, ps);  // a load and a store per element, R of them per butterfly either way
      L = m;
    }
    if (best_score < 0 || score < best_score) { best_score = score; best = ps; }
  }
  return best;
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

// ---- composite radices: R = R1 * R2 <= 16 as one digit of the mixed-radix passes, computed in registers as R2 DFTs of length
// R1, constant twiddles exp(-2 pi i n2 k1 / R), R1 DFTs of length R2 (n = R2 n1 + n2 in, k = k1 + R1 k2 out).  9600 points
// (48 kHz) take 4 passes [16, 15, 10, 4] instead of 7 [4, 4, 4, 2, 3, 5, 5]: 3 fewer trips through LDS, 3 fewer barriers and
// 3 fewer table twiddles per element and direction.
template <int R> struct GenRootTab;
template <> struct GenRootTab<6> {
  static constexpr float c[6] = {1.00000000000000000000f, 0.50000000000000011102f, -0.49999999999999977796f, -1.00000000000000000000f, -0.50000000000000044409f, 0.50000000000000011102f};
  static constexpr float s[6] = {0.00000000000000000000f, 0.86602540378443859659f, 0.86602540378443870761f, 0.00000000000000012246f, -0.86602540378443837454f, -0.86602540378443859659f};
};
template <> struct GenRootTab<8> {
  static constexpr float c[8] = {1.00000000000000000000f, 0.70710678118654757274f, 0.00000000000000006123f, -0.70710678118654746172f, -1.00000000000000000000f, -0.70710678118654768376f, -0.00000000000000018370f, 0.70710678118654735069f};
  static constexpr float s[8] = {0.00000000000000000000f, 0.70710678118654746172f, 1.00000000000000000000f, 0.70710678118654757274f, 0.00000000000000012246f, -0.70710678118654746172f, -1.00000000000000000000f, -0.70710678118654768376f};
};
template <> struct GenRootTab<9> {
  static constexpr float c[9] = {1.00000000000000000000f, 0.76604444311897801345f, 0.17364817766693041445f, -0.49999999999999977796f, -0.93969262078590831688f, -0.93969262078590842791f, -0.50000000000000044409f, 0.17364817766692997036f, 0.76604444311897779141f};
  static constexpr float s[9] = {0.00000000000000000000f, 0.64278760968653925190f, 0.98480775301220802032f, 0.86602540378443870761f, 0.34202014332566887944f, -0.34202014332566865740f, -0.86602540378443837454f, -0.98480775301220813134f, -0.64278760968653958496f};
};
template <> struct GenRootTab<10> {
  static constexpr float c[10] = {1.00000000000000000000f, 0.80901699437494745126f, 0.30901699437494745126f, -0.30901699437494734024f, -0.80901699437494734024f, -1.00000000000000000000f, -0.80901699437494756229f, -0.30901699437494756229f, 0.30901699437494722922f, 0.80901699437494734024f};
  static constexpr float s[10] = {0.00000000000000000000f, 0.58778525229247313710f, 0.95105651629515353118f, 0.95105651629515364220f, 0.58778525229247324813f, 0.00000000000000012246f, -0.58778525229247302608f, -0.95105651629515353118f, -0.95105651629515364220f, -0.58778525229247335915f};
};
template <> struct GenRootTab<12> {
  static constexpr float c[12] = {1.00000000000000000000f, 0.86602540378443870761f, 0.50000000000000011102f, 0.00000000000000006123f, -0.49999999999999977796f, -0.86602540378443870761f, -1.00000000000000000000f, -0.86602540378443881863f, -0.50000000000000044409f, -0.00000000000000018370f, 0.50000000000000011102f, 0.86602540378443837454f};
  static constexpr float s[12] = {0.00000000000000000000f, 0.49999999999999994449f, 0.86602540378443859659f, 1.00000000000000000000f, 0.86602540378443870761f, 0.49999999999999994449f, 0.00000000000000012246f, -0.49999999999999972244f, -0.86602540378443837454f, -1.00000000000000000000f, -0.86602540378443859659f, -0.50000000000000044409f};
};
template <> struct GenRootTab<14> {
  static constexpr float c[14] = {1.00000000000000000000f, 0.90096886790241914600f, 0.62348980185873359439f, 0.22252093395631444839f, -0.22252093395631433737f, -0.62348980185873348336f, -0.90096886790241903498f, -1.00000000000000000000f, -0.90096886790241914600f, -0.62348980185873370541f, -0.22252093395631458717f, 0.22252093395631333816f, 0.62348980185873337234f, 0.90096886790241936804f};
  static constexpr float s[14] = {0.00000000000000000000f, 0.43388373911755812040f, 0.78183148246802980363f, 0.97492791218182361934f, 0.97492791218182361934f, 0.78183148246802991466f, 0.43388373911755823142f, 0.00000000000000012246f, -0.43388373911755800938f, -0.78183148246802969261f, -0.97492791218182361934f, -0.97492791218182384139f, -0.78183148246802991466f, -0.43388373911755750978f};
};
template <> struct GenRootTab<15> {
  static constexpr float c[15] = {1.00000000000000000000f, 0.91354545764260086660f, 0.66913060635885823757f, 0.30901699437494745126f, -0.10452846326765333207f, -0.49999999999999977796f, -0.80901699437494734024f, -0.97814760073380568883f, -0.97814760073380568883f, -0.80901699437494756229f, -0.50000000000000044409f, -0.10452846326765423413f, 0.30901699437494722922f, 0.66913060635885845961f, 0.91354545764260097762f};
  static constexpr float s[15] = {0.00000000000000000000f, 0.40673664307580015276f, 0.74314482547739413310f, 0.95105651629515353118f, 0.99452189536827340088f, 0.86602540378443870761f, 0.58778525229247324813f, 0.20791169081775931482f, -0.20791169081775906502f, -0.58778525229247302608f, -0.86602540378443837454f, -0.99452189536827328986f, -0.95105651629515364220f, -0.74314482547739402207f, -0.40673664307580015276f};
};
template <> struct GenRootTab<16> {
  static constexpr float c[16] = {1.00000000000000000000f, 0.92387953251128673848f, 0.70710678118654757274f, 0.38268343236508983729f, 0.00000000000000006123f, -0.38268343236508972627f, -0.70710678118654746172f, -0.92387953251128673848f, -1.00000000000000000000f, -0.92387953251128684951f, -0.70710678118654768376f, -0.38268343236509033689f, -0.00000000000000018370f, 0.38268343236509000382f, 0.70710678118654735069f, 0.92387953251128651644f};
  static constexpr float s[16] = {0.00000000000000000000f, 0.38268343236508978178f, 0.70710678118654746172f, 0.92387953251128673848f, 1.00000000000000000000f, 0.92387953251128673848f, 0.70710678118654757274f, 0.38268343236508989280f, 0.00000000000000012246f, -0.38268343236508967076f, -0.70710678118654746172f, -0.92387953251128651644f, -1.00000000000000000000f, -0.92387953251128662746f, -0.70710678118654768376f, -0.38268343236509039240f};
};
template <> struct GenRootTab<20> {
  static constexpr float c[20] = {1.00000000000000000000f, 0.95105651629515353118f, 0.80901699437494745126f, 0.58778525229247313710f, 0.30901699437494745126f, 0.00000000000000006123f, -0.30901699437494734024f, -0.58778525229247302608f, -0.80901699437494734024f, -0.95105651629515353118f, -1.00000000000000000000f, -0.95105651629515375323f, -0.80901699437494756229f, -0.58778525229247324813f, -0.30901699437494756229f, -0.00000000000000018370f, 0.30901699437494722922f, 0.58778525229247291506f, 0.80901699437494734024f, 0.95105651629515353118f};
  static constexpr float s[20] = {0.00000000000000000000f, 0.30901699437494739575f, 0.58778525229247313710f, 0.80901699437494745126f, 0.95105651629515353118f, 1.00000000000000000000f, 0.95105651629515364220f, 0.80901699437494745126f, 0.58778525229247324813f, 0.30901699437494750677f, 0.00000000000000012246f, -0.30901699437494689615f, -0.58778525229247302608f, -0.80901699437494734024f, -0.95105651629515353118f, -1.00000000000000000000f, -0.95105651629515364220f, -0.80901699437494756229f, -0.58778525229247335915f, -0.30901699437494761780f};
};
template <> struct GenRootTab<24> {
  static constexpr float c[24] = {1.00000000000000000000f, 0.96592582628906831221f, 0.86602540378443870761f, 0.70710678118654757274f, 0.50000000000000011102f, 0.25881904510252073948f, 0.00000000000000006123f, -0.25881904510252062845f, -0.49999999999999977796f, -0.70710678118654746172f, -0.86602540378443870761f, -0.96592582628906820119f, -1.00000000000000000000f, -0.96592582628906831221f, -0.86602540378443881863f, -0.70710678118654790580f, -0.50000000000000044409f, -0.25881904510252062845f, -0.00000000000000018370f, 0.25881904510252029539f, 0.50000000000000011102f, 0.70710678118654735069f, 0.86602540378443837454f, 0.96592582628906809017f};
  static constexpr float s[24] = {0.00000000000000000000f, 0.25881904510252073948f, 0.49999999999999994449f, 0.70710678118654746172f, 0.86602540378443859659f, 0.96592582628906831221f, 1.00000000000000000000f, 0.96592582628906831221f, 0.86602540378443870761f, 0.70710678118654757274f, 0.49999999999999994449f, 0.25881904510252101703f, 0.00000000000000012246f, -0.25881904510252079499f, -0.49999999999999972244f, -0.70710678118654712865f, -0.86602540378443837454f, -0.96592582628906831221f, -1.00000000000000000000f, -0.96592582628906842324f, -0.86602540378443859659f, -0.70710678118654768376f, -0.50000000000000044409f, -0.25881904510252157214f};
};
template <int R, bool INV>
RFX_HD cf gen_const_twiddle(cf x, int e) {  // x * exp(-/+ 2 pi i e / R); e is a compile-time constant wherever this is called
  e %= R;
  if (e == 0) return x;
  if (2 * e == R) return cf{-x.re, -x.im};
  if (4 * e == R) return INV ? cf{-x.im, x.re} : cf{x.im, -x.re};        // forward: * (-i)
  if (4 * e == 3 * R) return INV ? cf{x.im, -x.re} : cf{-x.im, x.re};    // forward: * (+i)
  const cf w{GenRootTab<R>::c[e], -GenRootTab<R>::s[e]};
  return INV ? cmulc(x, w) : cmul(x, w);
}
template <int R, bool INV>
RFX_HD void gen_dft(const cf (&v)[R], cf (&y)[R], const cf (&root)[R]);
template <int R1, int R2, bool INV>
RFX_HD void gen_dft_ct(const cf (&x)[R1 * R2], cf (&y)[R1 * R2]) {
  constexpr int R = R1 * R2;
  cf t[R2][R1];
  const cf none1[R1] = {};
  const cf none2[R2] = {};
#pragma unroll
  for (int n2 = 0; n2 < R2; ++n2) {
    cf a[R1], b[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) a[n1] = x[R2 * n1 + n2];
    gen_dft<R1, INV>(a, b, none1);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) t[n2][k1] = gen_const_twiddle<R, INV>(b[k1], n2 * k1);
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    cf a[R2], b[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) a[n2] = t[n2][k1];
    gen_dft<R2, INV>(a, b, none2);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) y[k1 + R1 * k2] = b[k2];
  }
}

// the R-point DFT of v into y (INV: exp(+i ...) kernels); root = exp(-2 pi i t / R), used by the O(R^2) radices only
template <int R, bool INV>
RFX_HD void gen_dft(const cf (&v)[R], cf (&y)[R], const cf (&root)[R]) {
  if constexpr (R == 2) {
    y[0] = cf{v[0].re + v[1].re, v[0].im + v[1].im};
    y[1] = cf{v[0].re - v[1].re, v[0].im - v[1].im};
  } else if constexpr (R == 4) {
    const cf a{v[0].re + v[2].re, v[0].im + v[2].im}, b{v[0].re - v[2].re, v[0].im - v[2].im};
    const cf c{v[1].re + v[3].re, v[1].im + v[3].im}, d{v[1].re - v[3].re, v[1].im - v[3].im};
    // forward: y1 = b - i d, y3 = b + i d ; inverse swaps them
    const cf md = INV ? cf{-d.im, d.re} : cf{d.im, -d.re};  // (-i d) forward, (+i d) inverse
    y[0] = cf{a.re + c.re, a.im + c.im};
    y[2] = cf{a.re - c.re, a.im - c.im};
    y[1] = cf{b.re + md.re, b.im + md.im};
    y[3] = cf{b.re - md.re, b.im - md.im};
  } else if constexpr (R == 3) {
    cf a = v[0], b = v[1], c = v[2];
    dft3<INV>(a, b, c);
    y[0] = a;
    y[1] = b;
    y[2] = c;
  } else if constexpr (R == 5) {
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
  } else if constexpr (R == 7) {
    cf x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3], x4 = v[4], x5 = v[5], x6 = v[6];
    dft7<INV>(x0, x1, x2, x3, x4, x5, x6);
    y[0] = x0; y[1] = x1; y[2] = x2; y[3] = x3; y[4] = x4; y[5] = x5; y[6] = x6;
  } else if constexpr (R == 6) {
    gen_dft_ct<3, 2, INV>(v, y);
  } else if constexpr (R == 8) {
    gen_dft_ct<4, 2, INV>(v, y);
  } else if constexpr (R == 9) {
    gen_dft_ct<3, 3, INV>(v, y);
  } else if constexpr (R == 10) {
    gen_dft_ct<5, 2, INV>(v, y);
  } else if constexpr (R == 12) {
    gen_dft_ct<4, 3, INV>(v, y);
  } else if constexpr (R == 14) {
    gen_dft_ct<7, 2, INV>(v, y);
  } else if constexpr (R == 15) {
    gen_dft_ct<5, 3, INV>(v, y);
  } else if constexpr (R == 16) {
    gen_dft_ct<4, 4, INV>(v, y);
  } else if constexpr (R == 20) {  // 20, 21, 24: digits of the row transforms of rfx_fam_core.h
    gen_dft_ct<4, 5, INV>(v, y);
  } else if constexpr (R == 24) {
    gen_dft_ct<8, 3, INV>(v, y);
  } else if constexpr (R == 21) {
    cf x[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) x[i] = v[i];
    dft21<INV>(x);
#pragma unroll
    for (int i = 0; i < 21; ++i) y[i] = x[i];
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
// One butterfly in three steps - load, compute, store - so that a thread can run the loads of SEVERAL butterflies before
// the first store: the passes work in place, every store may alias every later load as far as the compiler can tell, and a
// loop of load-compute-store butterflies is executed strictly one LDS round trip after the other (measured: the generic
// engine spent more than half of its cycles waiting on such chains).
template <int R>
struct GenBfly {
  cf v[R];
  int base, i;
};
template <int R>
RFX_HD void gen_ip_load(GenBfly<R>& b, const cf* buf, int j, int m, float inv_m, int L, int ps) {
  const int blk = gen_div(j, m, inv_m);
  b.i = j - blk * m;
  b.base = blk * L + b.i;
#pragma unroll
  for (int q = 0; q < R; ++q) b.v[q] = buf[gen_ipad(b.base + q * m, ps)];
}
// Twiddles of a pass: W_L^{i p}, p = 1..R-1, for the butterfly at position i of its sub-block.  `tw` (nullable) is the pass's
// EXACT table, [m][R-1] values rounded once from double precision (built at plan creation, L2-resident: 77 KB at 48 kHz for
// all passes); without it they come from the two-level table (two LDS reads and a complex product each: more than half of
// the instructions of a radix-16 butterfly).  The last pass (m == 1) has no twiddles at all.
template <int R, bool INV>
RFX_HD void gen_ip_compute(GenBfly<R>& b, int m, int tstep, const cf* lo, const cf* hi, const cf (&root)[R], const cf* tw) {
  cf v[R], y[R], w[R];
  if (m > 1) {
#pragma unroll
    for (int q = 1; q < R; ++q) {
#if defined(__HIP_DEVICE_COMPILE__)
      w[q] = tw[b.i * (R - 1) + q - 1];  // the kernels always pass the exact tables
#else
      w[q] = tw ? tw[b.i * (R - 1) + q - 1] : gen_tw(lo, hi, q * b.i * tstep);  // q * i * tstep < nc
#endif
    }
  }
  v[0] = b.v[0];
#pragma unroll
  for (int q = 1; q < R; ++q) v[q] = (INV && m > 1) ? cmulc(b.v[q], w[q]) : b.v[q];
  gen_dft<R, INV>(v, y, root);
  b.v[0] = y[0];
#pragma unroll
  for (int p = 1; p < R; ++p) b.v[p] = (INV || m == 1) ? y[p] : cmul(y[p], w[p]);
}
template <int R>
RFX_HD void gen_ip_store(const GenBfly<R>& b, cf* buf, int m, int ps) {
#pragma unroll
  for (int p = 0; p < R; ++p) buf[gen_ipad(b.base + p * m, ps)] = b.v[p];
}
template <int R, bool INV>
RFX_HD void gen_ip_butterfly(cf* buf, int j, int m, float inv_m, int L, int tstep, const cf* lo, const cf* hi, const cf (&root)[R], int ps,
                             const cf* tw) {
  GenBfly<R> b;
  gen_ip_load<R>(b, buf, j, m, inv_m, L, ps);
  gen_ip_compute<R, INV>(b, m, tstep, lo, hi, root, tw);
  gen_ip_store<R>(b, buf, m, ps);
}
// butterflies a thread keeps in flight per trip: about sixteen complex values
template <int R>
struct GenBatch {
  static constexpr int value = R <= 2 ? 8 : R <= 4 ? 4 : R <= 8 ? 2 : 1;
};
template <int R, bool INV>
RFX_HD void gen_ip_stage_r(cf* buf, int nc, int L, const cf* lo, const cf* hi, int tid, int nthr, int ps, const cf* tw) {
  const int m = L / R, tstep = nc / L, nbf = nc / R;
  const float inv_m = 1.0f / (float)m;
  cf root[R];
#pragma unroll
  for (int t = 0; t < R; ++t) root[t] = gen_tw(lo, hi, t * (nc / R));
  constexpr int U = GenBatch<R>::value;
  int j = tid;
  for (; j + (U - 1) * nthr < nbf; j += U * nthr) {  // full batches: all loads, then all arithmetic, then all stores
    GenBfly<R> b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) gen_ip_load<R>(b[u], buf, j + u * nthr, m, inv_m, L, ps);
#pragma unroll
    for (int u = 0; u < U; ++u) gen_ip_compute<R, INV>(b[u], m, tstep, lo, hi, root, tw);
#pragma unroll
    for (int u = 0; u < U; ++u) gen_ip_store<R>(b[u], buf, m, ps);
  }
  for (; j < nbf; j += nthr) gen_ip_butterfly<R, INV>(buf, j, m, inv_m, L, tstep, lo, hi, root, ps, tw);
}
template <bool INV, int MAXR = 13>
RFX_HD void gen_ip_stage(cf* buf, int nc, int L, int R, const cf* lo, const cf* hi, int tid, int nthr, int ps = 0, const cf* tw = nullptr) {
  switch (R) {
    case 2: gen_ip_stage_r<2, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 3: gen_ip_stage_r<3, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 4: gen_ip_stage_r<4, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 5: gen_ip_stage_r<5, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 6: gen_ip_stage_r<6, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 8: gen_ip_stage_r<8, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 9: gen_ip_stage_r<9, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 10: gen_ip_stage_r<10, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 12: gen_ip_stage_r<12, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 15: gen_ip_stage_r<15, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 16: gen_ip_stage_r<16, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 7: if (MAXR >= 7) gen_ip_stage_r<7, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 14: if (MAXR >= 7) gen_ip_stage_r<14, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    case 11: if (MAXR >= 11) gen_ip_stage_r<11, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
    default: if (MAXR >= 13) gen_ip_stage_r<13, INV>(buf, nc, L, lo, hi, tid, nthr, ps, tw); break;
  }
}
// exact per-pass twiddle tables: pass s (blocks of length L_s, sub-blocks m_s = L_s / R_s) owns m_s * (R_s - 1) entries
// W_{L_s}^{i p} at [i * (R_s - 1) + p - 1]; the passes' tables follow each other in pass order
RFX_HD int gen_tw_table_offset(const GenGeom& g, int pass) {
  int off = 0, L = g.nc;
  for (int s2 = 0; s2 < pass; ++s2) {
    const int m = L / g.radix[s2];
    off += m * (g.radix[s2] - 1);
    L = m;
  }
  return off;
}
RFX_HD int gen_tw_table_elems(const GenGeom& g) { return gen_tw_table_offset(g, g.nstages); }

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
    case 6: gen_stage_r<6, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 8: gen_stage_r<8, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 9: gen_stage_r<9, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 10: gen_stage_r<10, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 12: gen_stage_r<12, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 15: gen_stage_r<15, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 16: gen_stage_r<16, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 7: if (MAXR >= 7) gen_stage_r<7, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 14: if (MAXR >= 7) gen_stage_r<14, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    case 11: if (MAXR >= 11) gen_stage_r<11, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
    default: if (MAXR >= 13) gen_stage_r<13, INV>(in, out, nc, Ns, lo, hi, tid, nthr); break;
  }
}
// the class a radix list needs
RFX_HD int gen_radix_class(const int* radix, int nstages) {
  int cls = 5;  // 5: digits made of 2, 3, 5 only; 7: a digit 7 or 14; 13: a digit 11 or 13
  for (int i = 0; i < nstages; ++i) {
    const int r = radix[i];
    const int c = (r == 11 || r == 13) ? 13 : (r == 7 || r == 14) ? 7 : 5;
    cls = c > cls ? c : cls;
  }
  return cls;
}

// ---- real <-> packed-complex split.  Z: the nc-point complex spectrum (LDS), lo2/hi2: two-level table of
// exp(-2 pi i k / n_fft).  Returns bin k (0 <= k <= n_fft/2) of the real FFT.
// `rev` (nullable): position of element k inside Z (digit-reversed after the in-place passes)
// zk = Z[k] (Z[0] for k == nc), zc = Z[nc - k] (Z[0] for k == 0)
RFX_HD cf gen_split_forward_vals(const GenGeom& g, cf zk, cf zc, const cf* lo2, const cf* hi2, int k) {
  if (!g.even) return zk;
  const cf s{zk.re + zc.re, zk.im - zc.im}, d{zk.re - zc.re, zk.im + zc.im};  // Z[k] +- conj Z[nc-k]
  const cf p = cmul(gen_tw(lo2, hi2, k), d);
  return cf{0.5f * (s.re + p.im), 0.5f * (s.im - p.re)};  // (s - i p) / 2
}
RFX_HD cf gen_split_forward(const GenGeom& g, const cf* Z, const cf* lo2, const cf* hi2, int k, const int* rev = nullptr) {
  auto at = [&](int i) { return Z[rev ? rev[i] : gen_pad(i)]; };
  if (!g.even) return at(k);
  return gen_split_forward_vals(g, at(k == g.nc ? 0 : k), at(k == 0 ? 0 : g.nc - k), lo2, hi2, k);
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

// Griffin-Lim per-bin update (torchaudio functional.griffinlim):
//     a = rebuilt - m * tprev ; angles = a / (|a| + 1e-16) ; next = S * angles
// `rebuilt - m * tprev` is the spectrum of x_k - m x_{k-1} (the STFT is linear, see rfx_gl.hip): the kernels analyse that
// signal and `a` arrives here directly.  The normalisation is gl_project of rfx_core.h (one v_rsq_f32 on the device, exact
// sqrt / divide on the host).
//
// In-place update of the packed spectrum between the forward and the inverse passes, one call per PAIR of elements
// (k, nc - k), 0 <= k <= nc / 2, of the nc-point complex spectrum of the packed frame (even n_fft); for odd n_fft one call per
// one-sided bin 0 <= k <= (n_fft - 1) / 2, which owns elements k and n_fft - k.  `Z` holds the forward transform (element i at
// `at(i)`: digit-reversed after the in-place passes), S the frame's magnitudes by bin.  The pair is self-contained: the two
// one-sided bins it yields (k and nc - k; for k = 0: bins 0 and nc) are projected and turned back into the two elements the
// inverse passes need, which are written where they were read.
// (three steps - load, compute, store - like the butterflies: a thread loads several pairs before it stores the first)
struct GenPair {
  cf zk, zc;      // elements k and nc - k (odd n_fft: zk only)
  float sk, sc;   // magnitudes of bins k and nc - k (k == 0: bins 0 and nc)
  int k;
};
template <class AT>
RFX_HD void gen_pair_load(GenPair& p, const GenGeom& g, const cf* Z, AT at, const float* S, int k) {
  p.k = k;
  p.zk = Z[at(k)];
  p.sk = S[k];
  if (g.even) {
    const int kc = g.nc - k;
    p.zc = (k == 0 || k == kc) ? p.zk : Z[at(kc)];
    p.sc = S[kc];  // k == 0: bin nc
  } else {
    p.zc = p.zk;
    p.sc = 0.f;
  }
}
// zk, zc become the two elements the inverse passes need.  The generic pair shares ONE table twiddle w = exp(-2 pi i k / n_fft)
// between both bins and both directions: bin nc - k has the twiddle -conj(w), so with s = Z[k] + conj Z[nc-k],
// d = Z[k] - conj Z[nc-k], p = w d
//     X[k] = (s - i p) / 2,   X[nc-k] = conj((s + i p) / 2)
// and back, with S = X'[k] + conj X'[nc-k], D = X'[k] - conj X'[nc-k], P = D conj(w):
//     Z'[k] = (S + i P) / 2,  Z'[nc-k] = conj((S - i P) / 2)
// (two complex multiplications and one twiddle per pair; the four gen_split_*_vals calls they replace fetched four).
RFX_HD void gen_pair_compute(GenPair& p, const GenGeom& g, const cf* lo2, const cf* hi2, float eps2 = 1e-32f) {
  const int k = p.k;
  if (!g.even) {
    const cf X = gl_project(p.zk, p.sk, eps2);
    p.zk = k == 0 ? cf{X.re, 0.f} : X;  // the c2r transform ignores the imaginary part of bin 0
    p.zc = cf{X.re, -X.im};
    return;
  }
  if (k == 0) {
    const cf X0 = gl_project(gen_split_forward_vals(g, p.zk, p.zk, lo2, hi2, 0), p.sk, eps2);
    const cf Xn = gl_project(gen_split_forward_vals(g, p.zk, p.zk, lo2, hi2, g.nc), p.sc, eps2);
    p.zk = gen_split_inverse_vals(g, X0, Xn, lo2, hi2, 0);
    return;
  }
  const cf w = gen_tw(lo2, hi2, k);
  const cf s{p.zk.re + p.zc.re, p.zk.im - p.zc.im}, d{p.zk.re - p.zc.re, p.zk.im + p.zc.im};
  const cf q = cmul(w, d);
  // X[k] = (s - i q) / 2 ; X[nc-k] = conj((s + i q) / 2)
  const cf Xk = gl_project(cf{0.5f * (s.re + q.im), 0.5f * (s.im - q.re)}, p.sk, eps2);
  const cf Xc = gl_project(cf{0.5f * (s.re - q.im), -0.5f * (s.im + q.re)}, p.sc, eps2);
  const cf S{Xk.re + Xc.re, Xk.im - Xc.im}, D{Xk.re - Xc.re, Xk.im + Xc.im};
  const cf P = cmulc(D, w);
  // Z'[k] = (S + i P) / 2 ; Z'[nc-k] = conj((S - i P) / 2)
  p.zk = cf{0.5f * (S.re - P.im), 0.5f * (S.im + P.re)};
  p.zc = cf{0.5f * (S.re + P.im), -0.5f * (S.im - P.re)};
}
template <class AT>
RFX_HD void gen_pair_store(const GenPair& p, const GenGeom& g, cf* Z, AT at) {
  Z[at(p.k)] = p.zk;
  if (g.even) {
    if (p.k != 0 && p.k != g.nc - p.k) Z[at(g.nc - p.k)] = p.zc;
  } else if (p.k != 0) {
    Z[at(g.n_fft - p.k)] = p.zc;
  }
}
template <class AT>
RFX_HD void gen_pair_project(const GenGeom& g, cf* Z, AT at, const float* S, const cf* lo2, const cf* hi2, int k, float eps2 = 1e-32f) {
  GenPair p;
  gen_pair_load(p, g, Z, at, S, k);
  gen_pair_compute(p, g, lo2, hi2, eps2);
  gen_pair_store(p, g, Z, at);
}
// number of gen_pair_project calls per frame
RFX_HD int gen_pair_count(const GenGeom& g) { return g.even ? g.nc / 2 + 1 : (g.n_fft - 1) / 2 + 1; }

}  // namespace rfx
