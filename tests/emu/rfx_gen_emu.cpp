// Host-side emulator of the GENERIC frame transform (csrc/rfx_gen_core.h).  TEST INFRASTRUCTURE ONLY, built with g++ by
// tests/test_gen_core.py: it runs the same Stockham pass / real-split functions the gfx950 kernels inline, looping over the
// logical threads of a workgroup pass by pass (a loop boundary stands where the kernel has a barrier).
#include <cmath>
#include <vector>
#include "../../riffusion-hobby_amd/csrc/rfx_gen_core.h"

using namespace rfx;

namespace {
struct Tables {
  std::vector<cf> lo, hi, lo2, hi2;
};
bool make_geom(int n_fft, int win, int hop, GenGeom& g, Tables& t) {
  g = GenGeom{};
  g.n_fft = n_fft; g.win = win; g.hop = hop; g.n_stft = n_fft / 2 + 1;
  g.even = n_fft % 2 == 0;
  g.nc = g.even ? n_fft / 2 : n_fft;
  g.left = (n_fft - win) / 2;
  g.fs = (g.n_stft + 63) / 64 * 64;
  g.nhi = g.nc / kGenTwLo + 1;
  g.nhi2 = g.nc / kGenTwLo + 2;
  if (!gen_factor(g.nc, g.radix, &g.nstages)) return false;
  const double PI2 = 6.283185307179586476925286766559;
  auto root = [&](long long num, long long den) {
    const double a = -PI2 * (double)(num % den) / (double)den;
    return cf{(float)cos(a), (float)sin(a)};
  };
  t.lo.resize(kGenTwLo); t.lo2.resize(kGenTwLo); t.hi.resize(g.nhi); t.hi2.resize(g.nhi2);
  for (int i = 0; i < kGenTwLo; ++i) { t.lo[i] = root(i, g.nc); t.lo2[i] = root(i, g.n_fft); }
  for (int i = 0; i < g.nhi; ++i) t.hi[i] = root((long long)i * kGenTwLo, g.nc);
  for (int i = 0; i < g.nhi2; ++i) t.hi2[i] = root((long long)i * kGenTwLo, g.n_fft);
  return true;
}
template <bool INV>
cf* run_fft(const GenGeom& g, const Tables& t, cf* a, cf* b, int nthr) {
  cf *in = a, *out = b;
  int Ns = 1;
  for (int s = 0; s < g.nstages; ++s) {
    for (int tid = 0; tid < nthr; ++tid) gen_stage<INV>(in, out, g.nc, Ns, g.radix[s], t.lo.data(), t.hi.data(), tid, nthr);
    Ns *= g.radix[s];
    cf* x = in; in = out; out = x;
  }
  return in;
}
// in-place passes: forward DIF over blocks L = nc, nc/R0, ...; inverse DIT over the same (L, m) pairs in reverse order
// exact per-pass twiddles, as plan creation builds them (double precision, rounded once)
std::vector<cf> make_tw_table(const GenGeom& g) {
  std::vector<cf> tab(gen_tw_table_elems(g) + 1);
  const double PI2 = 6.283185307179586476925286766559;
  int L = g.nc;
  for (int s = 0; s < g.nstages; ++s) {
    const int R = g.radix[s], m = L / R, off = gen_tw_table_offset(g, s);
    for (int i = 0; i < m; ++i)
      for (int p = 1; p < R; ++p) {
        const double a = -PI2 * (double)(((long long)i * p) % L) / (double)L;
        tab[off + i * (R - 1) + p - 1] = cf{(float)cos(a), (float)sin(a)};
      }
    L = m;
  }
  return tab;
}
template <bool INV>
void run_fft_inplace(const GenGeom& g, const Tables& t, cf* buf, int nthr, int ps = 0, const cf* tw = nullptr) {
  int Ls[kGenMaxStages];
  int L = g.nc;
  for (int s = 0; s < g.nstages; ++s) { Ls[s] = L; L /= g.radix[s]; }
  for (int i = 0; i < g.nstages; ++i) {
    const int s = INV ? g.nstages - 1 - i : i;
    for (int tid = 0; tid < nthr; ++tid)
      gen_ip_stage<INV>(buf, g.nc, Ls[s], g.radix[s], t.lo.data(), t.hi.data(), tid, nthr, ps, tw ? tw + gen_tw_table_offset(g, s) : nullptr);
  }
}
}  // namespace

extern "C" {

int emu_gen_rfft_inplace(int n_fft, const float* frame, float* out, int nthr) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return -1;
  std::vector<cf> a(g.nc);
  std::vector<int> rev(g.nc);
  for (int k = 0; k < g.nc; ++k) rev[k] = gen_digit_reverse(g, k);
  for (int n = 0; n < g.nc; ++n) a[n] = g.even ? cf{frame[2 * n], frame[2 * n + 1]} : cf{frame[n], 0.f};
  run_fft_inplace<false>(g, t, a.data(), nthr);
  for (int k = 0; k < g.n_stft; ++k) {
    const cf X = gen_split_forward(g, a.data(), t.lo2.data(), t.hi2.data(), k, rev.data());
    out[2 * k] = X.re; out[2 * k + 1] = X.im;
  }
  return 0;
}

int emu_gen_irfft_inplace(int n_fft, const float* spec, float* out, int nthr) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return -1;
  std::vector<cf> a(g.nc);
  auto X = [&](int k) { return cf{spec[2 * k], spec[2 * k + 1]}; };
  for (int k = 0; k < g.nc; ++k) a[gen_digit_reverse(g, k)] = gen_split_inverse(g, X, t.lo2.data(), t.hi2.data(), k);
  run_fft_inplace<true>(g, t, a.data(), nthr);
  const float scale = 1.0f / (float)g.nc;
  for (int i = 0; i < n_fft; ++i) {
    const cf zz = a[g.even ? i >> 1 : i];
    out[i] = ((g.even && (i & 1)) ? zz.im : zz.re) * scale;
  }
  return 0;
}


// returns the number of passes (0: unsupported length); radices written to radix_out[16]
int emu_gen_factor(int n_fft, int* radix_out) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return 0;
  for (int i = 0; i < g.nstages; ++i) radix_out[i] = g.radix[i];
  return g.nstages;
}

// frame: n_fft reals (already windowed / zero padded) -> n_stft complex bins (interleaved re, im)
int emu_gen_rfft(int n_fft, const float* frame, float* out, int nthr) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return -1;
  std::vector<cf> a(gen_buf_elems(g.nc)), b(gen_buf_elems(g.nc));
  for (int n = 0; n < g.nc; ++n) a[gen_pad(n)] = g.even ? cf{frame[2 * n], frame[2 * n + 1]} : cf{frame[n], 0.f};
  const cf* Z = run_fft<false>(g, t, a.data(), b.data(), nthr);
  for (int k = 0; k < g.n_stft; ++k) {
    const cf X = gen_split_forward(g, Z, t.lo2.data(), t.hi2.data(), k);
    out[2 * k] = X.re; out[2 * k + 1] = X.im;
  }
  return 0;
}

// n_stft complex bins -> n_fft reals, scaled like numpy / torch irfft (1/n_fft)
int emu_gen_irfft(int n_fft, const float* spec, float* out, int nthr) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return -1;
  std::vector<cf> a(gen_buf_elems(g.nc)), b(gen_buf_elems(g.nc));
  auto X = [&](int k) { return cf{spec[2 * k], spec[2 * k + 1]}; };
  for (int k = 0; k < g.nc; ++k) a[gen_pad(k)] = gen_split_inverse(g, X, t.lo2.data(), t.hi2.data(), k);
  const cf* z = run_fft<true>(g, t, a.data(), b.data(), nthr);
  const float scale = 1.0f / (float)g.nc;
  for (int i = 0; i < n_fft; ++i) {
    const cf zz = z[gen_pad(g.even ? i >> 1 : i)];
    out[i] = ((g.even && (i & 1)) ? zz.im : zz.re) * scale;
  }
  return 0;
}

// the per-bin update in the reference's op order: a = rebuilt - m * tprev, then the projection the kernels apply to `a`
void emu_gen_gl_update(const float* rebuilt, const float* tprev, float mom, float S, float* out) {
  const cf z = gl_project(cf{rebuilt[0] - tprev[0] * mom, rebuilt[1] - tprev[1] * mom}, S);
  out[0] = z.re; out[1] = z.im;
}

// One frame of the fused Griffin-Lim kernel (gen_gl_kernel): forward in-place passes, the pairwise in-place projection
// gen_pair_project, inverse in-place passes.  frame: n_fft reals; S: n_stft magnitudes; out: n_fft reals = irfft(S * X / |X|).
// exact != 0: the passes take their twiddles from the exact per-pass tables (the kernels' form), else from the two-level table
int emu_gen_gl_frame_padded(int n_fft, const float* frame, const float* S, float* out, int nthr, int ps, int exact) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return -1;
  const std::vector<cf> twt = make_tw_table(g);
  const cf* tw = exact ? twt.data() : nullptr;
  std::vector<cf> a(gen_ibuf_elems(g.nc, ps));
  std::vector<int> rev(g.nc);
  for (int k = 0; k < g.nc; ++k) rev[k] = gen_ipad(gen_digit_reverse(g, k), ps);  // the table the kernels get: padded positions
  for (int n = 0; n < g.nc; ++n) a[gen_ipad(n, ps)] = g.even ? cf{frame[2 * n], frame[2 * n + 1]} : cf{frame[n], 0.f};
  run_fft_inplace<false>(g, t, a.data(), nthr, ps, tw);
  const int npairs = gen_pair_count(g);
  for (int tid = 0; tid < nthr; ++tid)
    for (int k = tid; k < npairs; k += nthr)
      gen_pair_project(g, a.data(), [&](int i) { return rev[i]; }, S, t.lo2.data(), t.hi2.data(), k);
  run_fft_inplace<true>(g, t, a.data(), nthr, ps, tw);
  const float scale = 1.0f / (float)g.nc;
  for (int i = 0; i < n_fft; ++i) {
    const cf zz = a[gen_ipad(g.even ? i >> 1 : i, ps)];
    out[i] = ((g.even && (i & 1)) ? zz.im : zz.re) * scale;
  }
  return 0;
}
int emu_gen_gl_frame(int n_fft, const float* frame, const float* S, float* out, int nthr) {
  return emu_gen_gl_frame_padded(n_fft, frame, S, out, nthr, 0, 0);
}
// what plan creation picks for a geometry: LDS padding shift (given the buffer room in elements) and thread count
int emu_gen_pick_pad(int n_fft, int max_elems) {
  GenGeom g; Tables t;
  if (!make_geom(n_fft, n_fft, 1, g, t)) return -1;
  return gen_pick_pad(g, max_elems);
}
}
