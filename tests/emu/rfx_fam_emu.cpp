// Host-side emulator of the ROW-FAMILY frame transform (csrc/rfx_fam_core.h).  TEST INFRASTRUCTURE ONLY, built with g++ by
// tests/test_fam_core.py: it runs the per-thread functions the gfx950 kernels of rfx_fam.hip inline, looping over the logical
// threads of a workgroup phase by phase (a loop boundary stands where the kernel has a barrier).
#include <cmath>
#include <vector>
#include "../../riffusion-hobby_amd/csrc/rfx_fam_core.h"

using namespace rfx;

namespace {
int g_stream_fwd = 1, g_stream_inv = 0;  // which form of the radix-24 pass A runs (emu_fam_set_stream)
struct FamTables {
  std::vector<cf> tw1, twa;
};
void make_tables(const FamGeom& g, FamTables& t) {
  const double PI2 = 6.283185307179586476925286766559;
  t.tw1.resize((size_t)g.rows * g.h);
  for (int k1 = 0; k1 < g.rows; ++k1)
    for (int n = 0; n < g.h; ++n) {
      const long long e = ((long long)k1 * (n + g.left)) % g.n_fft;
      t.tw1[(size_t)k1 * g.h + n] = cf{(float)cos(PI2 * (double)e / g.n_fft), (float)(-sin(PI2 * (double)e / g.n_fft))};
    }
  t.twa.resize((size_t)g.rb * (g.ra - 1));
  for (int i = 0; i < g.rb; ++i)
    for (int p = 1; p < g.ra; ++p) {
      const int e = (i * p) % g.h;
      t.twa[(size_t)(p - 1) * g.rb + i] = cf{(float)cos(PI2 * e / (double)g.h), (float)(-sin(PI2 * e / (double)g.h))};
    }
}
template <int RA, int RB, int NR>
void forward(const FamGeom& g, const FamTables& t, const float* win_samples, std::vector<cf>& slots) {
  constexpr int ROWS = NR / 2 + 1, WH = NR / 4;
  std::vector<cf> cube((size_t)ROWS * g.rs);
  for (int n = 0; n < g.h; ++n) {  // P1
    float u[WH];
    for (int j = 0; j < WH; ++j) u[j] = win_samples[j * g.h + n];
    cf w[12];  // what a thread of the kernel holds: g^1 .. g^10 and (40 h family) g^20
    for (int k = 1; k <= (NR == 40 ? 11 : 10); ++k) w[k] = t.tw1[(size_t)(k <= 10 ? k : 20) * g.h + n];
    fam_p1_forward_store<NR>(u, [&](int k1) { return fam_g_pow(w, k1); }, cube.data(), (n + g.left) % g.h, g.rs);
  }
  for (int tid = 0; tid < ROWS * RB; ++tid) {  // pass A
    const int row = tid / RB, i = tid % RB;
    // (the radix-24 pass exists plain and streamed: the kernels stream the forward pass and not the inverse one, the default here)
    if (g_stream_fwd) fam_pass_a_forward<RA, RB, true>(cube.data() + (size_t)row * g.rs, i, [&](int p) { return t.twa[(size_t)(p - 1) * RB + i]; });
    else fam_pass_a_forward<RA, RB, false>(cube.data() + (size_t)row * g.rs, i, [&](int p) { return t.twa[(size_t)(p - 1) * RB + i]; });
  }
  slots.assign((size_t)g.fsf, cf{0.f, 0.f});
  for (int tid = 0; tid < ROWS * RA; ++tid) {  // pass B
    const int row = tid / RA, p = tid % RA;
    cf R[RB];
    fam_pass_b_forward<RA, RB>(cube.data() + (size_t)row * g.rs, p, R);
    for (int s = 0; s < RB; ++s) slots[(size_t)s * g.nthr + tid] = R[s];
  }
}
template <int RA, int RB, int NR>
void inverse(const FamGeom& g, const FamTables& t, const std::vector<cf>& slots, float* win_samples) {
  constexpr int ROWS = NR / 2 + 1, WH = NR / 4;
  std::vector<cf> cube((size_t)ROWS * g.rs);
  for (int tid = 0; tid < ROWS * RA; ++tid) {
    const int row = tid / RA, p = tid % RA;
    cf Z[RB];
    for (int s = 0; s < RB; ++s) Z[s] = slots[(size_t)s * g.nthr + tid];
    fam_pass_b_inverse<RA, RB>(cube.data() + (size_t)row * g.rs, p, Z);
  }
  for (int tid = 0; tid < ROWS * RB; ++tid) {
    const int row = tid / RB, i = tid % RB;
    if (g_stream_inv) fam_pass_a_inverse<RA, RB, true>(cube.data() + (size_t)row * g.rs, i, [&](int p) { return t.twa[(size_t)(p - 1) * RB + i]; });
    else fam_pass_a_inverse<RA, RB, false>(cube.data() + (size_t)row * g.rs, i, [&](int p) { return t.twa[(size_t)(p - 1) * RB + i]; });
  }
  const float sc = 2.0f / (float)g.n_fft;
  for (int n = 0; n < g.h; ++n) {
    float y[WH];
    cf w[12];
    for (int k = 1; k <= (NR == 40 ? 11 : 10); ++k) w[k] = t.tw1[(size_t)(k <= 10 ? k : 20) * g.h + n];
    fam_p1_load_inverse<NR>(cube.data(), [&](int k1) { return fam_g_pow(w, k1); }, y, (n + g.left) % g.h, g.rs);
    for (int j = 0; j < WH; ++j) win_samples[j * g.h + n] = y[j] * sc;
  }
}
template <int RA, int RB, int NR = 40>
int run(const FamGeom& g, int dir, const float* in, float* out) {
  FamTables t;
  make_tables(g, t);
  std::vector<cf> slots;
  if (dir == 0) {  // in: the win windowed samples of a frame; out: one-sided spectrum (n_fft/2 + 1 complex), duplicates checked
    forward<RA, RB, NR>(g, t, in, slots);
    std::vector<int> seen(g.n_stft, 0);
    for (int k1 = 0; k1 < g.rows; ++k1)
      for (int p = 0; p < RA; ++p)
        for (int s = 0; s < RB; ++s) {
          bool cj;
          const int bin = fam_slot_bin(g, k1, p, s, &cj);
          const cf v = slots[(size_t)s * g.nthr + k1 * RA + p];
          const cf x{v.re, cj ? -v.im : v.im};
          if (seen[bin]) {  // a duplicate slot must agree with the first one
            const float dr = out[2 * bin] - x.re, di = out[2 * bin + 1] - x.im;
            const float mag = fabsf(x.re) + fabsf(x.im) + 1e-20f;
            if (fabsf(dr) + fabsf(di) > 1e-3f * mag + 1e-3f) return -2;
          }
          seen[bin]++;
          out[2 * bin] = x.re;
          out[2 * bin + 1] = x.im;
        }
    for (int b = 0; b < g.n_stft; ++b)
      if (seen[b] < 1 || seen[b] > 2) return -3;
    return 0;
  }
  // in: one-sided spectrum; out: the win samples the window covers of its inverse real FFT
  slots.assign((size_t)g.fsf, cf{0.f, 0.f});
  for (int k1 = 0; k1 < g.rows; ++k1)
    for (int p = 0; p < RA; ++p)
      for (int s = 0; s < RB; ++s) {
        bool cj;
        const int bin = fam_slot_bin(g, k1, p, s, &cj);
        slots[(size_t)s * g.nthr + k1 * RA + p] = cf{in[2 * bin], cj ? -in[2 * bin + 1] : in[2 * bin + 1]};
      }
  inverse<RA, RB, NR>(g, t, slots, out);
  return 0;
}
}  // namespace

extern "C" {
void emu_fam_set_stream(int fwd, int inv) { g_stream_fwd = fwd; g_stream_inv = inv; }
// dir 0: forward, dir 1: inverse; rs_pad: extra LDS elements between cube rows (the kernels pad by up to 7)
int emu_fam_transform(int n_fft, int dir, int rs_pad, const float* in, float* out) {
  FamGeom g;
  if (!fam_make_geom(n_fft, n_fft / 4, n_fft / 40, &g)) return -1;
  g.rs += rs_pad;
  if (g.nrad == 20) return g.h == 441 ? run<21, 21, 20>(g, dir, in, out) : -1;
  switch (g.h) {
    case 80: return run<10, 8>(g, dir, in, out);
    case 160: return run<16, 10>(g, dir, in, out);
    case 240: return run<16, 15>(g, dir, in, out);
    case 320: return run<20, 16>(g, dir, in, out);
    case 441: return run<21, 21>(g, dir, in, out);
    case 480: return run<24, 20>(g, dir, in, out);
  }
  return -1;
}
// how many primary slots (fam_slot_is_primary: the forward kernel's writers) each one-sided bin has: must be exactly one
int emu_fam_primary_writers(int n_fft, int* count /* [n_fft / 2 + 1] */) {
  FamGeom g;
  if (!fam_make_geom(n_fft, n_fft / 4, n_fft / 40, &g)) return -1;
  for (int b = 0; b < g.n_stft; ++b) count[b] = 0;
  for (int k1 = 0; k1 < g.rows; ++k1)
    for (int p = 0; p < g.ra; ++p)
      for (int s = 0; s < g.rb; ++s) {
        bool cj;
        const int bin = fam_slot_bin(g, k1, p, s, &cj);
        if (fam_slot_is_primary(g.nrad, k1, cj)) count[bin]++;
      }
  return 0;
}
int emu_fam_geom(int n_fft, int win, int hop, int* out6) {
  FamGeom g;
  if (!fam_make_geom(n_fft, win, hop, &g)) return -1;
  out6[0] = g.h; out6[1] = g.ra; out6[2] = g.rb; out6[3] = g.nthr; out6[4] = g.fsf; out6[5] = g.n_stft;
  return 0;
}
}

// Griffin-Lim of one clip exactly as the kernels of rfx_fam.hip + gen_fold_kernel run it (rfx_api.hip::gen_griffinlim): launch
// 0 synthesises S * angles0, launch it >= 1 analyses x_{it-1} - m x_{it-2} (momentum applied in the time domain), projects every
// slot, synthesises; the windowed frames are overlap-added and divided by the window envelope.  Host arithmetic (exact sqrt and
// divide in gl_project where the device uses v_rsq_f32).
namespace {
template <int RA, int RB, int NR = 40>
int gl_run(const FamGeom& g, int T, int n_iter, float momentum, const float* mag, const float* ang, const float* win, float* out) {
  FamTables t;
  make_tables(g, t);
  const int L = g.hop * (T - 1);
  if (n_iter > 0 && L <= g.n_fft / 2) return -4;
  const float m = momentum / (1.f + momentum);
  std::vector<float> gen[3];
  for (auto& v : gen) v.assign((size_t)L, 0.f);
  std::vector<float> frames((size_t)T * g.win), u(g.win), y(g.win);
  std::vector<cf> slots;
  const float sc = 1.0f;  // inverse<> already applies 2 / n_fft
  for (int it = 0; it <= n_iter; ++it) {
    const std::vector<float>& xc = gen[(it + 2) % 3];
    const std::vector<float>& xp = gen[(it + 1) % 3];
    for (int fr = 0; fr < T; ++fr) {
      if (it == 0) {
        slots.assign((size_t)g.fsf, cf{0.f, 0.f});
      } else {
        for (int j = 0; j < g.win; ++j) {
          const int p = reflect_index(g.hop * fr + j + g.off, L);
          const float x = it >= 2 ? fmaf(-m, xp[p], xc[p]) : xc[p];
          u[j] = x * win[j];
        }
        forward<RA, RB, NR>(g, t, u.data(), slots);
      }
      for (int k1 = 0; k1 < g.rows; ++k1)
        for (int p = 0; p < RA; ++p)
          for (int s2 = 0; s2 < RB; ++s2) {
            bool cj;
            const int bin = fam_slot_bin(g, k1, p, s2, &cj);
            cf& z = slots[(size_t)s2 * g.nthr + k1 * RA + p];
            const float S = mag[(size_t)bin * T + fr];
            if (it == 0) {
              const cf a{ang[2 * ((size_t)bin * T + fr)], ang[2 * ((size_t)bin * T + fr) + 1]};
              z = cf{S * a.re, cj ? -(S * a.im) : S * a.im};
            } else {
              z = gl_project(z, S);
            }
          }
      inverse<RA, RB, NR>(g, t, slots, y.data());
      for (int j = 0; j < g.win; ++j) frames[(size_t)fr * g.win + j] = y[j] * sc * win[j];
    }
    // gen_fold_kernel: sample p of the output sits at P = p + n_fft/2 of the padded signal; frame t contributes j = P - hop t - left
    std::vector<float>& dst = gen[it % 3];
    for (int p = 0; p < L; ++p) {
      const int q = p + g.n_fft / 2 - g.left;
      int tlo = q - (g.win - 1) <= 0 ? 0 : (q - (g.win - 1) + g.hop - 1) / g.hop;
      int thi = q / g.hop;
      if (thi > T - 1) thi = T - 1;
      float acc = 0.f, env = 0.f;
      for (int tt = tlo; tt <= thi; ++tt) {
        const int j = q - g.hop * tt;
        acc += frames[(size_t)tt * g.win + j];
        env = fmaf(win[j], win[j], env);
      }
      dst[p] = acc / env;
    }
  }
  for (int p = 0; p < L; ++p) out[p] = gen[n_iter % 3][p];
  return 0;
}
}  // namespace

extern "C" int emu_fam_griffinlim(int n_fft, int hop, int T, int n_iter, float momentum, const float* mag, const float* ang,
                                  const float* win, float* out) {
  FamGeom g;
  if (!fam_make_geom(n_fft, n_fft / 4, hop, &g)) return -1;
  if (g.nrad == 20) return g.h == 441 ? gl_run<21, 21, 20>(g, T, n_iter, momentum, mag, ang, win, out) : -1;
  switch (g.h) {
    case 80: return gl_run<10, 8>(g, T, n_iter, momentum, mag, ang, win, out);
    case 160: return gl_run<16, 10>(g, T, n_iter, momentum, mag, ang, win, out);
    case 240: return gl_run<16, 15>(g, T, n_iter, momentum, mag, ang, win, out);
    case 320: return gl_run<20, 16>(g, T, n_iter, momentum, mag, ang, win, out);
    case 441: return gl_run<21, 21>(g, T, n_iter, momentum, mag, ang, win, out);
    case 480: return gl_run<24, 20>(g, T, n_iter, momentum, mag, ang, win, out);
  }
  return -1;
}
