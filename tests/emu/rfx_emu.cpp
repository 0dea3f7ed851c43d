// Host-side phase emulator of the gfx950 frame kernels.  TEST INFRASTRUCTURE ONLY (built by
// tests/test_fft_core.py with g++): it executes the same per-thread functions of rfx_core.h that
// the HIP kernels inline, looping over the 441 logical threads phase by phase (a loop boundary
// stands where the kernel has a barrier), so that every index map / twiddle / butterfly is checked
// against numpy on the CPU before any GPU time is spent.
#include <cmath>
#include <cstring>
#include <vector>
#include "../../riffusion-hobby_amd/csrc/rfx_core.h"

using namespace rfx;

static void make_tables(std::vector<cf>& tw1, std::vector<cf>& tw2) {
  tw1.resize(441 * 21);
  tw2.resize(21 * 21);
  const double PI2 = 6.283185307179586476925286766559;
  for (int n = 0; n < 441; ++n)
    for (int k1 = 0; k1 < 21; ++k1) {
      long long e = ((long long)k1 * (n + 6615)) % 17640;
      tw1[n * 21 + k1] = cf{(float)cos(PI2 * e / 17640.0), (float)(-sin(PI2 * e / 17640.0))};
    }
  for (int i = 0; i < 21; ++i)
    for (int j = 0; j < 21; ++j) {
      int e = (i * j) % 441;
      tw2[i * 21 + j] = cf{(float)cos(PI2 * e / 441.0), (float)(-sin(PI2 * e / 441.0))};
    }
}

extern "C" {

// seg: 4410 windowed samples u[441*j+n'].  out: 9261 complex slots indexed [q=k1*21+ka][kb]
void emu_forward(const float* seg, float* out_slots) {
  std::vector<cf> tw1, tw2, cube(kCubeElems);
  make_tables(tw1, tw2);
  for (int n = 0; n < 441; ++n) {  // P1
    float u[10];
    for (int j = 0; j < 10; ++j) u[j] = seg[441 * j + n];
    cf t1[21];
    for (int k = 0; k < 21; ++k) t1[k] = tw1[n * 21 + k];
    p1_forward_store(u, [&](int k) { return t1[k]; }, cube.data(), n);
  }
  for (int k1 = 0; k1 < 21; ++k1)  // P2
    for (int b = 0; b < 21; ++b) {
      cf t2[21];
      for (int i = 0; i < 21; ++i) t2[i] = tw2[b * 21 + i];
      p2_forward(cube.data(), [&](int k) { return t2[k]; }, k1, b);
    }
  for (int k1 = 0; k1 < 21; ++k1)  // P3
    for (int ka = 0; ka < 21; ++ka) {
      cf R[21];
      p3_forward(cube.data(), R, k1, ka);
      for (int kb = 0; kb < 21; ++kb) {
        out_slots[((k1 * 21 + ka) * 21 + kb) * 2 + 0] = R[kb].re;
        out_slots[((k1 * 21 + ka) * 21 + kb) * 2 + 1] = R[kb].im;
      }
    }
}

// in_slots: 9261 complex [q][kb] -> y: 4410 floats, y[441*j+n'] = sum (un-normalised, before 2/N*window)
void emu_inverse(const float* in_slots, float* y) {
  std::vector<cf> tw1, tw2, cube(kCubeElems);
  make_tables(tw1, tw2);
  for (int k1 = 0; k1 < 21; ++k1)
    for (int ka = 0; ka < 21; ++ka) {
      cf Z[21], t2[21];
      for (int kb = 0; kb < 21; ++kb)
        Z[kb] = cf{in_slots[((k1 * 21 + ka) * 21 + kb) * 2], in_slots[((k1 * 21 + ka) * 21 + kb) * 2 + 1]};
      for (int i = 0; i < 21; ++i) t2[i] = tw2[ka * 21 + i];
      p3_inverse(cube.data(), Z, [&](int k) { return t2[k]; }, k1, ka);
    }
  for (int k1 = 0; k1 < 21; ++k1)
    for (int b = 0; b < 21; ++b) p2_inverse(cube.data(), k1, b);
  for (int n = 0; n < 441; ++n) {
    cf t1[21];
    for (int k = 0; k < 21; ++k) t1[k] = tw1[n * 21 + k];
    float yy[10];
    p1_load_inverse(cube.data(), [&](int k) { return t1[k]; }, yy, n);
    for (int j = 0; j < 10; ++j) y[441 * j + n] = yy[j];
  }
}

void emu_slot_maps(int* bin, int* conj, int* pos_c, int* pos_f) {
  for (int k1 = 0; k1 < 21; ++k1)
    for (int ka = 0; ka < 21; ++ka)
      for (int kb = 0; kb < 21; ++kb) {
        int i = (k1 * 21 + ka) * 21 + kb;
        bool c;
        bin[i] = slot_bin(k1, ka, kb, &c);
        conj[i] = c;
        pos_c[i] = slot_pos_c(k1 * 21 + ka, kb);
        pos_f[i] = slot_pos_f(k1 * 21 + ka, kb);
      }
}

// the counter RNG of the random starts (rfx_core.h): `n` values of frame `frame` under `seed`
void emu_rand_unit(unsigned long long seed, unsigned long long frame, int n, float* out) {
  const unsigned key = rand_frame_key(seed, frame);
  for (int f = 0; f < n; ++f) out[f] = rand_unit(key, f);
}
void emu_rand_unit_pair(unsigned long long seed, unsigned long long frame, int n, float* out_re_im) {
  const unsigned key = rand_frame_key(seed, frame);
  for (int b = 0; b < n; ++b) {
    const cf r = rand_unit_pair(key, b);
    out_re_im[2 * b] = r.re;
    out_re_im[2 * b + 1] = r.im;
  }
}

void emu_dft21(float* x, int inv) {
  cf v[21];
  for (int i = 0; i < 21; ++i) v[i] = cf{x[2 * i], x[2 * i + 1]};
  if (inv) dft21<true>(v); else dft21<false>(v);
  for (int i = 0; i < 21; ++i) { x[2 * i] = v[i].re; x[2 * i + 1] = v[i].im; }
}
}
