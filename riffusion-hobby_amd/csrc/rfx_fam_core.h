// rfx_fam_core.h - per-thread arithmetic of the ROW-FAMILY frame transform: every geometry with n_fft = 40 h and
// win_length = 10 h, i.e. the reference's default 400 ms / 100 ms (riffusion/spectrogram_params.py:24-27, :62-81) at any
// sample rate that is a multiple of 100 Hz - 48 kHz (h = 480), 32 kHz, 24 kHz, 16 kHz, 8 kHz; 44.1 kHz (h = 441) is the
// specialised engine's own case (rfx_core.h) and runs here only as a cross-check.  Written once for the gfx950 kernels of
// rfx_fam.hip (hipcc) and for the host emulator of the CPU tests (g++), like rfx_core.h / rfx_gen_core.h.
//
// Same decomposition as rfx_core.h with h in place of 441:
//   sample index  n = h J + n'      J = 15..24 (only the 10 windowed blocks are non-zero), n' = 0..h-1
//   bin index     k = k1 + 40 k'    k1 = 0..39, k' = 0..h-1
//   X[k1 + 40 k'] = sum_{n'} w_h^{n' k'} g(n')^{k1} sum_{j=0..9} u[h j + n'] w40^{j k1},   g(n') = exp(-2 pi i (n' + 15 h) / (40 h))
// P1 is rfx_core.h's pruned 40-point transform of ten reals (rows k1 = 0..20 kept, the rest are conjugates); the 21 rows are
// h-point complex FFTs done in place in two passes, h = RA * RB:
//   pass A (thread = (row, i), i < RB):  y = DFT_RA(row[i + q RB]),  row[i + p RB] = y[p] W_h^{i p}
//   pass B (thread = (row, p), p < RA):  R = DFT_RB(row[p RB + q])   ->  R[s] = X[k1 + 40 (p + RA s)]   (stays in registers)
// (decimation in frequency; the inverse retraces it, decimation in time).  The 21 h row outputs ("slots") cover the
// 20 h + 1 one-sided bins: slot k <= n_fft / 2 holds X[k], slot k > n_fft / 2 holds conj(X[n_fft - k]); bins on rows 0 and
// 20 appear twice.  Everything Griffin-Lim does between the two transforms is per bin and commutes with conjugation, so
// slots are independent bins (rfx_core.h).  The hop length is free: frames are overlap-added by gen_fold_kernel.
#pragma once
#include "rfx_gen_core.h"

namespace rfx {

constexpr int kFamRows = 21;  // rows of the 40 h family (the 20 h family keeps 11)

// Round 4: the same machinery for n_fft = 20 h, win_length = 5 h - 22.05 kHz at the reference's default durations
// (8820 / 2205 / 220 with h = 441: the window is not ten "hops", spectrogram_params.py:62-81).  With N = nrad h and the window
// sample i = h j + m (m < h, j < nrad / 4) at frame position left + i, left = (n_fft - win_length) / 2 as torch.stft centres it:
//   X[k1 + nrad k'] = sum_m w_h^{n'' k'} g(m)^{k1} sum_j u[h j + m] w_nrad^{j k1},   g(m) = exp(-2 pi i (m + left) / N),
// n'' = (m + left) mod h the cube column of thread m.  P1 is a pruned nrad-point transform of nrad / 4 reals keeping the rows
// k1 = 0 .. nrad / 2; rows 0 and nrad / 2 hold their bins twice.  (40 h family: left = 15 h, so n'' = m: nothing changes.)
struct FamGeom {
  int h;         // n_fft / nrad = win_length / (nrad / 4)
  int ra, rb;    // h = ra * rb
  int nthr;      // threads per workgroup: max(h, rows rb, rows ra) rounded up to whole waves
  int rs;        // LDS elements between rows of the cube (>= h)
  int fsf;       // floats per frame of the slot-ordered magnitudes: rb * nthr (thread t's slot s at s * nthr + t)
  int n_fft, win, hop, n_stft;
  int nrad;      // 40 or 20: length of the pruned first transform
  int rows;      // nrad / 2 + 1 rows of h points
  int left;      // frame position of the first window sample, (n_fft - win) / 2
  int off;       // left - n_fft / 2: signal position of window sample 0 of frame 0 (before reflection)
};

// the digit pairs the kernels are instantiated for (both digits implemented by gen_dft)
RFX_HD bool fam_digits(int h, int* ra, int* rb) {
  switch (h) {
    case 80: *ra = 10; *rb = 8; return true;     // 8 kHz
    case 160: *ra = 16; *rb = 10; return true;   // 16 kHz
    case 240: *ra = 16; *rb = 15; return true;   // 24 kHz
    case 320: *ra = 20; *rb = 16; return true;   // 32 kHz
    case 441: *ra = 21; *rb = 21; return true;   // 44.1 kHz (cross-check of the specialised engine)
    case 480: *ra = 24; *rb = 20; return true;   // 48 kHz
    default: return false;
  }
}
RFX_HD bool fam_make_geom(int n_fft, int win, int hop, FamGeom* g) {
  if (win * 4 != n_fft || hop < 1) return false;
  int nrad = 0;
  if (n_fft % 40 == 0 && win % 10 == 0) nrad = 40;
  else if (n_fft % 20 == 0 && win % 5 == 0 && n_fft / 20 == 441) nrad = 20;  // 22.05 kHz: the one 20 h geometry instantiated
  if (!nrad) return false;
  const int h = n_fft / nrad;
  int ra, rb;
  if (!fam_digits(h, &ra, &rb)) return false;
  g->h = h;
  g->ra = ra;
  g->rb = rb;
  g->nrad = nrad;
  g->rows = nrad / 2 + 1;
  int n = h;
  if (g->rows * rb > n) n = g->rows * rb;
  if (g->rows * ra > n) n = g->rows * ra;
  g->nthr = (n + 63) / 64 * 64;
  g->rs = h;
  g->fsf = rb * g->nthr;
  g->n_fft = n_fft;
  g->win = win;
  g->hop = hop;
  g->n_stft = n_fft / 2 + 1;
  g->left = (n_fft - win) / 2;
  g->off = g->left - n_fft / 2;
  return true;
}
// bin held by slot s of pass-B thread (row k1, p); *conj_out set when the slot holds the conjugate of that bin
RFX_HD int fam_slot_bin(const FamGeom& g, int k1, int p, int s, bool* conj_out) {
  const int k = k1 + g.nrad * (p + g.ra * s);
  const bool c = k > g.n_fft / 2;
  if (conj_out) *conj_out = c;
  return c ? g.n_fft - k : k;
}

// g(n')^k1 from the eleven values a thread keeps, w[1..10] = g^1 .. g^10 and w[11] = g^20: g^(20-k) = g^20 conj(g^k) - one more
// complex product for the rows 11..19 instead of nine more table loads per frame and eighteen registers.  (20 h family: the
// ten rows 1..10 are the ten values.)
RFX_HD cf fam_g_pow(const cf (&w)[12], int k) { return k <= 10 ? w[k] : k == 20 ? w[11] : cmulc(w[11], w[20 - k]); }

// Which slot writes a bin when the frame leaves in bin order (forward kernel): bins with k mod nrad above nrad / 2 exist as a
// conjugate slot only, bins on rows 0 and nrad / 2 exist twice (the slot k and the slot n_fft - k): the direct one writes
RFX_HD bool fam_slot_is_primary(int nrad, int k1, bool conj) { return !conj || (k1 != 0 && k1 != nrad / 2); }

// ---- the pruned first transform of the 20 h family: v[k] = sum_{j=0..4} u[j] w20^{j k}, k = 0..10 (u real, w20 = exp(-2 pi i / 20)).
// Even / odd j:  Ee[k] = u0 + u2 w^{2k} + u4 w^{4k},  Eo[k] = u1 w^k + u3 w^{3k};  v[k] = Ee + Eo and, because
// w^{j (10 - k)} = (-1)^j conj(w^{j k}),  v[10 - k] = conj(Ee - Eo): only k = 1..4 need the trigonometric sums.
template <class PUT>
RFX_HD void p1_forward_rows20(const float (&u)[5], PUT put) {
  constexpr float C40[40] = RFX_C40F_TABLE;  // cos / sin of 2 pi i / 40: the 20th roots are the even entries
  constexpr float S40[40] = RFX_S40F_TABLE;
  {
    const float ee = u[0] + u[2] + u[4], eo = u[1] + u[3];
    put(0, cf{ee + eo, 0.f});
    put(10, cf{ee - eo, 0.f});
    put(5, cf{u[0] - u[2] + u[4], u[3] - u[1]});  // w20^5 = -i
  }
#pragma unroll
  for (int k = 1; k <= 4; ++k) {
    const float eer = fmaf(C40[(8 * k) % 40], u[4], fmaf(C40[(4 * k) % 40], u[2], u[0]));
    const float eei = -fmaf(S40[(8 * k) % 40], u[4], S40[(4 * k) % 40] * u[2]);
    const float eor = fmaf(C40[(6 * k) % 40], u[3], C40[(2 * k) % 40] * u[1]);
    const float eoi = -fmaf(S40[(6 * k) % 40], u[3], S40[(2 * k) % 40] * u[1]);
    put(k, cf{eer + eor, eei + eoi});
    put(10 - k, cf{eer - eor, -(eei - eoi)});
  }
}
// inverse: y[j] = 0.5 (V0.re + (-1)^j V10.re) + sum_{k=1..9} Re(V[k] w20^{-k j}), j = 0..4 (the caller folds 2 / N and the
// synthesis window into one multiplier).  Rows k and 10 - k pair up: w20^{-(10-k) j} = (-1)^j conj(w20^{-k j}), so
// B[k] = V[k] + conj(V[10-k]) feeds the even outputs and D[k] = V[k] - conj(V[10-k]) the odd ones.
template <class RAW, class FIX>
RFX_HD void p1_inverse_rows20(RAW raw, FIX fix, float (&y)[5]) {
  constexpr float C40[40] = RFX_C40F_TABLE;
  constexpr float S40[40] = RFX_S40F_TABLE;
  {
    const cf V0 = fix(0, raw(0)), V10 = fix(10, raw(10)), V5 = fix(5, raw(5));
    const float he = 0.5f * (V0.re + V10.re), ho = 0.5f * (V0.re - V10.re);
    y[0] = he + V5.re;   // Re(V5 i^j): +re, -im, -re, +im, +re
    y[1] = ho - V5.im;
    y[2] = he - V5.re;
    y[3] = ho + V5.im;
    y[4] = he + V5.re;
  }
#pragma unroll
  for (int k = 1; k <= 4; ++k) {
    const cf Vk = fix(k, raw(k)), Vm = fix(10 - k, raw(10 - k));
    const cf B{Vk.re + Vm.re, Vk.im - Vm.im}, D{Vk.re - Vm.re, Vk.im + Vm.im};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const cf z = (j & 1) ? D : B;  // Re(z w20^{-k j}) = z.re cos(2 pi k j / 20) - z.im sin(2 pi k j / 20)
      y[j] = fmaf(-S40[(2 * k * j) % 40], z.im, fmaf(C40[(2 * k * j) % 40], z.re, y[j]));
    }
  }
}

// ---- P1: thread m < h, cube column `col` = (m + left) mod h.  tw1(k1) = g(m)^k1
template <int NR, class TW>
RFX_HD void fam_p1_forward_store(const float (&u)[NR / 4], TW tw1, cf* cube, int col, int rs) {
  auto put = [&](int k1, cf v) { cube[k1 * rs + col] = k1 == 0 ? v : cmul(v, tw1(k1)); };
  if constexpr (NR == 40) p1_forward_rows(u, put);
  else p1_forward_rows20(u, put);
}
// un-normalised: the caller multiplies by 2 / n_fft (and the synthesis window)
template <int NR, class TW>
RFX_HD void fam_p1_load_inverse(const cf* cube, TW tw1, float (&y)[NR / 4], int col, int rs) {
  auto raw = [&](int k1) { return cube[k1 * rs + col]; };
  auto fix = [&](int k1, cf c) { return k1 == 0 ? c : cmulc(c, tw1(k1)); };
  if constexpr (NR == 40) p1_inverse_rows(raw, fix, y);
  else p1_inverse_rows20(raw, fix, y);
}

// ---- pass A: thread (row, i).  tw(p) = W_h^{i p}, p = 1..RA-1, used batch by batch.
// Twiddle batches.  Every radix but 24: two halves, p = 1 .. (RA-1)/2 | the rest, for both directions.  RA = 24 (48 kHz) streams
// its butterfly instead (below) and wants the twiddles in the order the 8 x 3 decomposition produces / consumes them:
//   forward (DIF): the second stage emits the outputs p = k1 + 8 k2 pair of k1 by pair of k1  -> four batches, p in batch (p % 8) / 2
//   inverse (DIT): the first stage takes the inputs p = 3 n1 + n2 column n2 by column n2      -> three batches, p in batch p % 3
constexpr int fam_tw_batches(int ra, bool inv, bool stream) { return ra == 24 && stream ? (inv ? 3 : 4) : 2; }
constexpr bool fam_tw_in_batch(int ra, bool inv, bool stream, int batch, int p) {
  if (ra == 24 && stream) return inv ? p % 3 == batch : (p % 8) / 2 == batch;
  return batch == 0 ? p <= ra / 2 : p > ra / 2;
}
// `stage(0)` runs once the first row values are requested, `stage(b)` (b >= 1) right before batch b's twiddles are needed - the
// kernels fetch them there: batches of registers instead of 2 (RA - 1) next to the butterfly's own 4 RA.
template <int RA, int RB, bool STREAM = false, class TW, class STAGE = NoStage>
RFX_HD void fam_pass_a_forward(cf* row, int i, TW tw, STAGE stage = STAGE()) {
  if constexpr (RA == 24 && STREAM) {
    // The 24-point butterfly as gen_dft_ct<8, 3> does it, value for value, but streamed: the row is read one column of eight at a
    // time and the outputs leave as soon as their 3-point butterfly is done, so 24 intermediate values and one column are alive
    // instead of 24 inputs + 24 intermediates + 24 outputs (the 48 kHz kernels held 88 - 132 B of scratch until round 5).  In place:
    // every input has been read before the first output is stored (the thread owns exactly these 24 elements).
    cf t[3][8];
    const cf none8[8] = {}, none3[3] = {};
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
      cf a[8], b[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) a[n1] = row[i + (3 * n1 + n2) * RB];
      if (n2 == 0) stage(0);
      gen_dft<8, false>(a, b, none8);
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) t[n2][k1] = gen_const_twiddle<24, false>(b[k1], n2 * k1);
      RFX_SCHED_FENCE();  // (left alone the compiler requests all 24 inputs first)
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      if (k1 > 0 && k1 % 2 == 0) stage(k1 / 2);
      cf a[3] = {t[0][k1], t[1][k1], t[2][k1]}, b[3];
      gen_dft<3, false>(a, b, none3);
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) {
        const int p = k1 + 8 * k2;
        row[i + p * RB] = p == 0 ? b[0] : cmul(b[k2], tw(p));
      }
    }
  } else {
    cf v[RA], y[RA];
    const cf none[RA] = {};
#pragma unroll
    for (int q = 0; q < RA; ++q) v[q] = row[i + q * RB];
    stage(0);
    gen_dft<RA, false>(v, y, none);
    row[i] = y[0];
#pragma unroll
    for (int p = 1; p < RA; ++p)
      if (fam_tw_in_batch(RA, false, false, 0, p)) row[i + p * RB] = cmul(y[p], tw(p));
    stage(1);
#pragma unroll
    for (int p = 1; p < RA; ++p)
      if (fam_tw_in_batch(RA, false, false, 1, p)) row[i + p * RB] = cmul(y[p], tw(p));
  }
}
// inverse: the twiddles come first.  `stage(b)` runs before batch b is read
template <int RA, int RB, bool STREAM = false, class TW, class STAGE = NoStage>
RFX_HD void fam_pass_a_inverse(cf* row, int i, TW tw, STAGE stage = STAGE()) {
  if constexpr (RA == 24 && STREAM) {  // streamed like the forward pass: one column of eight inputs (and its eight twiddles) at a time
    cf t[3][8];
    const cf none8[8] = {}, none3[3] = {};
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
      stage(n2);
      cf a[8], b[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) {
        const int p = 3 * n1 + n2;
        a[n1] = p == 0 ? row[i] : cmulc(row[i + p * RB], tw(p));
      }
      gen_dft<8, true>(a, b, none8);
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) t[n2][k1] = gen_const_twiddle<24, true>(b[k1], n2 * k1);
      RFX_SCHED_FENCE();
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      cf a[3] = {t[0][k1], t[1][k1], t[2][k1]}, b[3];
      gen_dft<3, true>(a, b, none3);
#pragma unroll
      for (int k2 = 0; k2 < 3; ++k2) row[i + (k1 + 8 * k2) * RB] = b[k2];
    }
  } else {
    cf v[RA], y[RA];
    const cf none[RA] = {};
    stage(0);
    v[0] = row[i];
#pragma unroll
    for (int p = 1; p < RA; ++p)
      if (fam_tw_in_batch(RA, true, false, 0, p)) v[p] = cmulc(row[i + p * RB], tw(p));
    stage(1);
#pragma unroll
    for (int p = 1; p < RA; ++p)
      if (fam_tw_in_batch(RA, true, false, 1, p)) v[p] = cmulc(row[i + p * RB], tw(p));
    gen_dft<RA, true>(v, y, none);
#pragma unroll
    for (int q = 0; q < RA; ++q) row[i + q * RB] = y[q];
  }
}

// ---- pass B: thread (row, p).  R[s] = slot k1 + 40 (p + RA s).  `row` points at the thread's RB contiguous elements; with
// VEC (RB even and the block 16-byte aligned: even row stride) the kernels move two elements per LDS instruction.
RFX_HD void fam_ld_pair(const cf* p, cf& a, cf& b, bool vec) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (vec) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 t = *reinterpret_cast<const v4*>(p);
    a = cf{t.x, t.y};
    b = cf{t.z, t.w};
    return;
  }
#endif
  (void)vec;
  a = p[0];
  b = p[1];
}
RFX_HD void fam_st_pair(cf* p, cf a, cf b, bool vec) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (vec) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<v4*>(p) = v4{a.re, a.im, b.re, b.im};
    return;
  }
#endif
  (void)vec;
  p[0] = a;
  p[1] = b;
}
template <int RA, int RB, bool VEC = false>
RFX_HD void fam_pass_b_forward(const cf* row, int p, cf (&R)[RB]) {
  cf v[RB];
  const cf none[RB] = {};
  if (VEC && RB % 2 == 0) {
#pragma unroll
    for (int q = 0; q + 1 < RB; q += 2) fam_ld_pair(row + p * RB + q, v[q], v[q + 1], true);
  } else {
#pragma unroll
    for (int q = 0; q < RB; ++q) v[q] = row[p * RB + q];
  }
  gen_dft<RB, false>(v, R, none);
}
template <int RA, int RB, bool VEC = false>
RFX_HD void fam_pass_b_inverse(cf* row, int p, const cf (&Z)[RB]) {
  cf y[RB];
  const cf none[RB] = {};
  gen_dft<RB, true>(Z, y, none);
  if (VEC && RB % 2 == 0) {
#pragma unroll
    for (int q = 0; q + 1 < RB; q += 2) fam_st_pair(row + p * RB + q, y[q], y[q + 1], true);
  } else {
#pragma unroll
    for (int q = 0; q < RB; ++q) row[p * RB + q] = y[q];
  }
}

}  // namespace rfx
