// rfx_kernels.h - argument blocks and host launchers of the gfx950 kernels (internal to librfx.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "rfx_core.h"
#include "rfx_gen_core.h"
#include "rfx_fam_core.h"

namespace rfx {

// Per-row numeric range of Griffin-Lim (round 6): row r of a call analyses its signal times row_scale[2 r] (a power of two that
// brings the row's magnitudes to about 2^25) and projects with eps^2 = row_scale[2 r + 1] (rfx_core.h::gl_project); written by
// range_scale_kernel (rfx_imel.hip) from the data or the caller's magnitude_hint.  A null table means {1, 1e-32}.
struct GlArgs {
  const float* S;             // [B*T][kFrameStride] magnitudes, slot_pos_f order
  const float* row_scale;     // [B][2] or null
  const cf* angles0;          // optional injected initial angles, slot_pos_c order (MODE 0)
  const float* audio_in[2];   // parity partial sums of x_k, the current estimate, [B][Lpad]
  const float* audio_prev[2]; // parity partial sums of x_{k-1} (MODE 2)
  float* audio_out[2];        // x_{k+1}
  const float* out_scale;     // [L]  (2/N) / window-envelope  (torch.istft's division by sum w^2)
  const cf* tw1;              // [21][441]
  const cf* tw2;              // [21][21]
  const float* win;           // [4410]
  int B, T, L, Lpad;          // (the runs of a launch are its workgroups: gridDim.x)
  // Run lengths by dispatch order (round 5; gl_run_start below): the first `run_h` workgroups of a launch are the ones the
  // dispatcher places first - one per CU - and every later one joins a CU that already runs one.  The earlier workgroup's waves
  // are the older ones and win the CU's issue arbitration: measured (tools/probe_wgclock.py, 64 tiles) it finishes its 64 frames
  // in 656 us, its partner in 738 us, every launch, on every CU (spread 0.3 % / 0.1 %) - and for the last tenth of the launch
  // every CU runs one workgroup alone, at three quarters of its two-workgroup rate.  So the early workgroups get run_w1 / 1000
  // and the late ones run_w2 / 1000 of the mean run length, and the pair finishes together.
  int run_h, run_w1, run_w2;
  float mom;                  // momentum / (1 + momentum)
  unsigned long long seed;
  unsigned long long frame_base;  // rfx_call_options::row_base * T: frame (row, t) of this call draws its phases from key (seed, frame_base + row T + t)
  unsigned long long* timing;  // optional [nblocks][8] phase timers (RFX_TIMING builds only)
#ifdef RFX_WGCLOCK
  int launch;                  // diagnostic build: index of this launch inside one rfx_griffinlim call (tools/probe_wgclock.py)
#endif
};

// first frame of run b when the launch's N frames are cut into `runs` runs: the runs b < h weigh w1, the others w2 (integers,
// any common unit: round 6 passes q + 1 and q groups for N = q runs + r, h = r).  Exact integer arithmetic, the same on the host
// (checks) and in the kernel; N * (w1 h + w2 (runs - h)) must fit 63 bits (N < 2^27 groups and weights <= N do).
RFX_HD long long gl_run_start(long long b, long long runs, long long N, long long h, long long w1, long long w2) {
  const long long hb = b < h ? b : h, ht = runs < h ? runs : h;
  const long long Wb = w1 * hb + w2 * (b - hb), Wt = w1 * ht + w2 * (runs - ht);
  return N * Wb / Wt;
}
// Round 6: CANONICAL GROUPS.  torch.istft's overlap-add sums ten windowed frames per sample, and fp32 addition is not associative:
// where two runs share a hop block, the block is the sum of two partial chains instead of one - so until round 5 a clip's audio
// depended (in the last bit, then - Griffin-Lim amplifies it - in the last few PCM steps) on where the run boundaries of the
// launch fell, i.e. on the batch it was converted in.  Now every row (clip-channel) is cut into the same groups of kGlGroup frames
// whatever the batch is, EVERY group boundary splits the chains of the nine blocks across it (the run kernel flushes its partial
// sums there exactly as it does at the end of a run, the fold kernel of the per-frame form sums the two sides separately), and
// runs are whole groups.  A block's value is then s * chain(frames before the cut) + s * chain(frames from the cut on) for every
// partition and both forms: a clip's bits are a function of the clip, its seed and its row index alone.  (>= 10 frames per group:
// a block's ten frames span at most one cut.  16 divides the 64 frames per run of the headline shape, 64 tiles x 512 frames on 512
// resident workgroups; the price is the granularity - a launch lasts as long as its longest run, a whole number of groups.)
constexpr int kGlGroup = 16;
RFX_HD int gl_groups_per_row(int T) { return (T + kGlGroup - 1) / kGlGroup; }
// first frame (counted row after row) of run b of a launch over B rows of T frames: gl_run_start in units of groups
RFX_HD long long gl_run_start_frame(long long b, long long runs, int B, int T, long long h, long long w1, long long w2) {
  const long long ng = gl_groups_per_row(T);
  const long long u = gl_run_start(b, runs, (long long)B * ng, h, w1, w2);
  const long long row = u / ng;
  return row * T + (u - row * ng) * kGlGroup;
}

hipError_t launch_gl_iter(int mode, const GlArgs& g, int nblocks, hipStream_t stream);
// per-device set-up, called by rfx_plan_create with the plan's device current
hipError_t prepare_frame_kernels();  // dynamic-LDS attributes of the STFT and Griffin-Lim kernels
hipError_t prepare_gl_kernels();
int gl_blocks_per_cu();  // resident Griffin-Lim workgroups per CU on the current device (occupancy query)
hipError_t launch_gl_combine(const float* a0, const float* a1, float* out, int B, int L, int Lpad, hipStream_t stream);

// small-batch (latency) form: one frame per unit of work, synthesis frames to a buffer, then a fold
struct GlFrameArgs {
  const float* S;           // [B*T][kFrameStride] magnitudes
  const cf* angles0;        // optional injected initial angles (MODE 0)
  const float* audio_in;    // x_k      [B][Lpad]
  const float* audio_prev;  // x_{k-1}  (MODE 2)
  float* frames;            // [B*T][4416] windowed synthesis frames
  const float* row_scale;   // [B][2] or null (see GlArgs)
  const cf* tw1;
  const cf* tw2;
  const float* win;
  int B, T, L, Lpad;
  float mom;
  unsigned long long seed;
  unsigned long long frame_base;  // as GlArgs::frame_base
};
hipError_t launch_gl_frame(int mode, const GlFrameArgs& g, int nblocks, hipStream_t stream);
hipError_t launch_gl_fold(const float* frames, const float* win, const float* scale, float* out, int B, int T, int L, size_t out_stride, hipStream_t stream);
size_t gl_frame_buffer_bytes(int B, int T);

// layout conversion between the reference's (B, n_stft, T) tensors and slot-major frames
hipError_t launch_pack_mag(const float* lin_bft, float* S_slots, int B, int T, hipStream_t stream);
hipError_t launch_pack_angles(const cf* ang_bft, cf* slots, int B, int T, hipStream_t stream);
hipError_t launch_unpack_mag(const float* slots, float* out_bft, int B, int T, hipStream_t stream);
hipError_t launch_unpack_complex(const cf* slots, cf* out_bft, int B, int T, hipStream_t stream);

// forward STFT magnitude of arbitrary-length waveforms: wave [B][Lw] -> mag slots [B*T][kFrameStride]
struct StftArgs {
  const float* wave;   // [B][Lw]
  float* mag;          // [B*T][kFrameStride] |X| in slot_pos_f order (nullable)
  cf* spec;            // [B*T][kFrameStride] X in slot_pos_c order (nullable)
  const cf* tw1;
  const cf* tw2;
  const float* win;
  int B, T, Lw, frames_per_block;
};
hipError_t launch_stft(const StftArgs& a, hipStream_t stream);

// fused forward path: waveform -> framed transform -> |X| -> banded mel projection, nothing but the (B, M, T) mel
// amplitudes leaves the chip.  Valid for banded filterbanks (every filter's support one contiguous run of bins).
struct StftMelArgs {
  const float* wave;     // [B][Lw]
  float* mel;            // [B][M][T] result (written by the transpose that follows the transform kernel)
  float* mel_tm;         // [B][T][Mpad] frame-major scratch the transform kernel writes
  const cf* tw1;
  const cf* tw2;
  const float* win;
  const float* band_wt;  // [band_rows][Mpad]: weight of filter m on its i-th bin (band_lo[m] + i), zero past its end
  const int* band_addr;  // [band_rows][Mpad]: LDS float index (2 * cube_at(k1, ka, 0) + kb) of that bin's primary slot
  const int* band_lo;    // [Mpad] first bin of filter m's band (padding filters: 0)
  const int* band_len;   // [Mpad] bins in filter m's band (padding filters: 0)
  int B, T, Lw, frames_per_block;
  int run_skew;          // stft_mel2_kernel: > 0 = the first-dispatched half of the grid walks runs this many frames longer, the other half as many shorter
  int M, Mpad;           // Mpad = M rounded up to 64
  int f_lo, f_hi;        // bins with a non-zero filterbank row: [f_lo, f_hi)
  // product form (stft_mel2_kernel; valid when slot_tab != nullptr): the thread that owns a bin's primary slot multiplies
  // |X| by the bin's two filterbank weights and scatters the products group by group; a filter is then two contiguous sums
  const void* slot_tab;  // [21][kQPad] x {w0, w1}: weights of the slot's bin on its first / second filter (zero for a slot that contributes nothing)
  const int* slot_at;    // [21][kQPad] LDS position of the slot's w0 product (the lane's dump position for a slot that contributes nothing)
  const int* pad_tab;    // [kMelPadsPerThread][kQPad] LDS positions of the zero padding this thread rewrites every frame (dump if none)
  const int* filt_seg;   // [2][Mpad]: rising segment, falling segment, each as (first float << 4) | number of 16-byte reads
  int prod_arr;          // floats between the w0 and the w1 product arrays (and between their dump floats)
  unsigned kb_mask;      // bit kb set when any thread's slot kb contributes
  // the same tables in the packed form the default-bank kernel reads (round 5; null when the bank needs kb outside kKbMaskLow or
  // more than kMelProdArr floats per array): LDS BYTE addresses as 16-bit halves, prod_arr == kMelProdArr
  const unsigned* pk_at;   // [5][kQPad]: word i = byte address of the w0 product of the thread's contributing slots 2 i (low half) and 2 i + 1 (high half), slots in increasing kb
  const unsigned* pk_pad;  // [kQPad][2]: the thread's kMelPadsPerThread zero-padding byte addresses, two per word
  const unsigned* pk_seg;  // [Mpad][2]: {rising, falling} segment of filter m, each (first float << 4) | number of 16-byte reads
  // image_util.image_from_spectrogram's maximum (image_util.py:41) taken on the fly (rfx_image_from_waveform): when not null,
  // max_keys[clip * chunks + chunk] receives the order-preserving key (rfx_codec.hip::max_key) of the largest mel amplitude the
  // workgroup walking run `chunk` of row `clip` forms (chunks = ceil(T / frames_per_block); every word written exactly once per
  // launch: no atomics, nothing to zero); the launcher then leaves out the transpose to (B, M, T)
  unsigned* max_keys;
  int max_group;           // (unused since round 6: an image's words are the C * chunks consecutive ones of its rows)
};
constexpr int kMelPadsPerThread = 4;
// the kb (of a thread's 21 slots) that can contribute, as a compile-time set: 0x1F001F (kb 0..4 and 16..20) covers every bank that
// ends at or below bin 4200 (the default 0-10 kHz bank: bins 1..4000), 0x1FFFFF any bank
constexpr unsigned kKbMaskLow = 0x1F001Fu, kKbMaskAll = 0x1FFFFFu;
constexpr int kMelProdArr = 8192;  // floats between the w0 and the w1 product arrays whenever the first one ends below it (a compile-time LDS offset for the packed form)
hipError_t launch_stft_mel(const StftMelArgs& a, hipStream_t stream);

// mel projection GEMM: out[b][m][t] = sum_p fbs[p][m] * mag[b*T+t][p]
struct MelArgs {
  const float* mag;     // [N][kFrameStride]
  const float* fbs;     // [kFrameStride][Mp] slot-ordered filterbank, columns zero-padded to Mp
  const int* kblocks;   // indices of the 32-position K blocks that contain a non-zero filterbank row
  int n_kblocks;
  float* out;           // [B][M][T]
  int M, Mp, N, T;      // N = B*T frames, Mp = M rounded up to the 128-row tile
};
hipError_t launch_mel_gemm(const MelArgs& a, hipStream_t stream);

// Per-wave register budgets (bins per thread) of the InverseMelScale fast path: wave w owns the short groups
// 64w .. 64w+63 and the long groups M-1-64w .. M-64-64w.  Sized exactly to the reference's default bank
// (0-10 kHz, 512 HTK filters over 8821 bins); plan creation falls back to the uniform kernel when a bank
// does not fit.
constexpr int kImelLoCap[4] = {2, 3, 5, 6};
constexpr int kImelHiCap[4] = {23, 16, 12, 9};
// a second set for banks whose edges sit elsewhere: mel_scale_type "slaney" (spectrogram_params.py:35; linear below 1 kHz,
// so its low groups are longer, and its top groups reach 26 bins)
constexpr int kImelLoCapWide[4] = {5, 3, 5, 6};
constexpr int kImelHiCapWide[4] = {26, 17, 12, 9};
// a third set (round 5) for the LINE-FORM group kernel (rfx_imel.hip::imel_line_kernel_perwave): the long groups keep one register
// per bin (weights and momentum buffer are lines per group), so a thread can hold 62 of them - 512 filters up to the Nyquist
// frequency (group sizes 1.5 .. 61 bins), htk or slaney scale, or 256 / 384 filters over the default 0 - 10 kHz
constexpr int kImelLoCapLine[4] = {6, 5, 7, 11};
constexpr int kImelHiCapLine[4] = {62, 40, 26, 18};
// Wave kernel (round 4, rfx_imel.hip::imel_wave_kernel): ONE wave per frame.  The 512 groups are dealt to the 64 lanes in eight
// chunks of 64 consecutive groups, even chunks in lane order and odd chunks reversed (group 64 c + lane / 64 c + 63 - lane), so a
// lane's bin count is within 4 % of the mean although group sizes grow 12-fold over the bank, and every group's neighbours sit
// in the adjacent lane (a DPP wave shift) or, at the chunk seams, in the lane itself.  Budget per chunk in register PAIRS:
constexpr int kImelWaveChunks = 8;
constexpr int kImelWavePairs[kImelWaveChunks] = {2, 2, 3, 3, 5, 6, 8, 12};
// leading pairs that EVERY lane of the chunk fills (plan creation checks n >= 2 full for each group); the pairs behind them carry a
// per-lane 0 / 1 mask for the slots a shorter group leaves empty
constexpr int kImelWaveFullPairs[kImelWaveChunks] = {0, 1, 1, 2, 3, 4, 5, 8};
RFX_HD int imel_wave_group(int chunk, int lane) { return (chunk & 1) ? 64 * chunk + 63 - lane : 64 * chunk + lane; }
#ifndef RFX_IMEL_WAVE
#define RFX_IMEL_WAVE 1  // 0: build without selecting the wave kernel (A/B against the group kernels)
#endif

// banded InverseMelScale SGD (torchaudio 0.13 semantics), one workgroup per frame
struct ImelTables {
  const float* csr_w;    // [nnz] filterbank weights, mel-major (column m = bins fs[m] .. fe[m]-1)
  const int* csr_ptr;    // [M+1]
  const int* band_lo;    // [M] first bin of mel m's band
  const int* bin_m0;     // [n_stft] first mel a bin feeds (-1: zero filterbank row)
  const float* bin_w0;   // [n_stft] weight into mel m0
  const float* bin_w1;   // [n_stft] weight into mel m0+1 (0 if none)
  const int* bin_pos;    // [n_stft] slot position of the bin's primary slot
  const int* bin_pos2;   // [n_stft] slot position of its duplicate slot, or -1
  const int* pos_bin;    // [frame stride] the inverse map: bin held by output position p, -1 for the padding (round 5: the wave kernel's staged epilogue)
  const int* grp_start;  // [M+1] first bin of group g (bins whose first filter is g), fast path only
  int f_lo, f_hi;        // bins with a non-zero filterbank row: [f_lo, f_hi)
  int nnz;
  int fast_ok;           // 0: general kernel; 1: group formulation with <8, 24> bins per thread; 2: per-wave budgets (default set) fit too;
                         // 3: only the wide per-wave set fits; 5: the line-form group kernel's set (long groups as lines)
  int unit_form;         // 1: in every long group (the top 256) a bin's two weights sum to one (to 1e-6) - the last group, whose second
                         // filter does not exist, carries w1 == 0: the per-wave kernels compute the gradient as d1 + (d0 - d1) w0
  const float* lin;      // [4][M] a0 | s0 | a1 | s1: within group g the weights are w0 = a0 + s0 i, w1 = a1 + s1 i for the group's i-th bin
                         // (triangular filters on a uniform bin grid; fitted and checked to 1e-6 per bin at plan creation)
  int line_from;         // groups below this index are NOT lines (group 0 of a bank whose first filter rises over several bins): the
                         // line-form group kernel keeps such a long group in its thread's table-form slot
  int wave_ok;           // 1: imel_wave_kernel serves this bank (M == 512, the chunk budgets fit, the weights are linear per group)
};
struct ImelArgs {
  ImelTables tb;
  const float* mel;      // [B][M][T]
  const float* spec0;    // optional [B][T][n_stft] injected init (reference layout), else seeded RNG
  float* out_slots;      // [B*T][kFrameStride]
  const float* clip_scale;  // [nclips][2] {2^-e, 2^e}: the power of two the clip's SGD state is held in (range_scale_kernel), null = 2^-60
  float sc, un;             // the frame's pair, set inside the kernels (imel_set_scale)
  float* loss_hist;      // [B*T][max_iter] per-frame sum_m diff^2 before each step
  const int* it_limit;   // optional [nclips] number of steps to run (fix-up pass), NULL = max_iter
  int B, M, T, C;        // C = channels per clip (the loss mean couples them)
  int n_stft;            // linear bins per frame (8821 on the specialised geometry)
  int out_stride;        // elements between output frames (kFrameStride, or the generic path's fs)
  int plain;             // 1: output frames are plain bin-ordered rows (generic path), 0: slot layout
  int max_iter;
  float lr, momentum;
  unsigned long long seed;
  unsigned long long frame_base;  // rfx_call_options::row_base * T: frame f of this call draws its start from key (seed, frame_base + f)
};
hipError_t launch_imel(const ImelArgs& a, int variant, hipStream_t stream);  // variant: 0 best, 1 uniform groups, 2 general
// the kernel launch_imel runs: 4 wave, 5 line-form groups, 2 / 3 per-wave group budgets, 1 uniform groups, 0 general LDS kernel
// (every one but 0 leaves through imel_emit_frame: its output order is the pos_bin table)
int imel_kernel_choice(const ImelTables& tb, int M, int max_iter, int variant);
// scans loss_hist for the early-stop condition; it_stop[clip] = steps the reference would have run
// Numeric range (round 6, include/rfx.h): one workgroup per group of `count` contiguous floats of x (a clip's mel amplitudes, or a
// row's magnitudes) takes max |x| - or `hint` when > 0, without reading x - and writes the powers of two the kernels work in:
//   imel_scale[g] = {2^-e, 2^e}, e = max(k + 35, 30) for max in [2^(k-1), 2^k)     (nullable)
//   gl_scale[g * rows + r] = {2^-j, eps^2}, j = ks - 26, ks = k, or max(k, 0) + 1 when `mel_units` (the SGD's untouched bins keep
//   their U[0,1) start: a row's magnitudes reach 1 whatever the mel amplitudes are)                         (nullable)
// the arithmetic of it, shared by the kernel and rfx_debug_range_exponents (tests without a GPU): k with mx in [2^(k-1), 2^k)
// (0 for mx == 0 or NaN, 129 for +Inf) -> the SGD exponent e and the Griffin-Lim exponent j
RFX_HD void range_exponents(int k, int mel_units, int* e, int* j) {
  int ee = k + 35;
  *e = ee < 30 ? 30 : (ee > 126 ? 126 : ee);
  const int ks = mel_units ? (k > 0 ? k : 0) + 1 : k;
  int jj = ks - 26;
  *j = jj < -100 ? -100 : (jj > 100 ? 100 : jj);
}
// keys: [groups] scratch words
hipError_t launch_range_scale(const float* x, size_t count, int groups, float hint, unsigned* keys, float* imel_scale, float* gl_scale, int rows,
                              int mel_units, hipStream_t stream);
hipError_t launch_imel_scan(const float* loss_hist, int* it_stop, int* any_early, int nclips, int C, int T, int max_iter,
                            float tol_loss, float tol_change, hipStream_t stream);

// ---- generic-geometry path (rfx_generic.hip): any n_fft / win_length / hop_length, plain [B*T][fs] frame arrays
struct GenTables {
  const cf* lo;      // [128]   exp(-2 pi i t / nc)
  const cf* hi;      // [nhi]   exp(-2 pi i 128 t / nc)
  const cf* lo2;     // [128]   exp(-2 pi i t / n_fft)
  const cf* hi2;     // [nhi2]  exp(-2 pi i 128 t / n_fft)
  const float* win;  // [win]
  const int* rev;    // [nc] LDS position (padding included) of element k after the in-place forward passes (digit reversal)
  const cf* tw;      // exact per-pass twiddles W_L^{i p} (gen_tw_table_offset / gen_ip_compute), all passes back to back
};
struct GenStftArgs {
  GenGeom g;
  GenTables tb;
  const float* wave;   // [B][wave_stride], Lw valid samples each
  size_t wave_stride;
  float* mag;          // mode 0: [B*T][fs] |X|
  cf* spec;            // mode 1: [B*T][fs] X
  int B, T, Lw;
};
// one Griffin-Lim iteration of the generic engine, frame by frame: analysis of x_k - m x_{k-1}, projection, synthesis
struct GenGlArgs {
  GenGeom g;
  GenTables tb;
  const float* S;        // [B*T][fs] magnitudes
  const cf* angles0;     // mode 0: optional injected initial angles [B*T][fs] (drawn from `seed` when null)
  const float* x_cur;    // modes 1, 2: the signal to analyse, d = x_k - m x_{k-1} (x_0 in the first iteration), formed by the fold of
                         // the previous iteration (launch_gen_fold's `dout`): [B][audio_stride], L valid samples per clip
  const float* x_prev;   // unused since round 4 (the kernel used to form d itself from x_k and x_{k-1})
  size_t audio_stride;
  float* frames;         // [B*T][win] windowed, scaled synthesis frames (gen_fold_kernel overlap-adds them)
  const float* row_scale;  // [B][2] or null (see GlArgs): the fold applies the factor to d, the kernel takes eps^2 from it
  float mom;             // momentum / (1 + momentum)
  unsigned long long seed;
  unsigned long long frame_base;  // as GlArgs::frame_base
  int B, T, L;
};
hipError_t prepare_generic_kernels(const GenGeom& g);
size_t gen_lds_bytes(const GenGeom& g);
hipError_t launch_gen_stft(int mode, const GenStftArgs& a, int num_cus, hipStream_t stream);  // mode 0 mag, 1 spec
hipError_t launch_gen_gl(int mode, const GenGlArgs& a, int num_cus, hipStream_t stream);      // mode 0 init, 1 first iteration, 2 iteration
hipError_t launch_gen_env(const float* win, float* env, const GenGeom& g, int T, int L, hipStream_t stream);  // env[p] = sum_t w[j]^2, once per call
hipError_t launch_gen_fold(const float* frames, const float* env, float* out, const GenGeom& g, int B, int T, int L, size_t out_stride,
                           hipStream_t stream, const float* prev = nullptr, float* dout = nullptr, float mom = 0.f,
                           const float* row_scale = nullptr);  // L output samples per clip; dout = (x - mom prev) * row_scale[2 b]
hipError_t launch_gen_pack(const void* bft, void* frames, bool complex_, int B, int F, int T, int fs, hipStream_t stream);
hipError_t launch_gen_unpack(const void* frames, void* bft, bool complex_, int B, int F, int T, int fs, hipStream_t stream);
hipError_t launch_gen_mel(const float* mag, float* mel_tm, const float* band_wt, const int* band_lo, const int* band_len, long long nframes,
                          int fs, int M, int Mpad, int f_lo, int f_hi, hipStream_t stream);  // [f_lo, f_hi): bins with a non-zero filterbank row
hipError_t launch_mel_transpose(const float* mel_tm, float* mel, int B, int T, int M, int Mpad, hipStream_t stream);

// ---- row-family Griffin-Lim (rfx_fam.hip): n_fft = 40 h, win_length = 10 h; frames are folded by launch_gen_fold
struct FamGlArgs {
  FamGeom g;
  const float* S;        // [B*T][g.fsf] magnitudes in slot order (launch_fam_repack)
  const cf* angles0;     // mode 0: optional injected initial angles in the plan's PLAIN layout [B*T][fs_plain] (drawn from `seed` when null)
  int fs_plain;
  const float* x_cur;    // modes 1, 2: x_k      [B][audio_stride], L valid samples per clip
  const float* x_prev;   // mode 2:     x_{k-1}
  size_t audio_stride;
  float* frames;         // [B*T][fpitch] windowed, scaled synthesis frames, window sample j at fshift + j (GenGeom::fpitch / fshift)
  const float* row_scale;  // [B][2] or null (see GlArgs)
  int fpitch, fshift;
  const cf* tw1;         // [21][h]      g(n')^k1
  const cf* twa;         // [ra-1][rb]   W_h^{i p} at [p - 1][i]: the lanes of a wave (consecutive i) read consecutive entries (round 5; [rb][ra-1] until
                         // then: 29 cache lines per wave-wide load, which cost the 48 kHz kernels a fifth of their time)
  const float* win;      // [win]
  float mom;             // momentum / (1 + momentum)
  unsigned long long seed;
  unsigned long long frame_base;  // as GlArgs::frame_base
  int B, T, L;
};
// forward STFT of the family geometries into the plan's plain layout (mode 0: |X| floats, mode 1: X complex), or (mode 2) fused
// with the banded mel projection: |X| never leaves the chip, the frame-major mel amplitudes do (spectrogram_converter.py:165-185)
struct FamFwdArgs {
  FamGeom g;
  const float* wave;     // [B][wave_stride], Lw valid samples each
  size_t wave_stride;
  int Lw;
  float* mag;            // mode 0: [B*T][fs_plain]
  cf* spec;              // mode 1: [B*T][fs_plain]
  int fs_plain;
  const cf* tw1;
  const cf* twa;
  const float* win;
  int B, T;
  // mode 2: the band tables of gen_mel_kernel (weights transposed [rows][Mpad], rows a multiple of eight, zero past a filter's end)
  float* mel_tm;         // [B*T][Mpad] frame-major mel amplitudes (mel_transpose_kernel turns them into (B, M, T))
  const float* band_wt;
  const int* band_lo;
  const int* band_len;
  int M, Mpad;
};
hipError_t launch_fam_fwd(int mode, const FamFwdArgs& a, int nblocks, hipStream_t stream);
hipError_t prepare_fam_kernels(const FamGeom& g);
size_t fam_lds_bytes(const FamGeom& g);         // dynamic: the cube
size_t fam_static_lds_bytes(const FamGeom& g);  // static: the pass-A twiddles where they fit
int fam_blocks_per_cu(const FamGeom& g);
bool fam_row_stride_even(const FamGeom& g);     // the kernels use 16-byte LDS accesses in pass B: rows must start 16-byte aligned
hipError_t launch_fam_gl(int mode, const FamGlArgs& a, int nblocks, hipStream_t stream);  // mode 0 init, 1 first iteration, 2 iteration
hipError_t launch_fam_repack(const float* plain, float* slots, const int* bin_of, long long nframes, int fs_plain, int fsf, int n_stft,
                             hipStream_t stream);

// image / PCM codecs
hipError_t launch_image_decode(const uint8_t* img, const float* lut, float* out, int N, int H, int W, int C, hipStream_t s);
// order-preserving integer key of a float for atomicMax: positive NaN is the largest key (np.max's NaN propagation comes for free),
// key 0 is below every value
__device__ __forceinline__ unsigned max_key(float v) {
  unsigned b = __float_as_uint(v);
  if (v != v) b = 0x7FC00000u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
hipError_t launch_clip_max(const float* x, float* out, int nclips, size_t count, bool abs_value, hipStream_t s);
hipError_t launch_image_encode(const float* mel, const float* clip_max, const float* thr, uint8_t* img, int N, int M, int T,
                               int C, hipStream_t s);
// encode straight from the forward kernels' frame-major mel amplitudes [N*C][T][Mpad] (rfx_image_from_waveform): the transpose to the
// image's (mel, time) order happens in LDS; the maximum comes as a key (max_keys, from the forward kernel) or as a float (clip_max)
hipError_t launch_image_encode_tm(const float* mel_tm, const unsigned* max_keys, int keys_per_image, const float* clip_max_in, const float* thr,
                                  uint8_t* img, float* clip_max_out, int N, int M, int Mpad, int T, int C, hipStream_t s);
hipError_t launch_pcm16(const float* wave, const float* clip_peak, int16_t* pcm, int N, int L, int C, int normalize, hipStream_t s);

}  // namespace rfx
