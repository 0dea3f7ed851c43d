/*
 * rfx.h - C ABI of librfx.so: the gfx950 (MI355X) implementation of riffusion's
 * spectrogram <-> audio hot path.
 *
 * The reference (riffusion-hobby @ v0.3.1, pure Python) has no FFI for this path: its boundary is
 * the Python class riffusion/spectrogram_converter.py:12-204 whose arithmetic members are four
 * torchaudio modules.  Each entry point below replaces the torchaudio call named in its comment;
 * INTEGRATION.md shows the ctypes binding a maintainer adds under that class.
 *
 * Conventions: every function returns 0 on success and a negative rfx_status on failure;
 * rfx_last_error() returns a thread-local message.  All pointers named d_* are DEVICE pointers
 * (e.g. torch.Tensor.data_ptr()) owned by the caller; `stream` is a hipStream_t passed as void*
 * (torch.cuda.current_stream().cuda_stream).  No entry point allocates device memory except
 * rfx_plan_create; scratch space is a caller-provided workspace sized by the *_workspace_bytes
 * queries.  Plans are immutable after creation and may be shared between threads.
 *
 * Layouts: "BFT" is the reference's (batch, n_stft, frames) tensor layout; "slots" is this
 * library's frame-major stream layout: [batch*frames][rfx_frame_stride()] with every one-sided
 * bin stored at the position its owning thread streams it from (440 bins are stored twice).
 */
#ifndef RFX_H_
#define RFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rfx_plan rfx_plan;

typedef enum {
  RFX_OK = 0,
  RFX_ERR_INVALID = -1,     /* bad argument / unsupported geometry */
  RFX_ERR_HIP = -2,         /* a HIP runtime call failed */
  RFX_ERR_WORKSPACE = -3,   /* workspace too small */
  RFX_ERR_UNSUPPORTED = -4  /* FFT length with a prime factor above 13 or whose frame buffer exceeds the 160 KiB of LDS (n_fft above
                               about 39 000); non-banded filterbank */
} rfx_status;

/* Mirrors the fields of riffusion/spectrogram_params.py:21-42 that the arithmetic depends on,
 * already resolved to samples (spectrogram_params.py:62-81). */
typedef struct {
  int32_t sample_rate;
  int32_t n_fft;        /* 17640 at 44.1 kHz: that geometry (win 4410, hop 441) runs on the specialised engine;        */
  int32_t win_length;   /* 4410     n_fft = 40 h, win = 10 h with h in {80 .. 480} (48 kHz: 19200 / 4800 / 480; 32, 24, 16, 8 kHz) and     */
  int32_t hop_length;   /* 441      n_fft = 20 h', win = 5 h' (22.05 kHz: 8820 / 2205 / 220) on the row-family kernels;    */
                        /*          every other geometry (11.025 kHz, 96 kHz, custom durations) on the generic engine      */
  int32_t n_mels;       /* num_frequencies */
  int32_t max_mel_iters;
} rfx_params;

const char* rfx_last_error(void);
int rfx_version(void);
/* number of elements (complex or float) between consecutive frames of a slot-major array */
int rfx_frame_stride(void);   /* of the default 44.1 kHz geometry; per plan: rfx_plan_frame_stride */
int rfx_num_bins(void);
/* frame stride of THIS plan's slot arrays (generic-geometry plans store plain bin-ordered frames, n_stft rounded up to 64) */
int rfx_plan_frame_stride(const rfx_plan* plan);
/* 1 when the plan is a generic-geometry plan (any geometry but 17640 / 4410 / 441): plain bin-ordered frames, and either the
 * row-family kernels (n_fft = 40 h, win = 10 h: the specialised engine's factorisation, 1.7x its per-tile time at 48 kHz) or the
 * generic engine - an in-place mixed-radix FFT (digits up to 16, exact per-pass twiddles) in LDS - for Griffin-Lim and the
 * forward STFT (rfx_plan_griffinlim_engine says which); Griffin-Lim is fused per frame with the momentum applied in the time
 * domain, two launches per iteration, on both */
int rfx_plan_is_generic(const rfx_plan* plan);

/* Builds the device constants that spectrogram_converter.py:47-99 builds as torchaudio module
 * buffers: the periodic Hann window (h_window: win_length floats, as torch.hann_window gives it)
 * and the mel filterbank (h_melfb: n_stft x n_mels floats, torchaudio functional.melscale_fbanks;
 * may be NULL when no mel entry point will be used). */
int rfx_plan_create(const rfx_params* params, const float* h_window, const float* h_melfb, int device,
                    rfx_plan** out_plan);
int rfx_plan_destroy(rfx_plan* plan);

/* Which of the two Griffin-Lim device forms of the specialised engine a call takes (see rfx_griffinlim below). */
typedef enum {
  RFX_GL_FORM_AUTO = 0,   /* per call, from B*T: frames for the few-tiles-per-request case, runs for batches */
  RFX_GL_FORM_RUNS = 1,   /* always the run-based fused kernel (one launch per iteration) */
  RFX_GL_FORM_FRAMES = 2  /* always the per-frame kernel + fold (two launches per iteration) */
} rfx_gl_form;

/* Which frame engine Griffin-Lim runs on for geometries other than 17640 / 4410 / 441 */
typedef enum {
  RFX_ENGINE_AUTO = 0,    /* the row-family kernels where n_fft = 40 h and win_length = 10 h with h in {80, 160, 240, 320, 441, 480}
                             (the default 400 / 100 ms at 8, 16, 24, 32, 44.1, 48 kHz; any hop), the generic FFT engine otherwise */
  RFX_ENGINE_GENERIC = 1  /* always the generic FFT engine (cross-checks) */
} rfx_frame_engine;

/* Which plan (layouts + kernels) a parameter set gets */
typedef enum {
  RFX_LAYOUT_AUTO = 0,    /* 17640 / 4410 / 441 (the reference's geometry at 44.1 kHz): the specialised engine and its slot-major
                             frames; every other geometry: the generic plan (plain bin-ordered frames) */
  RFX_LAYOUT_GENERIC = 1  /* the generic plan also for 17640 / 4410 / 441: a second, independent implementation of the same
                             transform (row family with h = 441, or the generic FFT engine with RFX_ENGINE_GENERIC) - cross-checks */
} rfx_plan_layout;

/* Which InverseMelScale kernel family a plan may select (rfx_plan_imel_kernel reports the choice) */
typedef enum {
  RFX_IMEL_FORM_AUTO = 0,   /* the wave kernel (one wave per frame) where the bank admits it, else the group kernels */
  RFX_IMEL_FORM_GROUPS = 1  /* never the wave kernel: the group kernels (one workgroup of four waves per frame) - cross-checks */
} rfx_imel_form;

/* Plan-creation options.  Set struct_size = sizeof(rfx_plan_options); zero in every other field means "default".
 * This struct is the library's ONLY configuration surface: a release build reads no environment variable (the RFX_*
 * experiment switches of the source exist only in builds made with -DRFX_ABLATION, tools/build_variants.sh). */
typedef struct {
  uint32_t struct_size;
  int32_t gl_form;             /* rfx_gl_form */
  int32_t gl_frames_per_slot;  /* RFX_GL_FORM_AUTO takes the per-frame form up to this many frames per resident
                                  workgroup slot of the chip (0 = default, 6: the measured crossover - six 512-frame tiles per call;
                                  4 until round 6) */
  int32_t frame_engine;        /* rfx_frame_engine */
  int32_t plan_layout;         /* rfx_plan_layout; added in round 4 - a caller built against the shorter struct gets AUTO */
  int32_t imel_form;           /* rfx_imel_form; added in round 4 after plan_layout, same rule */
} rfx_plan_options;

/* rfx_plan_create with options (NULL = defaults = rfx_plan_create). */
int rfx_plan_create_ex(const rfx_params* params, const float* h_window, const float* h_melfb, int device,
                       const rfx_plan_options* options, rfx_plan** out_plan);
/* the engine rfx_griffinlim runs on for this plan: 0 = specialised (17640 / 4410 / 441), 1 = generic FFT engine, 2 = row family */
int rfx_plan_griffinlim_engine(const rfx_plan* plan);
/* the form (RFX_GL_FORM_RUNS / _FRAMES) an rfx_griffinlim call of B x T frames takes on this plan */
int rfx_griffinlim_form(const rfx_plan* plan, int B, int T);
/* How one launch of the run-based form cuts the call's B*T frames (counted clip after clip) into runs, one per workgroup: returns
 * the number of runs and, if run_starts != NULL, writes min(runs + 1, capacity) run boundaries (run b = frames
 * [run_starts[b], run_starts[b + 1])).  which = 0: the first (synthesis-only) launch, 1: the iterations.  0 for a generic plan.
 * No device work: the partition is a function of (chip, B, T) alone - results are bit-reproducible call to call.  (tests) */
int rfx_griffinlim_runs(const rfx_plan* plan, int B, int T, int which, int64_t* run_starts, int capacity);
/* the arithmetic behind it, callable without a plan or a GPU (tests): first frame of run b of `runs` over n_frames frames when
 * the first h runs weigh w1 and the others w2 (per mille of the mean run length) */
int64_t rfx_debug_run_start(int64_t b, int64_t runs, int64_t n_frames, int64_t h, int64_t w1, int64_t w2);
/* the partition rfx_griffinlim_runs reports, for a chip with `slots` resident workgroup slots, without a plan or a GPU (tests):
 * runs are whole groups of 16 frames of a row (round 6), the first N mod runs of them one group longer */
int rfx_debug_gl_partition(int slots, int B, int T, int64_t* run_starts, int capacity);
/* the two internal exponents of "Numeric range" above for a largest magnitude max_abs (mel_units != 0: max_abs is a mel amplitude,
 * the Griffin-Lim exponent is then taken for max(max_abs, 1) x 2), without a GPU (tests) */
int rfx_debug_range_exponents(float max_abs, int mel_units, int* sgd_exponent, int* gl_exponent);
/* frames torch.stft(center=True, pad_mode="reflect") makes of Lw samples: 1 + (Lw + 2*(n_fft/2) - n_fft) / hop, i.e.
 * 1 + Lw/hop for even n_fft and 1 + (Lw-1)/hop for odd n_fft; 0 when Lw <= n_fft/2 (the reference raises there).
 * Every forward entry point below produces exactly this many frames. */
int rfx_stft_frames(const rfx_plan* plan, int Lw);

/* Per-call options of the inverse entry points (round 6; the *_ex forms below; NULL = defaults = the plain forms).
 * Set struct_size = sizeof(rfx_call_options) and zero everything you do not use.
 *
 * row_base: index, in the caller's WHOLE batch, of the first row (clip-channel) this call converts.  The random starts the
 *   reference draws from torch's global generator (spectrogram_converter.py:72 rand_init=True; torchaudio InverseMelScale's
 *   torch.rand) are drawn here from (seed, row_base + r, frame, bin) for row r of the call.  With the same seed, a batch converted
 *   in one call, in chunks (row_base = rows before the chunk) or sharded over the GPUs of a node gives the same audio for every clip,
 *   bit for bit: nothing else in a clip's arithmetic depends on the batch it travels in (csrc/rfx_kernels.h: kGlGroup).
 *   Must be a multiple of channels_per_clip where the entry point has one.
 * magnitude_hint: see "Numeric range" below; 0 = not given.
 *
 * Numeric range of the inverse entry points (rfx_inverse_mel, rfx_griffinlim, rfx_waveform_from_mel, rfx_audio_from_image_u8
 * and their *_ex forms).  The reference scales a decoded image by a caller-chosen `max_value` (image_util.py:59-108, default 30e6
 * at spectrogram_image_converter.py:69) and runs torchaudio in plain float32; so does this library, with two internal powers of
 * two that keep its fast paths inside float32 whatever the units are:
 *   - InverseMelScale holds the SGD state of a clip times 2^-e, e = max(k + 35, 30) for a largest mel amplitude in [2^(k-1), 2^k)
 *     (e = 60 at max_value 30e6): the `clamp(min=0)` of every step is then the output clamp of the FMA that makes it;
 *   - Griffin-Lim analyses a row's signal times 2^-j, j = k' - 26 for a largest magnitude in [2^(k'-1), 2^k') (j = 0 at 30e6),
 *     so that |a|^2 in a / (|a| + 1e-16) can neither overflow nor vanish.
 * Both commute with every float32 rounding of the linear steps: the results are those of the unscaled arithmetic, and an input
 * times 2^n gives the output times 2^n bit for bit (tests/test_gpu_round6_range.py).  The exponents are taken PER CLIP / PER ROW from
 * the data by one small reduction launch - or from magnitude_hint > 0, an upper bound of the call's magnitudes in the input's
 * units (the image path passes max_value), without reading the data.  A hint below the data is the caller's error: magnitudes
 * above 2^35 x hint saturate the SGD state.
 * Supported, and held against the oracle at 1e-6, 1, 30e6, 1e12, 1e20 (Griffin-Lim also 1e30): magnitudes from 0 up to 2^91 = 2.5e27
 * (InverseMelScale; above it the headroom of the clamp shrinks, at 2^126 it is gone) and 1e33 (Griffin-Lim: float32 itself ends
 * where 8821 bins of that size add up).  Not supported: NaN / Inf magnitudes (garbage in, garbage out, as in the reference).
 * Where the reference itself stops being scale-free, this library follows the reference, not the scale: its stopping rule is
 * absolute (loss < 1e-5, |change| < 1e-8: tiny spectrograms stop after one step - reproduced), the bins no filter reaches keep
 * their U[0,1) start whatever max_value is (reproduced), and below |a| = 1e-8 in the input's units the `+ 1e-16` guard becomes
 * visible: there this library's guard, rsq(|a|^2 + 1e-32), differs from the reference's 1 / (|a| + 1e-16) by up to 41 %
 * (at |a| = 1e-16) in the LENGTH of the phase factor - never in its direction, and it is exactly 0 for a = 0 in both. */
typedef struct {
  uint32_t struct_size;
  uint32_t flags;           /* must be 0 */
  uint64_t row_base;
  float magnitude_hint;
  float reserved;           /* must be 0 */
} rfx_call_options;

/* ---- layout converters ------------------------------------------------------------------- */
/* (B, n_stft, T) float32 magnitudes -> slots (float32) */
int rfx_pack_magnitudes(const rfx_plan* plan, const float* d_lin_bft, int B, int T, float* d_slots, void* stream);
/* (B, n_stft, T) complex64 -> slots (complex64); conjugate slots are conjugated */
int rfx_pack_complex(const rfx_plan* plan, const void* d_bft, int B, int T, void* d_slots, void* stream);
/* slots (complex64) -> (B, n_stft, T) complex64 */
int rfx_unpack_complex(const rfx_plan* plan, const void* d_slots, int B, int T, void* d_bft, void* stream);

/* ---- forward: torchaudio.transforms.Spectrogram(power=None) [+ torch.abs] ------------------
 * spectrogram_converter.py:179 (+ :182).  d_wave: (B, Lw) float32, Lw > n_fft/2.
 * T = rfx_stft_frames(plan, Lw).  Either output may be NULL. */
int rfx_stft(const rfx_plan* plan, const float* d_wave, int B, int Lw, float* d_mag_slots, void* d_spec_slots,
             void* stream);

/* ---- inverse: torchaudio.transforms.GriffinLim(n_iter, momentum=0.99, rand_init=True, power=1)
 * spectrogram_converter.py:62-73, called at :204.
 * d_mag_slots: magnitudes in slot layout; d_angles0_slots: optional injected initial angles
 * (NULL = draw U[0,1) real/imag per bin from `seed`); d_wave_out: (B, rfx_griffinlim_output_samples(plan, T)) float32. */
/* Two device forms, chosen per call from B*T: the run-based fused kernel (one launch per iteration) for batches, and a
 * per-frame kernel + fold (two launches per iteration, every frame its own workgroup) for the few-tiles-per-request case;
 * the workspace query below accounts for whichever the shape will take. */
size_t rfx_griffinlim_workspace_bytes(const rfx_plan* plan, int B, int T);
/* samples per clip rfx_griffinlim writes for T frames: what torch.istft(center=True, length=None) returns,
 * hop*(T-1), plus one when n_fft is odd */
int rfx_griffinlim_output_samples(const rfx_plan* plan, int T);
int rfx_griffinlim(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                   int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                   void* stream);

/* rfx_griffinlim with per-call options (NULL = rfx_griffinlim) and, when h_launch_ms != NULL, rfx_griffinlim_timed's
 * per-launch durations (it then synchronises the stream). */
int rfx_griffinlim_ex(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                      int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                      void* stream, const rfx_call_options* options, float* h_launch_ms);

/* Same as rfx_griffinlim, but brackets every kernel launch with HIP events recorded on `stream`
 * and, after synchronising, writes the n_iter+1 launch durations (ms; [0] = the init ISTFT, [1] the
 * first iteration, [2..] the steady-state iterations) to the HOST array h_launch_ms.  Measurement
 * aid for bench.py's roofline figure; it synchronises the stream. */
int rfx_griffinlim_timed(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                         int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                         void* stream, float* h_launch_ms);

/* slots (float32) -> (B, n_stft, T) float32 */
int rfx_unpack_magnitudes(const rfx_plan* plan, const float* d_slots, int B, int T, float* d_bft, void* stream);

/* ---- forward: mel_amplitudes_from_waveform, spectrogram_converter.py:165-185
 * Spectrogram(power=None) -> torch.abs -> MelScale (matmul with the filterbank, on the fp32 MFMA).
 * d_wave (B, Lw) float32 -> d_mel_out (B, n_mels, T) float32, T = rfx_stft_frames(plan, Lw). */
size_t rfx_mel_workspace_bytes(const rfx_plan* plan, int B, int Lw);
int rfx_mel_from_waveform(const rfx_plan* plan, const float* d_wave, int B, int Lw, float* d_mel_out, void* d_workspace,
                          size_t workspace_bytes, void* stream);

/* ---- forward, all the way to the image: SpectrogramImageConverter.spectrogram_image_from_audio's device half
 * (spectrogram_image_converter.py:30-51: spectrogram_from_audio, then image_util.image_from_spectrogram, image_util.py:27-54).
 * d_wave (N*C, Lw) float32 (C = 2 when stereo: the channels of clip n are rows 2n, 2n+1) -> d_img_out (N, n_mels, T, 3) uint8 and
 * d_clip_max (N floats: the EXIF MAX_VALUE, spectrogram_image_converter.py:45-49).  Byte-identical to rfx_mel_from_waveform
 * followed by rfx_image_encode_u8; the (N*C, n_mels, T) tensor is never written: the maximum is taken while the mel amplitudes
 * are formed and the encoder reads the forward kernel's frame-major scratch. */
size_t rfx_image_from_waveform_workspace_bytes(const rfx_plan* plan, int N, int stereo, int Lw);
int rfx_image_from_waveform(const rfx_plan* plan, const float* d_wave, int N, int stereo, int Lw, const float* d_thresholds255,
                            float* d_clip_max, uint8_t* d_img_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* Standalone torchaudio.transforms.MelScale.forward (spectrogram_converter.py:185) for callers that hold linear
 * magnitudes in the reference's (B, n_stft, T) layout: packs them into slots and runs the same MFMA projection.
 * Workspace: rfx_mel_scale_workspace_bytes. */
size_t rfx_mel_scale_workspace_bytes(const rfx_plan* plan, int B, int T);
int rfx_mel_scale(const rfx_plan* plan, const float* d_lin_bft, int B, int T, float* d_mel_out, void* d_workspace,
                  size_t workspace_bytes, void* stream);

/* ---- inverse: torchaudio.transforms.InverseMelScale (SGD, max_iter = params.max_mel_iters,
 * tolerance_loss 1e-5, tolerance_change 1e-8, lr 0.1, momentum 0.9), spectrogram_converter.py:87-99,
 * called at :201.  d_mel (B, n_mels, T); the B rows are grouped into clips of `channels_per_clip`
 * consecutive rows, each clip being one call of the reference (its loss mean couples the clip's
 * channels and frames).  d_spec0: optional injected start (B, T, n_stft) float32 in the reference's
 * own layout, NULL = U[0,1) from `seed`.  Output: linear magnitudes in slot layout, ready for
 * rfx_griffinlim. */
/* which SGD kernel rfx_inverse_mel runs for this plan's filterbank: 4 = wave kernel (one wave per frame, weights as a line per
 * group: the default 512-filter HTK bank, with or without slaney normalisation), 2 = group kernel with per-wave register budgets sized to
 * the default 512-filter HTK bank (with or without slaney normalisation), 3 = the same kernel with the wider budget set
 * (mel_scale_type "slaney"), 5 = line-form group kernel (round 5: banks with groups of up to 62 bins whose long groups are lines -
 * 512 filters up to the Nyquist frequency, e.g. the 20 Hz .. 20 kHz of the reference's test/spectrogram_converter_test.py:46-53,
 * or 256 / 384 filters over 0 - 10 kHz), 1 = group kernel with a uniform budget (other banks whose groups fit 8 / 24 bins),
 * 0 = general LDS kernel (any banded bank), -1 = not banded (rfx_inverse_mel refuses) */
int rfx_plan_imel_kernel(const rfx_plan* plan);
/* 1 when that kernel (2, 3, 4 or 5) computes a bin's gradient in unit form, d1 + (d0 - d1) w0: valid when the two weights of every
 * bin of the long groups sum to one, i.e. triangular filters without area normalisation (mel_scale_norm None); 0 = both
 * weights are multiplied out (mel_scale_norm "slaney", other kernels) */
int rfx_plan_imel_unit_form(const rfx_plan* plan);
size_t rfx_inverse_mel_workspace_bytes(const rfx_plan* plan, int B, int T);
int rfx_inverse_mel(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, const float* d_spec0,
                    uint64_t seed, float* d_mag_slots, void* d_workspace, size_t workspace_bytes, void* stream);

int rfx_inverse_mel_ex(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, const float* d_spec0,
                       uint64_t seed, float* d_mag_slots, void* d_workspace, size_t workspace_bytes, void* stream,
                       const rfx_call_options* options);

/* ---- inverse, in one call: SpectrogramConverter.waveform_from_mel_amplitudes, spectrogram_converter.py:187-204
 * (`self.inverse_mel_scaler(amplitudes_mel)` :201 then `self.inverse_spectrogram_func(amplitudes_linear)` :204).
 * d_mel (B, n_mels, T) -> d_wave_out (B, rfx_griffinlim_output_samples(plan, T)); clips of `channels_per_clip` rows as in
 * rfx_inverse_mel; both random starts from `seed` (the SGD start from seed, the phases from seed + 1).  Exactly rfx_inverse_mel
 * followed by rfx_griffinlim - same bits - with the linear magnitudes kept inside the workspace. */
size_t rfx_waveform_from_mel_workspace_bytes(const rfx_plan* plan, int B, int T);
int rfx_waveform_from_mel(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, uint64_t seed, int n_iter,
                          float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, void* stream);

int rfx_waveform_from_mel_ex(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, uint64_t seed, int n_iter,
                             float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, void* stream,
                             const rfx_call_options* options);

/* ---- image codec: riffusion/util/image_util.py -----------------------------------------------
 * decode = spectrogram_from_image (:81-108): d_img (N, H, W, 3) uint8 RGB -> (N*C, H, W) float32,
 *   C = 2 (G,B planes) when stereo else 1 (R plane); d_lut256[p] is the float32 value numpy's chain
 *   255-p, /255, **(1/power), *max_value gives for pixel value p (built on the host WITH numpy).
 * encode = image_from_spectrogram (:27-54): d_mel (N*C, M, T) float32 -> d_img_out (N, M, T, 3) uint8;
 *   the per-clip maximum is written to d_clip_max (N floats; it is the EXIF MAX_VALUE of
 *   spectrogram_image_converter.py:59); d_thresholds255[v] = smallest float32 ratio x/max whose
 *   numpy result is <= v (descending, built on the host WITH numpy). */
int rfx_image_decode_u8(const uint8_t* d_img, int N, int H, int W, int stereo, const float* d_lut256, float* d_mel_out,
                        void* stream);
int rfx_image_encode_u8(const float* d_mel, int N, int M, int T, int stereo, const float* d_thresholds255, float* d_clip_max,
                        uint8_t* d_img_out, void* stream);

/* ---- PCM tail: riffusion/util/audio_util.py:22-28.  d_wave (N*C, L) float32 -> d_pcm_out (N, L, C)
 * int16: joint peak normalisation over a clip's channels (when normalize != 0), truncation toward
 * zero.  d_clip_peak (N floats) receives max|x| per clip. */
int rfx_pcm16(const float* d_wave, int N, int C, int L, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* stream);

/* ---- inverse, all the way from the image: SpectrogramImageConverter.audio_from_spectrogram_image's device half
 * (spectrogram_image_converter.py:54-91: image_util.spectrogram_from_image, audio_from_spectrogram -> waveform_from_mel_amplitudes
 * on the image's (C, n_mels, T) tensor, audio_util.audio_from_waveform).  d_img (N, n_mels, T, 3) uint8 -> d_pcm_out (N, L, C) int16,
 * L = rfx_griffinlim_output_samples(plan, T); d_clip_peak (N floats) as rfx_pcm16; d_lut256 as rfx_image_decode_u8.  Exactly
 * rfx_image_decode_u8, rfx_waveform_from_mel (clips of C rows, `seed`), rfx_pcm16 - same bytes - with the tensors in between
 * kept inside the workspace. */
size_t rfx_audio_from_image_workspace_bytes(const rfx_plan* plan, int N, int stereo, int T);
int rfx_audio_from_image_u8(const rfx_plan* plan, const uint8_t* d_img, int N, int T, int stereo, const float* d_lut256, uint64_t seed,
                            int n_iter, float momentum, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* d_workspace,
                            size_t workspace_bytes, void* stream);

/* options->row_base counts ROWS (clip-channels): the call's first image is image row_base / C of the caller's batch */
int rfx_audio_from_image_u8_ex(const rfx_plan* plan, const uint8_t* d_img, int N, int T, int stereo, const float* d_lut256, uint64_t seed,
                               int n_iter, float momentum, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* d_workspace,
                               size_t workspace_bytes, void* stream, const rfx_call_options* options);

#ifdef __cplusplus
}
#endif
#endif /* RFX_H_ */
