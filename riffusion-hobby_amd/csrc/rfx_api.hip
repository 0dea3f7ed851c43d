// rfx_api.hip - the C ABI of librfx.so (include/rfx.h): plan construction and the host-side drivers
// that sequence the gfx950 kernels.  No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/rfx.h"
#include "rfx_kernels.h"

using namespace rfx;

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define RFX_HIP(call)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(RFX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Every entry point runs on the device that owns its plan (or its buffers) and leaves the calling thread's
// current device as it found it: torch tracks the current device per thread, and one process may hold plans
// on several GPUs.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define RFX_ON_DEVICE(dev)   \
  DeviceGuard guard_((dev)); \
  if (guard_.err != hipSuccess) return fail(RFX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

// device that owns a caller buffer (the codec entry points take no plan)
int device_of(const void* d_ptr, int* device) {
  hipPointerAttribute_t attr;
  const hipError_t e = hipPointerGetAttributes(&attr, d_ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(RFX_ERR_INVALID, std::string("not a device pointer: ") + hipGetErrorString(e));
  }
  *device = attr.device;
  return RFX_OK;
}

// HIP events of the timed Griffin-Lim entry point; released on every exit path
struct EventList {
  std::vector<hipEvent_t> ev;
  ~EventList() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
  hipError_t create(size_t n) {
    ev.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      hipEvent_t e;
      const hipError_t rc = hipEventCreate(&e);
      if (rc != hipSuccess) return rc;
      ev.push_back(e);
    }
    return hipSuccess;
  }
};
}  // namespace

#ifndef RFX_FWD_RUN_SKEW
#define RFX_FWD_RUN_SKEW 170  // per mille: 74 / 54 frames instead of 64 / 64; -4.3 % on the forward kernel (profiles/r06_forward_skew.txt)
#endif

struct rfx_plan {
  rfx_params p;
  int device;
  int num_cus;
  int n_stft;
  int gl_wgs_per_cu = 1;    // resident Griffin-Lim workgroups per CU on this device (occupancy query at creation)
  int imel_variant = 0;     // debugging override read once at creation: 0 = best, 1 = uniform groups, 2 = general
  unsigned long long* timing = nullptr;  // RFX_TIMING builds only
  cf* d_tw1 = nullptr;      // [21][441]
  cf* d_tw2 = nullptr;      // [21][21]
  float* d_win = nullptr;   // [4410]
  float* d_melfb_slots = nullptr;  // [kFrameStride][n_mels]: filterbank rows permuted to slot order
  float* d_melfb = nullptr;        // [n_stft][n_mels] as given
  int* d_kblocks = nullptr;        // non-zero 32-position K blocks of d_melfb_slots
  int n_kblocks = 0;
  int melfb_cols = 0;              // columns of d_melfb_slots (n_mels rounded up to 128)
  // banded view of the filterbank for InverseMelScale (valid when imel_ok)
  bool imel_ok = false;
  std::string imel_why;
  ImelTables imel{};
  void* d_imel_blob = nullptr;
  int* d_bin_pos = nullptr;        // [n_stft] primary slot position, [n_stft] duplicate (-1)
  // fused forward path (banded mel projection inside the STFT kernel), valid when fwd_ok
  bool fwd_ok = false;
  float* d_band_wt = nullptr;      // [band_rows][Mpad]
  int* d_band_addr = nullptr;      // [band_rows][Mpad] LDS position of each band bin (specialised engine)
  int* d_band_lo = nullptr;        // [Mpad] followed by band_len [Mpad]
  int band_rows = 0, Mpad = 0;
  bool fwd_unfused = false;        // debugging override (RFX_FWD_UNFUSED), read once at creation
  void* d_slot_tab = nullptr;      // product form of the fused kernel: [21][kQPad] {w0, w1} per slot ...
  int* d_slot_idx = nullptr;       // ... [kMelPadsPerThread][kQPad] padding positions, [2][Mpad] filter segments, [21][kQPad] product positions; null: table form
  unsigned fwd_kb_mask = 0;
  int fwd_prod_arr = 0;
  int fwd_packed_off = 0;          // ints into d_slot_idx where the packed tables start (0: none)
  int fwd_run_skew = RFX_FWD_RUN_SKEW;  // per mille of the run length the first-dispatched workgroups of the forward kernel take on top (RFX_FWD_SKEW in ablation builds)
  int fwd_run_cap = 64;            // longest run of frames one workgroup of the product-form kernel walks (RFX_FWD_RUN, read at creation)
  // generic-geometry path (rfx_generic.hip): everything but n_fft = 17640 / win = 4410 / hop = 441
  bool gl_latency_mode = true;     // small batches use the per-frame Griffin-Lim kernels (RFX_GL_LATENCY_MODE=0 disables)
  int gl_latency_frames_per_slot = 6;  // ... up to this many frames per resident workgroup slot (RFX_GL_LATENCY_FRAMES).  4 until round 6;
                                       // runs are whole groups of 16 frames now, so the run form costs a batch below nine tiles what it costs
                                       // eight (3.4 - 3.7 ms per Griffin-Lim 32) and the per-frame form, linear in the batch, wins up to six
                                       // tiles (3.2 ms): profiles/r06_griffinlim_forms_by_batch.txt
  int gl_form = RFX_GL_FORM_AUTO;      // rfx_plan_options.gl_form
  bool generic = false;
  GenGeom gg{};
  GenTables gt{};
  void* d_gen_tables = nullptr;
  int* d_gen_rev = nullptr;
  cf* d_gen_tw = nullptr;
  int frame_stride = kFrameStride;
  // row-family Griffin-Lim (rfx_fam.hip) on top of a generic plan: n_fft = 40 h, win_length = 10 h
  bool fam_ok = false;
  FamGeom fam{};
  cf* d_fam_tw = nullptr;      // [21][h] g(n')^k1, then [rb][ra-1] W_h^{i p}
  int* d_fam_binof = nullptr;  // [fsf] bin held by each position of the slot-ordered magnitudes (-1: padding)
  int fam_wgs_per_cu = 1;
};

namespace rfx {
__global__ void out_scale_kernel(const float* __restrict__ win, float* __restrict__ out, int T, int L) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= L) return;
  // frames t with 0 <= p - 441 t + 2205 < 4410
  int tlo = (p + 2205 - (kWin - 1) + kHop - 1) / kHop;  // ceil, numerator may be negative
  if (p + 2205 - (kWin - 1) < 0) tlo = 0;
  int thi = (p + 2205) / kHop;
  if (thi > T - 1) thi = T - 1;
  float env = 0.f;
  for (int t = tlo; t <= thi; ++t) {
    const float w = win[p - kHop * t + 2205];
    env = fmaf(w, w, env);
  }
  out[p] = (2.0f / (float)kNfft) / env;
}
}  // namespace rfx

extern "C" {

const char* rfx_last_error(void) { return g_err.c_str(); }
int rfx_version(void) { return 1; }
int rfx_frame_stride(void) { return kFrameStride; }
int rfx_num_bins(void) { return kBins; }
int rfx_plan_frame_stride(const rfx_plan* plan) { return plan ? plan->frame_stride : 0; }
int rfx_plan_is_generic(const rfx_plan* plan) { return plan && plan->generic ? 1 : 0; }
int rfx_plan_griffinlim_engine(const rfx_plan* plan) { return !plan ? -1 : !plan->generic ? 0 : plan->fam_ok ? 2 : 1; }
int rfx_griffinlim_form(const rfx_plan* plan, int B, int T);
int rfx_plan_imel_unit_form(const rfx_plan* plan) {
  if (!plan || !plan->d_melfb || !plan->imel_ok || (plan->imel_variant != 0 && plan->imel_variant != 3)) return 0;
  return plan->imel.fast_ok >= 2 ? plan->imel.unit_form : 0;
}

int rfx_plan_imel_kernel(const rfx_plan* plan) {
  if (!plan || !plan->d_melfb || !plan->imel_ok) return -1;
  return rfx::imel_kernel_choice(plan->imel, plan->p.n_mels, plan->p.max_mel_iters, plan->imel_variant);
}
// torch.stft(center=True): the signal is reflect-padded by n_fft/2 on both sides, so a waveform of Lw samples gives
// 1 + (Lw + 2*(n_fft/2) - n_fft) / hop frames: 1 + Lw/hop for even n_fft, 1 + (Lw - 1)/hop for odd n_fft
static int stft_frames(const rfx_plan* plan, int Lw) {
  return 1 + (Lw + 2 * (plan->p.n_fft / 2) - plan->p.n_fft) / plan->p.hop_length;
}
int rfx_stft_frames(const rfx_plan* plan, int Lw) {
  if (!plan || Lw <= plan->p.n_fft / 2) return 0;
  return stft_frames(plan, Lw);
}
int rfx_griffinlim_output_samples(const rfx_plan* plan, int T) {
  if (!plan || T < 1) return 0;
  return plan->p.hop_length * (T - 1) + (plan->p.n_fft & 1);
}

int rfx_plan_destroy(rfx_plan* plan);

int rfx_plan_create(const rfx_params* params, const float* h_window, const float* h_melfb, int device,
                    rfx_plan** out_plan) {
  return rfx_plan_create_ex(params, h_window, h_melfb, device, nullptr, out_plan);
}

// Experiment switches (RFX_* environment variables) exist only in builds made with -DRFX_ABLATION (tools/build_variants.sh):
// a release build of librfx.so reads no environment variable at all, rfx_plan_options is its only configuration surface.
static inline const char* abl_env(const char* name) {
#ifdef RFX_ABLATION
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

int rfx_plan_create_ex(const rfx_params* params, const float* h_window, const float* h_melfb, int device,
                       const rfx_plan_options* options, rfx_plan** out_plan) {
  if (!params || !out_plan || !h_window) return fail(RFX_ERR_INVALID, "rfx_plan_create: null argument");
  rfx_plan_options opt{};
  opt.struct_size = sizeof(rfx_plan_options);
  if (options) {
    if (options->struct_size < 2 * sizeof(uint32_t) || options->struct_size > sizeof(rfx_plan_options))
      return fail(RFX_ERR_INVALID, "rfx_plan_create_ex: options->struct_size does not describe an rfx_plan_options this library knows");
    memcpy(&opt, options, options->struct_size);
    if (opt.gl_form < RFX_GL_FORM_AUTO || opt.gl_form > RFX_GL_FORM_FRAMES || opt.gl_frames_per_slot < 0)
      return fail(RFX_ERR_INVALID, "rfx_plan_create_ex: gl_form must be RFX_GL_FORM_AUTO / _RUNS / _FRAMES, gl_frames_per_slot >= 0");
    if (opt.frame_engine < RFX_ENGINE_AUTO || opt.frame_engine > RFX_ENGINE_GENERIC)
      return fail(RFX_ERR_INVALID, "rfx_plan_create_ex: frame_engine must be RFX_ENGINE_AUTO or RFX_ENGINE_GENERIC");
    if (opt.plan_layout < RFX_LAYOUT_AUTO || opt.plan_layout > RFX_LAYOUT_GENERIC)
      return fail(RFX_ERR_INVALID, "rfx_plan_create_ex: plan_layout must be RFX_LAYOUT_AUTO or RFX_LAYOUT_GENERIC");
    if (opt.imel_form < RFX_IMEL_FORM_AUTO || opt.imel_form > RFX_IMEL_FORM_GROUPS)
      return fail(RFX_ERR_INVALID, "rfx_plan_create_ex: imel_form must be RFX_IMEL_FORM_AUTO or RFX_IMEL_FORM_GROUPS");
  }
  const bool generic = params->n_fft != kNfft || params->win_length != kWin || params->hop_length != kHop ||
                       opt.plan_layout == RFX_LAYOUT_GENERIC;
  GenGeom gg{};
  if (generic) {
    // any geometry torch.stft accepts (0 < hop, 0 < win <= n_fft) whose FFT length factors into the implemented radices
    if (params->n_fft < 2 || params->hop_length < 1 || params->win_length < 1 || params->win_length > params->n_fft)
      return fail(RFX_ERR_INVALID, "rfx_plan_create: need 0 < hop_length, 0 < win_length <= n_fft");
    gg.n_fft = params->n_fft;
    gg.win = params->win_length;
    gg.hop = params->hop_length;
    gg.n_stft = params->n_fft / 2 + 1;
    gg.even = params->n_fft % 2 == 0;
    gg.nc = gg.even ? params->n_fft / 2 : params->n_fft;
    gg.left = (params->n_fft - params->win_length) / 2;
    gen_frame_layout(gg);
    gg.fs = (gg.n_stft + 63) / 64 * 64;
    gg.nhi = gg.nc / kGenTwLo + 1;
    gg.nhi2 = gg.nc / kGenTwLo + 2;
    if (gg.nc > kGenMaxNc)
      return fail(RFX_ERR_UNSUPPORTED, "rfx_plan_create: n_fft = " + std::to_string(params->n_fft) + ": the frame's FFT buffer (" +
                  std::to_string(gg.nc) + " complex numbers) does not fit the 160 KiB of LDS of a CU");
    if (!gen_factor(gg.nc, gg.radix, &gg.nstages))
      return fail(RFX_ERR_UNSUPPORTED, "rfx_plan_create: FFT length " + std::to_string(gg.nc) + " (from n_fft = " + std::to_string(params->n_fft) +
                  ") has a prime factor above 13; implemented radices: 2, 3, 4, 5, 7, 11, 13");
    // threads per workgroup: measured on MI355X, the engine is latency bound and more waves win over fuller rounds
    // (48 kHz, 64 tiles x 32 iterations: 512 threads 121 ms, 384: 134, 320 - the count gen_pick_threads prefers: 155, 256: 163)
    gg.nthr = 512;
    if (const char* e = abl_env("RFX_GEN_THREADS")) { const int v = atoi(e); if (v >= 64 && v <= 512 && v % 64 == 0) gg.nthr = v; }
    {  // LDS padding: keep as many workgroups per CU as the unpadded buffer allows
      const size_t tables = sizeof(cf) * (2 * (size_t)kGenTwLo + gg.nhi + gg.nhi2);
      const size_t plain = sizeof(cf) * (size_t)gg.nc + tables + 512;
      int per_cu = (int)((160u * 1024u) / plain);
      if (per_cu < 1) per_cu = 1;
      if (per_cu > 1024 / gg.nthr) per_cu = 1024 / gg.nthr;
      const size_t room = (160u * 1024u) / per_cu - tables - 512;
      gg.pad_shift = gen_pick_pad(gg, (int)(room / sizeof(cf)));
      if (const char* e = abl_env("RFX_GEN_PAD")) { const int v = atoi(e); if (v == 0 || (v >= 3 && v <= 8)) gg.pad_shift = v; }
      if (gen_lds_bytes(gg) > 160u * 1024u) gg.pad_shift = 0;
      if (gen_lds_bytes(gg) > 160u * 1024u)
        return fail(RFX_ERR_UNSUPPORTED, "rfx_plan_create: n_fft = " + std::to_string(params->n_fft) + ": the frame's FFT buffer and twiddle tables (" +
                    std::to_string(gen_lds_bytes(gg)) + " bytes) do not fit the 160 KiB of LDS of a CU (largest supported: n_fft about 39000 when "
                    "even, 19500 when odd)");
    }
  }
  // Griffin-Lim of the geometries with n_fft = 40 h, win_length = 10 h (the default 400 / 100 ms at 48 / 32 / 24 / 16 / 8 kHz)
  // runs on the row-family kernels; the generic engine keeps everything else of the plan (layouts, forward path)
  FamGeom fam{};
  bool fam_ok = generic && opt.frame_engine != RFX_ENGINE_GENERIC &&
                fam_make_geom(params->n_fft, params->win_length, params->hop_length, &fam);
  if (fam_ok) {
    // pad the rows by up to seven elements (bank spread of the row-to-row accesses) as long as that costs no resident workgroup
    const size_t plain = fam_lds_bytes(fam) + fam_static_lds_bytes(fam);
    int per_cu = (int)((160u * 1024u) / plain);
    if (per_cu > 1024 / fam.nthr) per_cu = 1024 / fam.nthr;
    if (per_cu < 1) fam_ok = false;
    for (int pad = 7; fam_ok && pad > 0; --pad) {
      FamGeom t = fam;
      t.rs = fam.h + pad;
      if (fam_row_stride_even(fam) && t.rs % 2) continue;
      if ((fam_lds_bytes(t) + fam_static_lds_bytes(t)) * per_cu <= 160u * 1024u) { fam = t; break; }
    }
  }
  RFX_ON_DEVICE(device);
  rfx_plan* pl = new rfx_plan();
  struct Guard {  // releases the half-built plan if any step below fails
    rfx_plan* p;
    ~Guard() { if (p) rfx_plan_destroy(p); }
  } guard{pl};
  pl->p = *params;
  pl->device = device;
  pl->n_stft = params->n_fft / 2 + 1;
  pl->generic = generic;
  pl->gg = gg;
  pl->frame_stride = generic ? gg.fs : kFrameStride;
  const int F = pl->n_stft;  // linear bins
  hipDeviceProp_t prop;
  RFX_HIP(hipGetDeviceProperties(&prop, device));
  pl->num_cus = prop.multiProcessorCount;
  // per-device kernel attributes (dynamic LDS above 64 KB) and occupancy; environment knobs are read here,
  // once, never on the hot calls
  RFX_HIP(prepare_frame_kernels());
  if (generic) RFX_HIP(prepare_generic_kernels(gg));
  if (fam_ok) {
    RFX_HIP(prepare_fam_kernels(fam));
    pl->fam = fam;
    pl->fam_wgs_per_cu = fam_blocks_per_cu(fam);
    if (const char* e = abl_env("RFX_FAM_WGS_PER_CU")) pl->fam_wgs_per_cu = atoi(e) > 0 ? atoi(e) : 1;
  }
  pl->gl_wgs_per_cu = gl_blocks_per_cu();
  if (const char* e = abl_env("RFX_GL_WGS_PER_CU")) pl->gl_wgs_per_cu = atoi(e) > 0 ? atoi(e) : 1;
  pl->imel_variant = abl_env("RFX_IMEL_GENERAL") ? 2 : abl_env("RFX_IMEL_UNIFORM") ? 1 : abl_env("RFX_IMEL_NO_PAIR") ? 3 : 0;  // 3: best one-frame kernel
  // which Griffin-Lim device form a call takes: the options of rfx_plan_create_ex decide; the environment (read here, once)
  // only changes what RFX_GL_FORM_AUTO / the default threshold mean, for experiments
  if (const char* e = abl_env("RFX_GL_LATENCY_MODE")) pl->gl_latency_mode = atoi(e) != 0;
  if (const char* e = abl_env("RFX_GL_LATENCY_FRAMES")) pl->gl_latency_frames_per_slot = atoi(e) > 0 ? atoi(e) : 6;
  pl->gl_form = opt.gl_form;
  if (opt.gl_frames_per_slot > 0) pl->gl_latency_frames_per_slot = opt.gl_frames_per_slot;
#if defined(RFX_TIMING) || defined(RFX_WGCLOCK)
  if (const char* e = getenv("RFX_TIMING_PTR")) pl->timing = (unsigned long long*)strtoull(e, nullptr, 0);
#endif

  const double PI2 = 6.283185307179586476925286766559;
  std::vector<cf> tw1(21 * kHop), tw2(21 * 21);
  for (int k1 = 0; k1 < 21; ++k1)
    for (int n = 0; n < kHop; ++n) {
      const long long e = ((long long)k1 * (n + 6615)) % kNfft;
      tw1[k1 * kHop + n] = cf{(float)cos(PI2 * (double)e / kNfft), (float)(-sin(PI2 * (double)e / kNfft))};
    }
  for (int i = 0; i < 21; ++i)
    for (int j = 0; j < 21; ++j) {
      const int e = (i * j) % kHop;
      tw2[i * 21 + j] = cf{(float)cos(PI2 * e / (double)kHop), (float)(-sin(PI2 * e / (double)kHop))};
    }
  RFX_HIP(hipMalloc(&pl->d_tw1, tw1.size() * sizeof(cf)));
  RFX_HIP(hipMalloc(&pl->d_tw2, tw2.size() * sizeof(cf)));
  RFX_HIP(hipMalloc(&pl->d_win, (size_t)params->win_length * sizeof(float)));
  RFX_HIP(hipMemcpy(pl->d_tw1, tw1.data(), tw1.size() * sizeof(cf), hipMemcpyHostToDevice));
  RFX_HIP(hipMemcpy(pl->d_tw2, tw2.data(), tw2.size() * sizeof(cf), hipMemcpyHostToDevice));
  RFX_HIP(hipMemcpy(pl->d_win, h_window, (size_t)params->win_length * sizeof(float), hipMemcpyHostToDevice));
  if (generic) {
    // two-level twiddle tables of the Stockham passes (base nc) and of the real <-> packed split (base n_fft)
    std::vector<cf> t(2 * kGenTwLo + gg.nhi + gg.nhi2);
    cf* lo = t.data();
    cf* hi = lo + kGenTwLo;
    cf* lo2 = hi + gg.nhi;
    cf* hi2 = lo2 + kGenTwLo;
    auto root = [&](long long num, long long den) {
      const double a = -PI2 * (double)(num % den) / (double)den;
      return cf{(float)cos(a), (float)sin(a)};
    };
    for (int i = 0; i < kGenTwLo; ++i) { lo[i] = root(i, gg.nc); lo2[i] = root(i, gg.n_fft); }
    for (int i = 0; i < gg.nhi; ++i) hi[i] = root((long long)i * kGenTwLo, gg.nc);
    for (int i = 0; i < gg.nhi2; ++i) hi2[i] = root((long long)i * kGenTwLo, gg.n_fft);
    RFX_HIP(hipMalloc(&pl->d_gen_tables, t.size() * sizeof(cf)));
    RFX_HIP(hipMemcpy(pl->d_gen_tables, t.data(), t.size() * sizeof(cf), hipMemcpyHostToDevice));
    cf* d = (cf*)pl->d_gen_tables;
    pl->gt.lo = d;
    pl->gt.hi = d + kGenTwLo;
    pl->gt.lo2 = d + kGenTwLo + gg.nhi;
    pl->gt.hi2 = d + 2 * kGenTwLo + gg.nhi;
    pl->gt.win = pl->d_win;
    std::vector<int> rev(gg.nc);
    for (int k = 0; k < gg.nc; ++k) rev[k] = gen_ipad(gen_digit_reverse(gg, k), gg.pad_shift);  // LDS position incl. padding
    RFX_HIP(hipMalloc(&pl->d_gen_rev, rev.size() * sizeof(int)));
    RFX_HIP(hipMemcpy(pl->d_gen_rev, rev.data(), rev.size() * sizeof(int), hipMemcpyHostToDevice));
    pl->gt.rev = pl->d_gen_rev;
    // exact twiddles of every pass (double precision, rounded once)
    std::vector<cf> twt((size_t)gen_tw_table_elems(gg) + 1);
    for (int s2 = 0, L = gg.nc; s2 < gg.nstages; ++s2) {
      const int R = gg.radix[s2], m = L / R, off = gen_tw_table_offset(gg, s2);
      for (int i = 0; i < m; ++i)
        for (int q = 1; q < R; ++q) {
          const double ang = -PI2 * (double)(((long long)i * q) % L) / (double)L;
          twt[(size_t)off + (size_t)i * (R - 1) + q - 1] = cf{(float)cos(ang), (float)sin(ang)};
        }
      L = m;
    }
    RFX_HIP(hipMalloc(&pl->d_gen_tw, twt.size() * sizeof(cf)));
    RFX_HIP(hipMemcpy(pl->d_gen_tw, twt.data(), twt.size() * sizeof(cf), hipMemcpyHostToDevice));
    pl->gt.tw = pl->d_gen_tw;
  }
  if (fam_ok) {
    const FamGeom& f = pl->fam;
    std::vector<cf> tw((size_t)f.rows * f.h + (size_t)f.rb * (f.ra - 1));
    for (int k1 = 0; k1 < f.rows; ++k1)
      for (int n = 0; n < f.h; ++n) {  // g(n)^k1 = exp(-2 pi i k1 (n + left) / n_fft); left = 15 h in the 40 h family
        const long long e = ((long long)k1 * (n + f.left)) % f.n_fft;
        tw[(size_t)k1 * f.h + n] = cf{(float)cos(PI2 * (double)e / f.n_fft), (float)(-sin(PI2 * (double)e / f.n_fft))};
      }
    cf* twa = tw.data() + (size_t)f.rows * f.h;
    for (int i = 0; i < f.rb; ++i)
      for (int q = 1; q < f.ra; ++q) {
        const int e = (i * q) % f.h;
        twa[(size_t)(q - 1) * f.rb + i] = cf{(float)cos(PI2 * e / (double)f.h), (float)(-sin(PI2 * e / (double)f.h))};
      }
    RFX_HIP(hipMalloc(&pl->d_fam_tw, tw.size() * sizeof(cf)));
    RFX_HIP(hipMemcpy(pl->d_fam_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    std::vector<int> binof((size_t)f.fsf, -1);
    for (int k1 = 0; k1 < f.rows; ++k1)
      for (int q = 0; q < f.ra; ++q)
        for (int s2 = 0; s2 < f.rb; ++s2) binof[(size_t)s2 * f.nthr + k1 * f.ra + q] = fam_slot_bin(f, k1, q, s2, nullptr);
    RFX_HIP(hipMalloc(&pl->d_fam_binof, binof.size() * sizeof(int)));
    RFX_HIP(hipMemcpy(pl->d_fam_binof, binof.data(), binof.size() * sizeof(int), hipMemcpyHostToDevice));
    pl->fam_ok = true;
  }

  if (h_melfb) {
    const int M = params->n_mels;
    if (M <= 0) return fail(RFX_ERR_INVALID, "rfx_plan_create: n_mels must be positive");
    RFX_HIP(hipMalloc(&pl->d_melfb, (size_t)F * M * sizeof(float)));
    RFX_HIP(hipMemcpy(pl->d_melfb, h_melfb, (size_t)F * M * sizeof(float), hipMemcpyHostToDevice));
    // slot-ordered copy: row of slot position p = filterbank row of its bin for PRIMARY slots, zero for
    // the 440 duplicate slots and the 3 padding positions, so a GEMM over slot order equals the
    // reference's GEMM over bins up to summation order
    const int Mp = (M + 127) / 128 * 128;  // columns padded to the GEMM's 128-row tile: aligned, test-free loads
    pl->melfb_cols = Mp;
    if (!generic) {
    std::vector<float> fbs((size_t)kFrameStride * Mp, 0.f);
    std::vector<char> seen(F, 0);
    for (int k1 = 0; k1 < 21; ++k1)
      for (int ka = 0; ka < 21; ++ka)
        for (int kb = 0; kb < 21; ++kb) {
          bool cj;
          const int bin = slot_bin(k1, ka, kb, &cj);
          if (seen[bin]) continue;
          seen[bin] = 1;
          const int pos = slot_pos_f(k1 * 21 + ka, kb);
          memcpy(&fbs[(size_t)pos * Mp], &h_melfb[(size_t)bin * M], M * sizeof(float));
        }
    RFX_HIP(hipMalloc(&pl->d_melfb_slots, fbs.size() * sizeof(float)));
    RFX_HIP(hipMemcpy(pl->d_melfb_slots, fbs.data(), fbs.size() * sizeof(float), hipMemcpyHostToDevice));
    // K blocks (32 slot positions) with at least one non-zero filterbank row
    std::vector<int> kb;
    for (int blk = 0; blk < kFrameStride / 32; ++blk) {
      bool nz = false;
      for (int r = blk * 32; r < blk * 32 + 32 && !nz; ++r)
        for (int m = 0; m < M; ++m)
          if (fbs[(size_t)r * Mp + m] != 0.f) { nz = true; break; }
      if (nz) kb.push_back(blk);
    }
    pl->n_kblocks = (int)kb.size();
    RFX_HIP(hipMalloc(&pl->d_kblocks, (kb.size() + 1) * sizeof(int)));
    RFX_HIP(hipMemcpy(pl->d_kblocks, kb.data(), kb.size() * sizeof(int), hipMemcpyHostToDevice));
    }  // !generic

    // ---- banded tables for InverseMelScale: every bin feeds at most two ADJACENT mel filters and
    // every filter's support is one contiguous run of bins (true for torchaudio's triangular banks)
    std::vector<int> bin_m0(F, -1), band_lo(M, 0), band_hi(M, 0), csr_ptr(M + 1, 0);
    std::vector<float> bin_w0(F, 0.f), bin_w1(F, 0.f), csr_w;
    bool ok = true;
    std::string why;
    int f_lo = F, f_hi = 0;
    for (int f = 0; f < F && ok; ++f) {
      int first = -1, cnt = 0, last = -1;
      for (int m = 0; m < M; ++m)
        if (h_melfb[(size_t)f * M + m] != 0.f) { if (first < 0) first = m; last = m; ++cnt; }
      if (cnt == 0) continue;
      if (cnt > 2 || last - first != cnt - 1) { ok = false; why = "a linear bin feeds more than two adjacent mel filters"; break; }
      bin_m0[f] = first;
      bin_w0[f] = h_melfb[(size_t)f * M + first];
      bin_w1[f] = cnt == 2 ? h_melfb[(size_t)f * M + first + 1] : 0.f;
      f_lo = f < f_lo ? f : f_lo;
      f_hi = f + 1;
    }
    for (int m = 0; m < M && ok; ++m) {
      int lo = -1, hi = -1;
      for (int f = 0; f < F; ++f)
        if (h_melfb[(size_t)f * M + m] != 0.f) { if (lo < 0) lo = f; hi = f + 1; }
      if (lo < 0) { lo = hi = (f_lo < F ? f_lo : 0); }
      for (int f = lo; f < hi; ++f)
        if (h_melfb[(size_t)f * M + m] == 0.f) { ok = false; why = "a mel filter's support is not contiguous"; break; }
      band_lo[m] = lo;
      band_hi[m] = hi;
      csr_ptr[m] = (int)csr_w.size();
      for (int f = lo; f < hi; ++f) csr_w.push_back(h_melfb[(size_t)f * M + m]);
    }
    csr_ptr[M] = (int)csr_w.size();
    if (ok && (f_hi <= f_lo)) { ok = false; why = "empty filterbank"; }
    if (ok && (f_hi - f_lo > 36 * 256)) { ok = false; why = "more than 9216 active bins"; }
    if (ok && M > 1024) { ok = false; why = "more than 1024 mel filters"; }
    std::vector<int> bin_pos(F, -1), bin_pos2(F, -1);
    if (generic)
      for (int f = 0; f < F; ++f) bin_pos[f] = f;  // plain bin-ordered frames
    else
    for (int k1 = 0; k1 < 21; ++k1)
      for (int ka = 0; ka < 21; ++ka)
        for (int kbq = 0; kbq < 21; ++kbq) {
          bool cj;
          const int bin = slot_bin(k1, ka, kbq, &cj);
          const int pos = slot_pos_f(k1 * 21 + ka, kbq);
          if (bin_pos[bin] < 0) bin_pos[bin] = pos; else bin_pos2[bin] = pos;
        }
    // group formulation (fast kernel): active bins contiguous with no zero row inside, first-filter index
    // non-decreasing, and the per-thread pairing (short group t, long group M-1-t) fits 8 + 24 registers
    std::vector<int> grp_start(M + 1, 0);
    bool fast = ok && M <= 512;
    int fast_code = 0;
    bool unit_form = false, wave_ok = false;
    int line_from_out = 0;  // groups below it are not lines (rfx_kernels.h, ImelTables::line_from)
    std::vector<float> lin;
    if (fast) {
      int prev = 0;
      for (int f = f_lo; f < f_hi && fast; ++f) {
        if (bin_m0[f] < 0 || bin_m0[f] < prev) { fast = false; break; }
        prev = bin_m0[f];
      }
      if (fast) {
        std::vector<int> cnt(M, 0);
        for (int f = f_lo; f < f_hi; ++f) cnt[bin_m0[f]]++;
        int acc = f_lo;
        for (int g2 = 0; g2 < M; ++g2) { grp_start[g2] = acc; acc += cnt[g2]; }
        grp_start[M] = acc;
        // which register budgets the bank's groups fit: thread role t2 owns the long group M-1-t2 and the short group t2
        auto fits = [&](const int* lo_cap, const int* hi_cap) {
          for (int t2 = 0; t2 < 256; ++t2) {
            const int gH = M - 1 - t2, gL = t2 < M - 256 ? t2 : -1;
            if (gH >= 0 && cnt[gH] > hi_cap[t2 >> 6]) return false;
            if (gL >= 0 && cnt[gL] > lo_cap[t2 >> 6]) return false;
            if (gL >= 0 && gH >= 0 && gL >= gH) return false;
          }
          return true;
        };
        // weights as a LINE per group: on a uniform bin grid a triangular filter's weight is linear in the bin index between two
        // centres, w0 = a0 + s0 i, w1 = a1 + s1 i for the group's i-th bin (least-squares line in double, checked per bin).  The
        // tolerance is RELATIVE to the group's largest weight (an area-normalised bank has weights ~1e-2: an absolute 1e-6 would
        // admit 1e-4 relative there).  Measured on the reference's banks (tests/test_round5_cpu.py): 0.72e-7 of the group maximum
        // for htk / no norm, 1.16e-7 for slaney - one ulp of the largest weight; 4e-7 leaves a factor of three.
        // line_from = the lowest group from which every group is a line (group 0 of a bank whose first filter rises over several
        // bins holds that rising edge AND its own falling one: a kink)
        lin.assign(4 * (size_t)M, 0.f);
        int line_from = 0;
        for (int g2 = 0; g2 < M; ++g2) {
          const int n = cnt[g2], f0 = grp_start[g2];
          if (n == 0) continue;
          for (int which = 0; which < 2; ++which) {
            const std::vector<float>& w = which ? bin_w1 : bin_w0;
            double sx = 0, sy = 0, sxx = 0, sxy = 0;
            for (int i = 0; i < n; ++i) { sx += i; sy += w[f0 + i]; sxx += (double)i * i; sxy += (double)i * w[f0 + i]; }
            const double den = n * sxx - sx * sx;
            const double slope = n > 1 ? (n * sxy - sx * sy) / den : 0.0, icpt = (sy - slope * sx) / n;
            const float af = (float)icpt, sf = (float)slope;
            double wmax = 0;
            for (int i = 0; i < n; ++i) wmax = fmax(wmax, fabs((double)w[f0 + i]));
            for (int i = 0; i < n; ++i)
              if (fabs((double)af + (double)sf * i - (double)w[f0 + i]) > 4e-7 * wmax) line_from = g2 + 1;
            lin[(size_t)(2 * which) * M + g2] = af;
            lin[(size_t)(2 * which + 1) * M + g2] = sf;
          }
        }
        const int uni_lo[4] = {8, 8, 8, 8}, uni_hi[4] = {24, 24, 24, 24};
        // per-wave budgets of imel_group_kernel_perwave (rfx_kernels.h): the default bank's exact set, then the wide set; banks
        // whose groups are too long for either (max_frequency above ~11 kHz at 512 filters - the reference's own round-trip test
        // uses 20 Hz .. 20 kHz, test/spectrogram_converter_test.py:46-53 - or fewer filters) take the line-form group kernel
        // (round 5: imel_line_kernel_perwave, code 5) when their LONG groups M-256 .. M-1 are lines; they ran on the general LDS
        // kernel until then: 169 ms per 64 tiles against 4.5 for the default bank
        // (a long group that is NOT a line - group 0 of a bank with at most 256 filters - moves into its thread's free table-form slot)
        auto fits_line = [&]() {
          for (int t2 = 0; t2 < 256; ++t2) {
            const int gH = M - 1 - t2, gL = t2 < M - 256 ? t2 : -1, c = t2 >> 6;
            if (gH >= 0 && gH < line_from) {
              if (gL >= 0 || cnt[gH] > rfx::kImelLoCapLine[c]) return false;
            } else if (gH >= 0 && cnt[gH] > rfx::kImelHiCapLine[c]) return false;
            if (gL >= 0 && cnt[gL] > rfx::kImelLoCapLine[c]) return false;
            if (gL >= 0 && gH >= 0 && gL >= gH) return false;
          }
          return true;
        };
        const bool line_set = fits_line();
        line_from_out = line_from;
        fast_code = fits(rfx::kImelLoCap, rfx::kImelHiCap) ? 2 : fits(rfx::kImelLoCapWide, rfx::kImelHiCapWide) ? 3 : line_set ? 5 : fits(uni_lo, uni_hi) ? 1 : 0;
        fast = fast_code != 0;
        // unit form of the gradient (rfx_imel.hip): the long groups M-256 .. M-1 must have w0 + w1 == 1 per bin (triangular
        // filters, no area normalisation), the last one w1 == 0 throughout (there is no filter M)
        unit_form = fast_code >= 2;
        for (int f = f_lo; f < f_hi && unit_form; ++f) {
          const int g2 = bin_m0[f];
          if (g2 < M - 256) continue;
          if (g2 == M - 1) unit_form = bin_w1[f] == 0.f;
          else unit_form = fabsf(bin_w0[f] + bin_w1[f] - 1.f) <= 1e-6f;
        }
        // wave kernel (rfx_imel.hip::imel_wave_kernel): 512 groups dealt to 64 lanes in eight chunks whose budgets must hold every
        // group, every group a line; with the unit form (no area normalisation) the upper four chunks need one weight only
        wave_ok = RFX_IMEL_WAVE && opt.imel_form == RFX_IMEL_FORM_AUTO && fast_code == 2 && M == 64 * rfx::kImelWaveChunks && line_from == 0;
        for (int c = 0; c < rfx::kImelWaveChunks && wave_ok; ++c)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = cnt[rfx::imel_wave_group(c, lane)];
            if (n > 2 * rfx::kImelWavePairs[c] || n < 2 * rfx::kImelWaveFullPairs[c]) wave_ok = false;
          }
      }
    }
    pl->imel_ok = ok;
    pl->imel_why = why;
    // ---- fused forward path: per-filter band tables, weights transposed so that lane m reads row i coalesced
    if (ok && (generic || M <= 2 * kThreads)) {
      const int Mpad = (M + 63) / 64 * 64;
      int rows = 1;
      for (int m = 0; m < M; ++m) rows = band_hi[m] - band_lo[m] > rows ? band_hi[m] - band_lo[m] : rows;
      rows = (rows + 7) / 8 * 8;  // the kernel reads eight rows per step
      std::vector<float> wt((size_t)rows * Mpad, 0.f);
      std::vector<int> lo_len(2 * (size_t)Mpad, 0);
      for (int m = 0; m < M; ++m) {
        lo_len[m] = band_lo[m];
        lo_len[Mpad + m] = band_hi[m] - band_lo[m];
        for (int f = band_lo[m]; f < band_hi[m]; ++f) wt[(size_t)(f - band_lo[m]) * Mpad + m] = h_melfb[(size_t)f * M + m];
      }
      RFX_HIP(hipMalloc(&pl->d_band_wt, wt.size() * sizeof(float)));
      RFX_HIP(hipMemcpy(pl->d_band_wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
      if (!generic) {  // where the fused kernel finds bin f in LDS: float view of the cube, primary slot of the bin
        std::vector<int> addr((size_t)rows * Mpad, 0);
        for (int m = 0; m < M; ++m)
          for (int f = band_lo[m]; f < band_hi[m]; ++f) {
            const int k = (f % 40 > 20) ? kNfft - f : f;  // bins with residue 21..39 live in conjugate slots
            const int k1 = k % 40, kp = k / 40;
            addr[(size_t)(f - band_lo[m]) * Mpad + m] = 2 * cube_at(k1, kp % 21, 0) + kp / 21;
          }
        RFX_HIP(hipMalloc(&pl->d_band_addr, addr.size() * sizeof(int)));
        RFX_HIP(hipMemcpy(pl->d_band_addr, addr.data(), addr.size() * sizeof(int), hipMemcpyHostToDevice));
      }
      RFX_HIP(hipMalloc(&pl->d_band_lo, lo_len.size() * sizeof(int)));
      RFX_HIP(hipMemcpy(pl->d_band_lo, lo_len.data(), lo_len.size() * sizeof(int), hipMemcpyHostToDevice));
      pl->band_rows = rows;
      pl->Mpad = Mpad;
      pl->fwd_ok = true;
      // product form of the fused kernel (stft_mel2_kernel).  Needs the group structure of the bank: active bins contiguous,
      // first-filter index non-decreasing, so that filter m = (w1 products of group m-1) + (w0 products of group m).
      // Its sum phase gives every thread one filter and the first wave a second one: Mpad <= kThreads + 64 (banks of up to 512
      // filters); wider banks keep the table form (stft_mel_kernel), which handles two filters per thread up to 2 * kThreads.
      if (!generic && Mpad <= kThreads + 64 && abl_env("RFX_FWD_V1") == nullptr) {
        bool v2 = true;
        std::vector<int> cnt(M, 0), gfirst(M, 0);
        int prev = 0;
        for (int f = f_lo; f < f_hi && v2; ++f) {
          if (bin_m0[f] < prev) v2 = false;
          else { prev = bin_m0[f]; cnt[prev]++; }
        }
        // (a zero row inside [f_lo, f_hi) has bin_m0 == -1 < prev and lands here as "not monotone")
        // LDS layout in floats: [0, kQPad) one dump float per lane, [kQPad, G[M]) the w0 products group by group, then at the distance
        // `arr` the same again for w1 - its dump floats [arr, arr + kQPad) sit behind the w0 array, its products at arr + G[g].
        // (Rounds 3-4 put the dump floats behind both arrays: the w1 dump stores of a non-contributing slot then aimed past the cube
        // for banks beyond 6000 padded bins and relied on the LDS range check dropping them.)
        std::vector<int> G(M + 1, kQPad);  // padded position of group g
        for (int g2 = 0, acc = f_lo; g2 < M; ++g2) { gfirst[g2] = acc; acc += cnt[g2]; G[g2 + 1] = G[g2] + (cnt[g2] + 3) / 4 * 4; }
        // the packed tables of the default-bank kernel want the second array at a compile-time distance: the gap behind G[M] is never read
        const bool packed_ok = G[M] <= kMelProdArr;
        const int arr = packed_ok ? kMelProdArr : G[M];
        const int dump0 = 0;
        if (v2 && arr + G[M] + 16 > 2 * kCubeElems) v2 = false;  // (a short segment's four unconditional 16-byte reads may run 12 floats past the last group)
        // every filter must equal its two group sums exactly: check weights against the dense bank
        for (int m = 0; m < M && v2; ++m)
          for (int f = band_lo[m]; f < band_hi[m] && v2; ++f) {
            const float want = h_melfb[(size_t)f * M + m];
            v2 = (bin_m0[f] == m && bin_w0[f] == want) || (bin_m0[f] == m - 1 && bin_w1[f] == want);
          }
        std::vector<int> pads;
        for (int g2 = 0; g2 < M; ++g2)
          for (int p = G[g2] + cnt[g2]; p < G[g2 + 1]; ++p) pads.push_back(p);
        if ((int)pads.size() > kMelPadsPerThread * kHop) v2 = false;
        if (v2) {
          struct SlotEntry { float w0, w1; };
          std::vector<SlotEntry> tab(21 * (size_t)kQPad, SlotEntry{0.f, 0.f});
          std::vector<int> tab_at(21 * (size_t)kQPad);
          for (int kb = 0; kb < 21; ++kb)
            for (int qp = 0; qp < kQPad; ++qp) tab_at[(size_t)kb * kQPad + qp] = dump0 + qp;
          std::vector<char> seen(F, 0);
          unsigned mask = 0;
          for (int k1 = 0; k1 < 21; ++k1)
            for (int ka = 0; ka < 21; ++ka)
              for (int kb = 0; kb < 21; ++kb) {
                bool cj;
                const int bin = slot_bin(k1, ka, kb, &cj);
                if (seen[bin]) continue;  // the duplicate slot of a bin contributes nothing
                seen[bin] = 1;
                if (bin < f_lo || bin >= f_hi) continue;
                const int g2 = bin_m0[bin];
                tab[(size_t)kb * kQPad + slot_qp(k1 * 21 + ka)] = SlotEntry{bin_w0[bin], bin_w1[bin]};
                tab_at[(size_t)kb * kQPad + slot_qp(k1 * 21 + ka)] = G[g2] + (bin - gfirst[g2]);
                mask |= 1u << kb;
              }
          std::vector<int> padtab((size_t)kMelPadsPerThread * kQPad);
          for (int i = 0; i < kMelPadsPerThread; ++i)
            for (int qp = 0; qp < kQPad; ++qp) padtab[(size_t)i * kQPad + qp] = dump0 + qp;
          for (size_t i = 0; i < pads.size(); ++i) padtab[(i / kHop) * kQPad + slot_qp((int)(i % kHop))] = pads[i];
          std::vector<int> seg(2 * (size_t)Mpad, 0);  // (first float << 4) | 16-byte reads; groups hold at most 60 bins here
          for (int m = 0; m < M && v2; ++m) {
            if (cnt[m] > 60) v2 = false;
            if (m > 0) seg[m] = ((arr + G[m - 1]) << 4) | ((cnt[m - 1] + 3) / 4);  // rising: w1 products of group m-1
            seg[(size_t)Mpad + m] = (G[m] << 4) | ((cnt[m] + 3) / 4);              // falling: w0 products of group m
          }
          if (v2) {
          // packed copies for the default-bank kernel (rfx_kernels.h: pk_at / pk_pad / pk_seg)
          std::vector<unsigned> pk;
          const bool packed = packed_ok && (mask & ~rfx::kKbMaskLow) == 0 && G[M] * 4 <= 65536;  // (16-bit byte addresses of the first array)
          if (packed) {
            pk.assign(5 * (size_t)kQPad + 2 * (size_t)kQPad + 2 * (size_t)Mpad, 0u);
            int kbs[10], n = 0;
            for (int kb = 0; kb < 21; ++kb)
              if ((rfx::kKbMaskLow >> kb) & 1u) kbs[n++] = kb;
            for (int i = 0; i < 5; ++i)
              for (int qp = 0; qp < kQPad; ++qp)
                pk[(size_t)i * kQPad + qp] = (unsigned)(4 * tab_at[(size_t)kbs[2 * i] * kQPad + qp]) | ((unsigned)(4 * tab_at[(size_t)kbs[2 * i + 1] * kQPad + qp]) << 16);
            unsigned* pkpad = pk.data() + 5 * (size_t)kQPad;
            for (int qp = 0; qp < kQPad; ++qp)
              for (int w = 0; w < 2; ++w)
                pkpad[2 * qp + w] = (unsigned)(4 * padtab[(size_t)(2 * w) * kQPad + qp]) | ((unsigned)(4 * padtab[(size_t)(2 * w + 1) * kQPad + qp]) << 16);
            unsigned* pkseg = pkpad + 2 * (size_t)kQPad;
            for (int m = 0; m < Mpad; ++m) {
              pkseg[2 * m] = (unsigned)seg[m];
              pkseg[2 * m + 1] = (unsigned)seg[(size_t)Mpad + m];
            }
          }
          RFX_HIP(hipMalloc(&pl->d_slot_tab, tab.size() * sizeof(SlotEntry)));
          RFX_HIP(hipMemcpy(pl->d_slot_tab, tab.data(), tab.size() * sizeof(SlotEntry), hipMemcpyHostToDevice));
          RFX_HIP(hipMalloc(&pl->d_slot_idx, (padtab.size() + seg.size() + tab_at.size() + pk.size()) * sizeof(int)));
          RFX_HIP(hipMemcpy(pl->d_slot_idx, padtab.data(), padtab.size() * sizeof(int), hipMemcpyHostToDevice));
          RFX_HIP(hipMemcpy(pl->d_slot_idx + padtab.size(), seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice));
          RFX_HIP(hipMemcpy(pl->d_slot_idx + padtab.size() + seg.size(), tab_at.data(), tab_at.size() * sizeof(int), hipMemcpyHostToDevice));
          if (packed) {
            RFX_HIP(hipMemcpy(pl->d_slot_idx + padtab.size() + seg.size() + tab_at.size(), pk.data(), pk.size() * sizeof(unsigned), hipMemcpyHostToDevice));
            pl->fwd_packed_off = (int)(padtab.size() + seg.size() + tab_at.size());
          }
          pl->fwd_kb_mask = mask;
          pl->fwd_prod_arr = arr;
          }
        }
      }
      pl->fwd_unfused = abl_env("RFX_FWD_UNFUSED") != nullptr;
      if (const char* e = abl_env("RFX_FWD_RUN")) pl->fwd_run_cap = atoi(e) > 0 ? atoi(e) : 64;
      if (const char* e = abl_env("RFX_FWD_SKEW")) pl->fwd_run_skew = atoi(e);
    }
    if (ok) {
      // one device blob: csr_w | csr_ptr | band_lo | bin_m0 | bin_w0 | bin_w1 | bin_pos | bin_pos2
      const size_t nnz = csr_w.size();
      size_t off = 0;
      auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
      const size_t o_w = take(nnz * 4), o_ptr = take((M + 1) * 4), o_lo = take(M * 4), o_m0 = take(F * 4),
                   o_w0 = take(F * 4), o_w1 = take(F * 4), o_p = take(F * 4), o_p2 = take(F * 4),
                   o_gs = take((M + 1) * 4), o_lin = take(4 * (size_t)M * 4), o_pb = take((size_t)pl->frame_stride * 4);
      std::vector<char> blob(off);
      memcpy(&blob[o_w], csr_w.data(), nnz * 4);
      memcpy(&blob[o_ptr], csr_ptr.data(), (M + 1) * 4);
      memcpy(&blob[o_lo], band_lo.data(), M * 4);
      memcpy(&blob[o_m0], bin_m0.data(), F * 4);
      memcpy(&blob[o_w0], bin_w0.data(), F * 4);
      memcpy(&blob[o_w1], bin_w1.data(), F * 4);
      memcpy(&blob[o_p], bin_pos.data(), F * 4);
      memcpy(&blob[o_p2], bin_pos2.data(), F * 4);
      memcpy(&blob[o_gs], grp_start.data(), (M + 1) * 4);
      {  // output position -> bin (a bin with two slots appears at both; padding positions hold zeros)
        std::vector<int> pos_bin((size_t)pl->frame_stride, -1);
        for (int f = 0; f < F; ++f) {
          if (bin_pos[f] >= 0) pos_bin[bin_pos[f]] = f;
          if (bin_pos2[f] >= 0) pos_bin[bin_pos2[f]] = f;
        }
        memcpy(&blob[o_pb], pos_bin.data(), pos_bin.size() * 4);
      }
      if (fast && lin.size() == 4 * (size_t)M) memcpy(&blob[o_lin], lin.data(), 4 * (size_t)M * 4);
      RFX_HIP(hipMalloc(&pl->d_imel_blob, off));
      RFX_HIP(hipMemcpy(pl->d_imel_blob, blob.data(), off, hipMemcpyHostToDevice));
      char* d = (char*)pl->d_imel_blob;
      pl->imel.csr_w = (const float*)(d + o_w);
      pl->imel.csr_ptr = (const int*)(d + o_ptr);
      pl->imel.band_lo = (const int*)(d + o_lo);
      pl->imel.bin_m0 = (const int*)(d + o_m0);
      pl->imel.bin_w0 = (const float*)(d + o_w0);
      pl->imel.bin_w1 = (const float*)(d + o_w1);
      pl->imel.bin_pos = (const int*)(d + o_p);
      pl->imel.bin_pos2 = (const int*)(d + o_p2);
      pl->imel.pos_bin = (const int*)(d + o_pb);
      pl->imel.grp_start = (const int*)(d + o_gs);
      pl->imel.fast_ok = fast ? fast_code : 0;
      pl->imel.unit_form = fast && unit_form ? 1 : 0;
      pl->imel.lin = (const float*)(d + o_lin);
      pl->imel.wave_ok = fast && wave_ok ? 1 : 0;
      pl->imel.line_from = line_from_out;
      pl->imel.f_lo = f_lo;
      pl->imel.f_hi = f_hi;
      pl->imel.nnz = (int)nnz;
    }
  }
  guard.p = nullptr;
  *out_plan = pl;
  return RFX_OK;
}

int rfx_plan_destroy(rfx_plan* plan) {
  if (!plan) return RFX_OK;
  {
    DeviceGuard guard(plan->device);
    (void)hipFree(plan->d_tw1);
    (void)hipFree(plan->d_tw2);
    (void)hipFree(plan->d_win);
    (void)hipFree(plan->d_melfb);
    (void)hipFree(plan->d_melfb_slots);
    (void)hipFree(plan->d_kblocks);
    (void)hipFree(plan->d_imel_blob);
    (void)hipFree(plan->d_fam_tw);
    (void)hipFree(plan->d_fam_binof);
    (void)hipFree(plan->d_band_wt);
    (void)hipFree(plan->d_band_lo);
    (void)hipFree(plan->d_band_addr);
    (void)hipFree(plan->d_slot_tab);
    (void)hipFree(plan->d_slot_idx);
    (void)hipFree(plan->d_gen_tables);
    (void)hipFree(plan->d_gen_rev);
    (void)hipFree(plan->d_gen_tw);
  }
  delete plan;
  return RFX_OK;
}

int rfx_pack_magnitudes(const rfx_plan* plan, const float* d_lin_bft, int B, int T, float* d_slots, void* stream) {
  if (!plan || !d_lin_bft || !d_slots || B <= 0 || T <= 0) return fail(RFX_ERR_INVALID, "rfx_pack_magnitudes: bad argument");
  RFX_ON_DEVICE(plan->device);
  if (plan->generic) RFX_HIP(launch_gen_pack(d_lin_bft, d_slots, false, B, plan->n_stft, T, plan->gg.fs, (hipStream_t)stream));
  else RFX_HIP(launch_pack_mag(d_lin_bft, d_slots, B, T, (hipStream_t)stream));
  return RFX_OK;
}
int rfx_pack_complex(const rfx_plan* plan, const void* d_bft, int B, int T, void* d_slots, void* stream) {
  if (!plan || !d_bft || !d_slots || B <= 0 || T <= 0) return fail(RFX_ERR_INVALID, "rfx_pack_complex: bad argument");
  RFX_ON_DEVICE(plan->device);
  if (plan->generic) RFX_HIP(launch_gen_pack(d_bft, d_slots, true, B, plan->n_stft, T, plan->gg.fs, (hipStream_t)stream));
  else RFX_HIP(launch_pack_angles((const cf*)d_bft, (cf*)d_slots, B, T, (hipStream_t)stream));
  return RFX_OK;
}
int rfx_unpack_complex(const rfx_plan* plan, const void* d_slots, int B, int T, void* d_bft, void* stream) {
  if (!plan || !d_bft || !d_slots || B <= 0 || T <= 0) return fail(RFX_ERR_INVALID, "rfx_unpack_complex: bad argument");
  RFX_ON_DEVICE(plan->device);
  if (plan->generic) RFX_HIP(launch_gen_unpack(d_slots, d_bft, true, B, plan->n_stft, T, plan->gg.fs, (hipStream_t)stream));
  else RFX_HIP(launch_unpack_complex((const cf*)d_slots, (cf*)d_bft, B, T, (hipStream_t)stream));
  return RFX_OK;
}

int rfx_stft(const rfx_plan* plan, const float* d_wave, int B, int Lw, float* d_mag_slots, void* d_spec_slots,
             void* stream) {
  if (!plan || !d_wave || B <= 0) return fail(RFX_ERR_INVALID, "rfx_stft: bad argument");
  // torch.stft(center=True, pad_mode="reflect") raises when the pad n_fft/2 is not smaller than the input
  if (Lw <= plan->p.n_fft / 2)
    return fail(RFX_ERR_INVALID, "rfx_stft: reflect padding needs more than n_fft/2 = " + std::to_string(plan->p.n_fft / 2) + " samples");
  RFX_ON_DEVICE(plan->device);
  if (plan->fam_ok) {  // row-family kernels (rfx_fam.hip), same plain layout as the generic engine's
    const FamGeom& f = plan->fam;
    FamFwdArgs fa{};
    fa.g = f;
    fa.wave = d_wave;
    fa.wave_stride = (size_t)Lw;
    fa.Lw = Lw;
    fa.mag = d_mag_slots;
    fa.spec = (cf*)d_spec_slots;
    fa.fs_plain = plan->gg.fs;
    fa.tw1 = plan->d_fam_tw;
    fa.twa = plan->d_fam_tw + (size_t)f.rows * f.h;
    fa.win = plan->d_win;
    fa.B = B;
    fa.T = stft_frames(plan, Lw);
    const long long nframes = (long long)B * fa.T, slots = (long long)plan->num_cus * plan->fam_wgs_per_cu;
    const int nblocks = (int)(nframes < slots ? nframes : slots);
    if (d_mag_slots) RFX_HIP(launch_fam_fwd(0, fa, nblocks, (hipStream_t)stream));
    if (d_spec_slots) RFX_HIP(launch_fam_fwd(1, fa, nblocks, (hipStream_t)stream));
    return RFX_OK;
  }
  if (plan->generic) {
    GenStftArgs g{};
    g.g = plan->gg;
    g.tb = plan->gt;
    g.wave = d_wave;
    g.wave_stride = (size_t)Lw;
    g.mag = d_mag_slots;
    g.spec = (cf*)d_spec_slots;
    g.B = B;
    g.T = stft_frames(plan, Lw);
    g.Lw = Lw;
    if (d_mag_slots) RFX_HIP(launch_gen_stft(0, g, plan->num_cus, (hipStream_t)stream));
    if (d_spec_slots) RFX_HIP(launch_gen_stft(1, g, plan->num_cus, (hipStream_t)stream));
    return RFX_OK;
  }
  StftArgs a;
  a.wave = d_wave;
  a.mag = d_mag_slots;
  a.spec = (cf*)d_spec_slots;
  a.tw1 = plan->d_tw1;
  a.tw2 = plan->d_tw2;
  a.win = plan->d_win;
  a.B = B;
  a.Lw = Lw;
  a.T = 1 + Lw / kHop;
  const long long frames = (long long)B * a.T;
  int fpb = (int)((frames + 2LL * plan->num_cus - 1) / (2LL * plan->num_cus));
  if (fpb < 1) fpb = 1;
  if (fpb > 16) fpb = 16;
  a.frames_per_block = fpb;
  RFX_HIP(launch_stft(a, (hipStream_t)stream));
  return RFX_OK;
}

// Griffin-Lim workspace: three generations (x_{k-1}, x_k, x_{k+1}) of the two parity audio buffers, and the
// istft normalisation table.  No spectral state is kept between iterations (see rfx_gl.hip).
// Small batches take the per-frame kernels (rfx_gl.hip: gl_frame_kernel + gl_fold_kernel): at most four frames per resident
// workgroup slot, where the run-based kernel (>= 10 frames per workgroup) would leave most of the chip idle.
static bool gl_use_latency_mode(const rfx_plan* plan, int B, int T) {
  if (plan->gl_form == RFX_GL_FORM_RUNS) return false;
  if (plan->gl_form == RFX_GL_FORM_FRAMES) return true;
  if (!plan->gl_latency_mode) return false;
  const long long nframes = (long long)B * T, slots = (long long)plan->num_cus * plan->gl_wgs_per_cu;
  if (nframes <= (long long)plan->gl_latency_frames_per_slot * slots) return true;
  // one stair further (round 6): as soon as the batch has more groups than the chip has CUs some CU walks two 16-frame runs and the
  // launch lasts as long as if all did (5.6 ms per Griffin-Lim 32 for nine tiles, 3.7 for eight); the per-frame form still beats
  // that up to ten frames per slot (5.0 / 5.2 ms for nine / ten tiles: profiles/r06_griffinlim_forms_by_batch.txt)
  const long long groups = (long long)B * rfx::gl_groups_per_row(T);
  return groups > plan->num_cus && groups <= slots && 3 * nframes <= 5LL * plan->gl_latency_frames_per_slot * slots;
}

// The runs of one launch of the run-based Griffin-Lim kernel: at most one run per resident workgroup slot of the chip (a launch of
// 520 workgroups on 512 slots runs eight of them alone in a second wave: the ceil(slots / B) runs per clip of rounds 1-4 did that
// for every B that does not divide the slot count).
// Round 6: the unit of the partition is the GROUP (kGlGroup = 16 consecutive frames of a row, rfx_kernels.h): runs are whole groups,
// at most one run per slot and never more runs than groups.  A clip's bits no longer depend on the partition at all (every group
// boundary splits the overlap-add chains, inside a run as between runs), so the partition is free to follow the chip and the batch.
struct GlPartition { int runs, h, w1, w2; };
static GlPartition gl_partition_for(long long slots, int B, int T) {
  const long long N = (long long)B * rfx::gl_groups_per_row(T);
  long long nruns = N;
  if (nruns > slots) nruns = slots;
  if (nruns < 1) nruns = 1;
  // N = q runs + r: the FIRST r runs take q + 1 groups, the others q (gl_run_start with weights q + 1 / q is exact: W_total = N).
  // First, because blocks 0 .. num_cus - 1 are the workgroups the dispatcher places first, one per CU, and the earlier workgroup
  // of a CU wins its issue arbitration: it runs ~12 % faster than the partner that joins it (profiles/r05_wgclock_dispatch_order.txt:
  // 659 against 746 us for 64 frames each), so the extra group of a batch that is not a whole number of groups per slot (B = 65:
  // 32 runs of 80 frames among 480 of 64) lands where there is slack.  (The per-mille skew of round 5 is gone with it: it never
  // moved the step, same file.)
  const long long q = N / nruns, r = N - q * nruns;
  return GlPartition{(int)nruns, (int)r, (int)(q + 1), (int)q};
}
static GlPartition gl_partition(const rfx_plan* plan, int B, int T, int which) {
  (void)which;
  return gl_partition_for((long long)plan->num_cus * plan->gl_wgs_per_cu, B, T);
}

int rfx_griffinlim_runs(const rfx_plan* plan, int B, int T, int which, int64_t* run_starts, int capacity) {
  if (!plan || B <= 0 || T < 2 || plan->generic) return 0;
  const GlPartition p = gl_partition(plan, B, T, which);
  if (run_starts)
    for (int b = 0; b <= p.runs && b < capacity; ++b) run_starts[b] = rfx::gl_run_start_frame(b, p.runs, B, T, p.h, p.w1, p.w2);
  return p.runs;
}

int64_t rfx_debug_run_start(int64_t b, int64_t runs, int64_t n_frames, int64_t h, int64_t w1, int64_t w2) {
  return rfx::gl_run_start(b, runs, n_frames, h, w1, w2);
}

int rfx_debug_gl_partition(int slots, int B, int T, int64_t* run_starts, int capacity) {
  if (slots <= 0 || B <= 0 || T < 2) return 0;
  const GlPartition p = gl_partition_for(slots, B, T);
  if (run_starts)
    for (int b = 0; b <= p.runs && b < capacity; ++b) run_starts[b] = rfx::gl_run_start_frame(b, p.runs, B, T, p.h, p.w1, p.w2);
  return p.runs;
}

int rfx_debug_range_exponents(float max_abs, int mel_units, int* sgd_exponent, int* gl_exponent) {
  if (!sgd_exponent || !gl_exponent) return fail(RFX_ERR_INVALID, "rfx_debug_range_exponents: null argument");
  int k = 0;
  if (max_abs > 0.f) {
    if (max_abs < __builtin_inff()) (void)frexpf(max_abs, &k);
    else k = 129;
  }
  rfx::range_exponents(k, mel_units, sgd_exponent, gl_exponent);
  return RFX_OK;
}

int rfx_griffinlim_form(const rfx_plan* plan, int B, int T) {
  if (!plan || B <= 0 || T < 2) return RFX_GL_FORM_AUTO;
  if (plan->generic) return RFX_GL_FORM_FRAMES;  // the generic engine has one form: frame kernels + fold
  return gl_use_latency_mode(plan, B, T) ? RFX_GL_FORM_FRAMES : RFX_GL_FORM_RUNS;
}

// [B][2] floats of scales + [B] key words (launch_range_scale)
static size_t range_table_bytes(int B) { return align_up((size_t)B * 3 * sizeof(float), 256); }

static void gl_layout(const rfx_plan* plan, int B, int T, size_t& off_audio, size_t& off_scale, size_t& off_frames, size_t& total, int& Lpad) {
  const int L = kHop * (T - 1);
  Lpad = (int)align_up((size_t)L, 64);
  size_t o = 0;
  off_audio = o;
  o += align_up(6 * (size_t)B * Lpad * sizeof(float), 256);
  off_scale = o;
  o += align_up((size_t)Lpad * sizeof(float), 256);
  off_frames = o;
  if (gl_use_latency_mode(plan, B, T)) o += align_up(gl_frame_buffer_bytes(B, T), 256);
  o += range_table_bytes(B);  // the call's own row-scale table (GlArgs::row_scale), at total - range_table_bytes(B)
  total = o;
}

// torch.istft(center=True, length=None) returns n_fft + hop*(T-1) - 2*(n_fft/2) samples: hop*(T-1), plus one when n_fft is odd
static int gen_out_len(const GenGeom& g, int T) { return g.hop * (T - 1) + (g.n_fft & 1); }

// generic path: the windowed synthesis frames and three generations of the audio estimate (x_{k-1}, x_k read; x_{k+1} written)
// (row family: plus the magnitudes re-ordered into slot order, at the end)
static void gen_gl_layout(const rfx_plan* plan, int B, int T, size_t& off_frames, size_t& off_audio, size_t& total, int& Lpad) {
  const GenGeom& g = plan->gg;
  const size_t nf = (size_t)B * T;
  Lpad = (int)align_up((size_t)gen_out_len(g, T), 64);
  size_t o = 0;
  off_frames = o;
  o += align_up(nf * g.fpitch * sizeof(float), 256);
  off_audio = o;
  o += align_up((3 * (size_t)B + 1) * Lpad * sizeof(float), 256);  // + the window envelope of the fold, [Lpad]
  if (plan->fam_ok) o += align_up(nf * plan->fam.fsf * sizeof(float), 256);
  o += range_table_bytes(B);  // as in gl_layout
  total = o;
}

// rfx_call_options as the entry points below see them (NULL / short struct = defaults)
struct CallOpt {
  uint64_t row_base = 0;
  float magnitude_hint = 0.f;
};
static int read_call_options(const rfx_call_options* o, CallOpt* out, const char* who) {
  *out = CallOpt{};
  if (!o) return RFX_OK;
  if (o->struct_size < offsetof(rfx_call_options, row_base) + sizeof(uint64_t))
    return fail(RFX_ERR_INVALID, std::string(who) + ": rfx_call_options.struct_size is not set");
  if (o->flags != 0) return fail(RFX_ERR_INVALID, std::string(who) + ": rfx_call_options.flags must be 0");
  out->row_base = o->row_base;
  if (o->struct_size >= offsetof(rfx_call_options, magnitude_hint) + sizeof(float)) out->magnitude_hint = o->magnitude_hint;
  if (!(out->magnitude_hint >= 0.f) || out->magnitude_hint > 3.0e38f)
    return fail(RFX_ERR_INVALID, std::string(who) + ": rfx_call_options.magnitude_hint must be a finite value >= 0");
  return RFX_OK;
}

// mag_in_fam_slots (rfx_waveform_from_mel on a row-family plan): d_mag already holds the family kernels' slot order [B*T][fsf] -
// InverseMelScale wrote it that way - so the once-per-call re-ordering of the plain frames is left out
static int gen_griffinlim(const rfx_plan* plan, const float* d_mag, const void* d_angles0, uint64_t seed, int B, int T, int n_iter,
                          float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, hipStream_t stream,
                          float* h_launch_ms, const CallOpt& opt, bool mag_in_fam_slots = false, const float* d_row_scale = nullptr) {
  const GenGeom& g = plan->gg;
  const int L = gen_out_len(g, T);
  if (n_iter > 0 && L <= g.n_fft / 2)
    return fail(RFX_ERR_INVALID, "rfx_griffinlim: Padding size should be less than the corresponding input dimension (reflect padding " +
                                 std::to_string(g.n_fft / 2) + " needs more than that many samples)");
  size_t ofr, oa, total;
  int Lpad;
  gen_gl_layout(plan, B, T, ofr, oa, total, Lpad);
  if (workspace_bytes < total) return fail(RFX_ERR_WORKSPACE, "rfx_griffinlim: workspace too small");
  char* ws = (char*)d_workspace;
  float* frames = (float*)(ws + ofr);
  float* gen[3];
  for (int i = 0; i < 3; ++i) gen[i] = (float*)(ws + oa) + (size_t)i * B * Lpad;  // x_k lives in gen[k % 3]
  float* env = (float*)(ws + oa) + (size_t)3 * B * Lpad;
  if (!d_row_scale) {  // numeric range: from the caller's hint, else from the magnitudes themselves (one pass over them)
    float* tab = (float*)(ws + total - range_table_bytes(B));
    const size_t per_row = (size_t)T * ((plan->fam_ok && mag_in_fam_slots) ? (size_t)plan->fam.fsf : (size_t)g.fs);
    RFX_HIP(launch_range_scale(d_mag, per_row, B, opt.magnitude_hint, (unsigned*)(tab + 2 * (size_t)B), nullptr, tab, 1, 0, stream));
    d_row_scale = tab;
  }
  RFX_HIP(launch_gen_env(plan->d_win, env, g, T, L, stream));
  // padded frame rows (gen_frame_layout): the kernels write the window samples only, the fold reads the padding as zeros
  if (g.fshift > 0) RFX_HIP(hipMemsetAsync(frames, 0, (size_t)B * T * g.fpitch * sizeof(float), stream));
  EventList events;
  if (h_launch_ms) {
    RFX_HIP(events.create(n_iter + 2));
    RFX_HIP(hipEventRecord(events.ev[0], stream));
  }
  if (plan->fam_ok) {
    const FamGeom& f = plan->fam;
    const float* S_slots = d_mag;
    if (!mag_in_fam_slots) {
      float* repacked = (float*)(ws + oa + align_up((3 * (size_t)B + 1) * Lpad * sizeof(float), 256));
      RFX_HIP(launch_fam_repack(d_mag, repacked, plan->d_fam_binof, (long long)B * T, g.fs, f.fsf, f.n_stft, stream));
      S_slots = repacked;
    }
    FamGlArgs fa{};
    fa.g = f;
    fa.S = S_slots;
    fa.angles0 = (const cf*)d_angles0;
    fa.fs_plain = g.fs;
    fa.audio_stride = (size_t)Lpad;
    fa.frames = frames;
    fa.row_scale = d_row_scale;
    fa.fpitch = g.fpitch;
    fa.fshift = g.fshift;
    fa.tw1 = plan->d_fam_tw;
    fa.twa = plan->d_fam_tw + (size_t)f.rows * f.h;
    fa.win = plan->d_win;
    fa.mom = momentum / (1.f + momentum);
    fa.seed = seed;
    fa.frame_base = opt.row_base * (uint64_t)T;
    fa.B = B;
    fa.T = T;
    fa.L = L;
    const long long nframes = (long long)B * T, slots = (long long)plan->num_cus * plan->fam_wgs_per_cu;
    const int nblocks = (int)(nframes < slots ? nframes : slots);
    // x_it lives in gen[it % 2]; gen[2] holds d_it = x_it - m x_{it-1} (d_0 = x_0), which the fold writes next to x_it and the
    // next launch analyses (the kernel used to form it from two generations itself)
    for (int it = 0; it <= n_iter; ++it) {
      fa.x_cur = gen[2];
      fa.x_prev = nullptr;
      RFX_HIP(launch_fam_gl(it == 0 ? 0 : 1, fa, nblocks, stream));
      const bool last = it == n_iter;
      RFX_HIP(launch_gen_fold(frames, env, last ? d_wave_out : gen[it % 2], g, B, T, L, last ? (size_t)L : (size_t)Lpad, stream,
                              it == 0 ? nullptr : gen[(it + 1) % 2], last ? nullptr : gen[2], fa.mom, d_row_scale));
      if (h_launch_ms) RFX_HIP(hipEventRecord(events.ev[it + 1], stream));
    }
    if (h_launch_ms) {
      RFX_HIP(hipEventSynchronize(events.ev[n_iter + 1]));
      for (int i = 0; i <= n_iter; ++i) RFX_HIP(hipEventElapsedTime(&h_launch_ms[i], events.ev[i], events.ev[i + 1]));
    }
    return RFX_OK;
  }
  GenGlArgs a{};
  a.g = g;
  a.tb = plan->gt;
  a.S = d_mag;
  a.angles0 = (const cf*)d_angles0;
  a.audio_stride = (size_t)Lpad;
  a.frames = frames;
  a.row_scale = d_row_scale;
  a.mom = momentum / (1.f + momentum);
  a.seed = seed;
  a.frame_base = opt.row_base * (uint64_t)T;
  a.B = B;
  a.T = T;
  a.L = L;
  for (int it = 0; it <= n_iter; ++it) {
    // iteration `it` analyses d = x_{it-1} - m * x_{it-2} (the momentum term of the reference's `rebuilt - m * tprev`, applied in
    // the time domain; d = x_0 for it == 1) and writes the frames of x_it; it == 0 synthesises the initial estimate from
    // S * angles0.  As on the row family, the fold of iteration it - 1 forms d next to x_{it-1} (x_it lives in gen[it % 2], d in
    // gen[2]), so the kernel runs its one-signal mode for every iteration: half the audio loads, same bits.
    a.x_cur = gen[2];
    a.x_prev = nullptr;
    RFX_HIP(launch_gen_gl(it == 0 ? 0 : 1, a, plan->num_cus, stream));
    const bool last = it == n_iter;
    RFX_HIP(launch_gen_fold(frames, env, last ? d_wave_out : gen[it % 2], g, B, T, L, last ? (size_t)L : (size_t)Lpad, stream,
                            it == 0 ? nullptr : gen[(it + 1) % 2], last ? nullptr : gen[2], a.mom, d_row_scale));
    if (h_launch_ms) RFX_HIP(hipEventRecord(events.ev[it + 1], stream));
  }
  if (h_launch_ms) {
    RFX_HIP(hipEventSynchronize(events.ev[n_iter + 1]));
    for (int i = 0; i <= n_iter; ++i) RFX_HIP(hipEventElapsedTime(&h_launch_ms[i], events.ev[i], events.ev[i + 1]));
  }
  return RFX_OK;
}

size_t rfx_griffinlim_workspace_bytes(const rfx_plan* plan, int B, int T) {
  if (!plan || B <= 0 || T < 2) return 0;
  if (plan->generic) {
    size_t a, b, total;
    int Lpad;
    gen_gl_layout(plan, B, T, a, b, total, Lpad);
    return total;
  }
  size_t a, c, fr, total;
  int Lpad;
  gl_layout(plan, B, T, a, c, fr, total, Lpad);
  return total;
}

static int griffinlim_impl(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                           int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                           void* stream_, float* h_launch_ms, const CallOpt& opt, const float* d_row_scale = nullptr) {
  if (!plan || !d_mag_slots || !d_wave_out || !d_workspace) return fail(RFX_ERR_INVALID, "rfx_griffinlim: null argument");
  if (B <= 0 || T < 2 || n_iter < 0) return fail(RFX_ERR_INVALID, "rfx_griffinlim: bad shape");
  if ((long long)B * T > 0x7fffffffLL) return fail(RFX_ERR_INVALID, "rfx_griffinlim: more than 2^31 - 1 frames in one call");
  if (!(momentum >= 0.f && momentum < 1.f)) return fail(RFX_ERR_INVALID, "rfx_griffinlim: momentum must be in [0, 1)");
  if (plan->generic) {
    RFX_ON_DEVICE(plan->device);
    return gen_griffinlim(plan, d_mag_slots, d_angles0_slots, seed, B, T, n_iter, momentum, d_wave_out, d_workspace, workspace_bytes,
                          (hipStream_t)stream_, h_launch_ms, opt, false, d_row_scale);
  }
  // every iteration re-analyses the hop*(T-1)-sample estimate with torch.stft(center=True, reflect):
  // the reference raises there unless the signal is longer than the n_fft/2 padding
  if (n_iter > 0 && kHop * (T - 1) <= kNfft / 2)
    return fail(RFX_ERR_INVALID, "rfx_griffinlim: Padding size should be less than the corresponding input dimension "
                                 "(reflect padding 8820 needs more than 8820 samples, i.e. at least 22 frames)");
  RFX_ON_DEVICE(plan->device);
  hipStream_t stream = (hipStream_t)stream_;
  size_t off_audio, off_scale, off_frames, total;
  int Lpad;
  gl_layout(plan, B, T, off_audio, off_scale, off_frames, total, Lpad);
  if (workspace_bytes < total) return fail(RFX_ERR_WORKSPACE, "rfx_griffinlim: workspace too small");
  const int L = kHop * (T - 1);
  char* ws = (char*)d_workspace;
  float* audio = (float*)(ws + off_audio);
  float* gen[3][2];
  for (int i = 0; i < 3; ++i)
    for (int p = 0; p < 2; ++p) gen[i][p] = audio + (size_t)(2 * i + p) * B * Lpad;
  float* scale = (float*)(ws + off_scale);

  hipLaunchKernelGGL(out_scale_kernel, dim3((L + 255) / 256), dim3(256), 0, stream, plan->d_win, scale, T, L);
  RFX_HIP(hipGetLastError());
  if (!d_row_scale) {  // numeric range: from the caller's hint, else from the magnitudes themselves (one pass over them)
    float* tab = (float*)(ws + total - range_table_bytes(B));
    RFX_HIP(launch_range_scale(d_mag_slots, (size_t)T * kFrameStride, B, opt.magnitude_hint, (unsigned*)(tab + 2 * (size_t)B), nullptr, tab, 1, 0, stream));
    d_row_scale = tab;
  }

  if (gl_use_latency_mode(plan, B, T)) {
    // x_k lives in generation k % 3 (one folded buffer each: gen[k][0]); frame kernel + fold per iteration
    GlFrameArgs fa;
    fa.S = d_mag_slots;
    fa.angles0 = (const cf*)d_angles0_slots;
    fa.frames = (float*)(ws + off_frames);
    fa.row_scale = d_row_scale;
    fa.tw1 = plan->d_tw1;
    fa.tw2 = plan->d_tw2;
    fa.win = plan->d_win;
    fa.B = B;
    fa.T = T;
    fa.L = L;
    fa.Lpad = Lpad;
    fa.mom = momentum / (1.f + momentum);
    fa.seed = seed;
    fa.frame_base = opt.row_base * (uint64_t)T;
    const long long nframes = (long long)B * T;
    const long long slots = (long long)plan->num_cus * plan->gl_wgs_per_cu;
    const int nblocks = (int)(nframes < slots ? nframes : slots);
    EventList events;
    if (h_launch_ms) {
      RFX_HIP(events.create(n_iter + 2));
      RFX_HIP(hipEventRecord(events.ev[0], stream));
    }
    for (int it = 0; it <= n_iter; ++it) {
      fa.audio_in = gen[(it + 2) % 3][0];    // x_{it-1}
      fa.audio_prev = gen[(it + 1) % 3][0];  // x_{it-2}
      RFX_HIP(launch_gl_frame(it == 0 ? 0 : it == 1 ? 1 : 2, fa, nblocks, stream));
      const bool last = it == n_iter;
      RFX_HIP(launch_gl_fold(fa.frames, plan->d_win, scale, last ? d_wave_out : gen[it % 3][0], B, T, L, last ? (size_t)L : (size_t)Lpad, stream));
      if (h_launch_ms) RFX_HIP(hipEventRecord(events.ev[it + 1], stream));
    }
    if (h_launch_ms) {
      RFX_HIP(hipEventSynchronize(events.ev[n_iter + 1]));
      for (int i = 0; i <= n_iter; ++i) RFX_HIP(hipEventElapsedTime(&h_launch_ms[i], events.ev[i], events.ev[i + 1]));
    }
    return RFX_OK;
  }

  GlArgs g;
  g.S = d_mag_slots;
  g.angles0 = (const cf*)d_angles0_slots;
  g.out_scale = scale;
  g.row_scale = d_row_scale;
  g.tw1 = plan->d_tw1;
  g.tw2 = plan->d_tw2;
  g.win = plan->d_win;
  g.B = B;
  g.T = T;
  g.L = L;
  g.Lpad = Lpad;
  g.mom = momentum / (1.f + momentum);
  g.seed = seed;
  g.frame_base = opt.row_base * (uint64_t)T;
  g.timing = plan->timing;
  // runs: the batch's B*T frames, counted clip after clip, are cut into one run per resident workgroup slot (gl_partition)
  GlPartition part = gl_partition(plan, B, T, 0);
  const int nblocks = part.runs;
  auto set_partition = [&](int which) {
    part = gl_partition(plan, B, T, which);
    g.run_h = part.h;
    g.run_w1 = part.w1;
    g.run_w2 = part.w2;
  };
  set_partition(0);

  // optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline leg)
  EventList events;
  std::vector<hipEvent_t>& ev = events.ev;
  if (h_launch_ms) {
    RFX_HIP(events.create(n_iter + 2));
    RFX_HIP(hipEventRecord(ev[0], stream));
  }
  // generation indices: x_k lives in gen[k % 3]
  auto set_io = [&](int k_in, int k_prev, int k_out) {
    for (int p = 0; p < 2; ++p) {
      g.audio_in[p] = gen[k_in][p];
      g.audio_prev[p] = gen[k_prev][p];
      g.audio_out[p] = gen[k_out][p];
    }
  };
  set_io(1, 2, 0);  // MODE 0 reads nothing; writes x_0
#ifdef RFX_WGCLOCK
  g.launch = 0;
#endif
  RFX_HIP(launch_gl_iter(0, g, nblocks, stream));
  if (h_launch_ms) RFX_HIP(hipEventRecord(ev[1], stream));
  set_partition(1);
  for (int it = 1; it <= n_iter; ++it) {
    // iteration `it` analyses x_{it-1} - m*x_{it-2} and writes x_it
    set_io((it - 1) % 3, (it + 1) % 3 /* == (it-2) mod 3 */, it % 3);
#ifdef RFX_WGCLOCK
    g.launch = it;
#endif
    RFX_HIP(launch_gl_iter(it == 1 ? 1 : 2, g, nblocks, stream));
    if (h_launch_ms) RFX_HIP(hipEventRecord(ev[it + 1], stream));
  }
  const int last = n_iter % 3;
  RFX_HIP(launch_gl_combine(gen[last][0], gen[last][1], d_wave_out, B, L, Lpad, stream));
  if (h_launch_ms) {
    RFX_HIP(hipEventSynchronize(ev[n_iter + 1]));
    for (int i = 0; i <= n_iter; ++i) RFX_HIP(hipEventElapsedTime(&h_launch_ms[i], ev[i], ev[i + 1]));
  }
  return RFX_OK;
}

int rfx_griffinlim(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                   int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                   void* stream) {
  return griffinlim_impl(plan, d_mag_slots, d_angles0_slots, seed, B, T, n_iter, momentum, d_wave_out, d_workspace,
                         workspace_bytes, stream, nullptr, CallOpt{});
}

int rfx_griffinlim_ex(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                      int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                      void* stream, const rfx_call_options* options, float* h_launch_ms) {
  CallOpt opt;
  if (int rc = read_call_options(options, &opt, "rfx_griffinlim_ex")) return rc;
  return griffinlim_impl(plan, d_mag_slots, d_angles0_slots, seed, B, T, n_iter, momentum, d_wave_out, d_workspace,
                         workspace_bytes, stream, h_launch_ms, opt);
}

int rfx_griffinlim_timed(const rfx_plan* plan, const float* d_mag_slots, const void* d_angles0_slots, uint64_t seed, int B,
                         int T, int n_iter, float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes,
                         void* stream, float* h_launch_ms) {
  if (!h_launch_ms) return fail(RFX_ERR_INVALID, "rfx_griffinlim_timed: null timing array");
  return griffinlim_impl(plan, d_mag_slots, d_angles0_slots, seed, B, T, n_iter, momentum, d_wave_out, d_workspace,
                         workspace_bytes, stream, h_launch_ms, CallOpt{});
}

int rfx_unpack_magnitudes(const rfx_plan* plan, const float* d_slots, int B, int T, float* d_bft, void* stream) {
  if (!plan || !d_bft || !d_slots || B <= 0 || T <= 0) return fail(RFX_ERR_INVALID, "rfx_unpack_magnitudes: bad argument");
  RFX_ON_DEVICE(plan->device);
  if (plan->generic) RFX_HIP(launch_gen_unpack(d_slots, d_bft, false, B, plan->n_stft, T, plan->gg.fs, (hipStream_t)stream));
  else RFX_HIP(launch_unpack_mag(d_slots, d_bft, B, T, (hipStream_t)stream));
  return RFX_OK;
}

size_t rfx_mel_workspace_bytes(const rfx_plan* plan, int B, int Lw) {
  if (!plan || B <= 0 || Lw <= plan->p.n_fft / 2) return 0;
  const size_t T = (size_t)stft_frames(plan, Lw);
  if (plan->generic)  // magnitudes [B*T][fs] + frame-major mel amplitudes [B*T][Mpad]
    return align_up((size_t)B * T * plan->gg.fs * sizeof(float), 256) + align_up((size_t)B * T * plan->Mpad * sizeof(float), 256);
  // the fused kernel keeps the magnitudes on chip: its scratch is the frame-major copy of the mel amplitudes
  if (plan->fwd_ok && !plan->fwd_unfused) return align_up((size_t)B * T * plan->Mpad * sizeof(float), 256);
  return align_up((size_t)B * T * kFrameStride * sizeof(float), 256);
}

// does the plan's forward path leave the mel amplitudes frame-major ([B*T][Mpad]) in the workspace before transposing them?
static bool forward_has_frame_major(const rfx_plan* plan) { return plan->generic ? plan->fwd_ok : (plan->fwd_ok && !plan->fwd_unfused); }

// rfx_mel_from_waveform, and the front half of rfx_image_from_waveform: there d_mel_out is null (no (B, M, T) copy is made),
// *mel_tm_out receives the frame-major amplitudes and - where the kernel can take it on the fly - max_keys the keys of the maxima its
// workgroups formed, *keys_per_row of them for every row, rows in order (0: it did not; up to T per row: image_keys_bytes)
static int mel_forward(const rfx_plan* plan, const float* d_wave, int B, int Lw, float* d_mel_out, void* d_workspace, size_t workspace_bytes,
                       void* stream, float** mel_tm_out, unsigned* max_keys, int max_group, int* keys_per_row) {
  if (!plan || !d_wave || !d_workspace || (!d_mel_out && !mel_tm_out)) return fail(RFX_ERR_INVALID, "rfx_mel_from_waveform: null argument");
  if (!plan->d_melfb) return fail(RFX_ERR_INVALID, "rfx_mel_from_waveform: plan was created without a mel filterbank");
  if (workspace_bytes < rfx_mel_workspace_bytes(plan, B, Lw) || Lw <= plan->p.n_fft / 2)
    return fail(Lw <= plan->p.n_fft / 2 ? RFX_ERR_INVALID : RFX_ERR_WORKSPACE, "rfx_mel_from_waveform: input too short or workspace too small");
  RFX_ON_DEVICE(plan->device);
  if (plan->generic) {
    if (!plan->fwd_ok) return fail(RFX_ERR_UNSUPPORTED, "rfx_mel_from_waveform: filterbank is not banded: " + plan->imel_why);
    const int T = stft_frames(plan, Lw);
    float* mag = (float*)d_workspace;
    float* mel_tm = (float*)((char*)d_workspace + align_up((size_t)B * T * plan->gg.fs * sizeof(float), 256));
    if (plan->fam_ok && !plan->fwd_unfused) {  // row family: transform and banded projection in one kernel, |X| stays on chip
      const FamGeom& f = plan->fam;
      FamFwdArgs fa{};
      fa.g = f;
      fa.wave = d_wave;
      fa.wave_stride = (size_t)Lw;
      fa.Lw = Lw;
      fa.fs_plain = plan->gg.fs;
      fa.tw1 = plan->d_fam_tw;
      fa.twa = plan->d_fam_tw + (size_t)f.rows * f.h;
      fa.win = plan->d_win;
      fa.B = B;
      fa.T = T;
      fa.mel_tm = mel_tm;
      fa.band_wt = plan->d_band_wt;
      fa.band_lo = plan->d_band_lo;
      fa.band_len = plan->d_band_lo + plan->Mpad;
      fa.M = plan->p.n_mels;
      fa.Mpad = plan->Mpad;
      const long long nframes = (long long)B * T, slots = (long long)plan->num_cus * plan->fam_wgs_per_cu;
      RFX_HIP(launch_fam_fwd(2, fa, (int)(nframes < slots ? nframes : slots), (hipStream_t)stream));
      if (mel_tm_out) *mel_tm_out = mel_tm;
      if (d_mel_out) RFX_HIP(launch_mel_transpose(mel_tm, d_mel_out, B, T, plan->p.n_mels, plan->Mpad, (hipStream_t)stream));
      return RFX_OK;
    }
    int rc = rfx_stft(plan, d_wave, B, Lw, mag, nullptr, stream);
    if (rc) return rc;
    RFX_HIP(launch_gen_mel(mag, mel_tm, plan->d_band_wt, plan->d_band_lo, plan->d_band_lo + plan->Mpad, (long long)B * T, plan->gg.fs,
                           plan->p.n_mels, plan->Mpad, plan->imel.f_lo, plan->imel.f_hi, (hipStream_t)stream));
    if (mel_tm_out) *mel_tm_out = mel_tm;
    if (d_mel_out) RFX_HIP(launch_mel_transpose(mel_tm, d_mel_out, B, T, plan->p.n_mels, plan->Mpad, (hipStream_t)stream));
    return RFX_OK;
  }
  if (plan->fwd_ok && !plan->fwd_unfused) {
    StftMelArgs f;
    f.wave = d_wave;
    f.mel = d_mel_out;
    f.mel_tm = (float*)d_workspace;
    f.tw1 = plan->d_tw1;
    f.tw2 = plan->d_tw2;
    f.win = plan->d_win;
    f.band_wt = plan->d_band_wt;
    f.band_addr = plan->d_band_addr;
    f.band_lo = plan->d_band_lo;
    f.band_len = plan->d_band_lo + plan->Mpad;
    f.B = B;
    f.Lw = Lw;
    f.T = 1 + Lw / kHop;
    f.M = plan->p.n_mels;
    f.Mpad = plan->Mpad;
    f.f_lo = plan->imel.f_lo;
    f.f_hi = plan->imel.f_hi;
    f.slot_tab = plan->d_slot_tab;
    f.pad_tab = plan->d_slot_idx;
    f.filt_seg = plan->d_slot_idx ? plan->d_slot_idx + (size_t)kMelPadsPerThread * kQPad : nullptr;
    f.slot_at = plan->d_slot_idx ? f.filt_seg + 2 * (size_t)plan->Mpad : nullptr;
    f.prod_arr = plan->fwd_prod_arr;
    f.kb_mask = plan->fwd_kb_mask;
    f.pk_at = plan->fwd_packed_off ? reinterpret_cast<const unsigned*>(plan->d_slot_idx + plan->fwd_packed_off) : nullptr;
    f.pk_pad = f.pk_at ? f.pk_at + 5 * (size_t)kQPad : nullptr;
    f.pk_seg = f.pk_at ? f.pk_pad + 2 * (size_t)kQPad : nullptr;
    f.max_keys = plan->d_slot_tab ? max_keys : nullptr;  // (the product-form kernel takes the maximum on the fly)
    f.max_group = max_group > 0 ? max_group : 1;
    if (mel_tm_out) *mel_tm_out = f.mel_tm;
    // runs of consecutive frames: every resident workgroup slot of the chip gets one run when the batch allows it (the
    // product-form kernel carries a sliding input window along a run), at most 64 frames, at least 1
    const long long frames = (long long)B * f.T;
    const int cap = plan->d_slot_tab ? plan->fwd_run_cap : 16;
    int fpb = (int)((frames + 2LL * plan->num_cus - 1) / (2LL * plan->num_cus));
    f.frames_per_block = fpb < 1 ? 1 : fpb > cap ? cap : fpb;
    {  // unequal runs by dispatch order (StftMelArgs::run_skew), only in the shape it was measured in: one wave of workgroups, two per CU
      const int chunks = (f.T + f.frames_per_block - 1) / f.frames_per_block;
      const bool shape_ok = plan->d_slot_tab && chunks % 2 == 0 && chunks * f.frames_per_block == f.T && (long long)B * chunks == 2LL * plan->num_cus;
      const int d = (int)((long long)f.frames_per_block * plan->fwd_run_skew / 1000);
      f.run_skew = shape_ok && d > 0 && d < f.frames_per_block ? d : 0;
      if (keys_per_row) *keys_per_row = f.max_keys ? chunks : 0;
    }
    RFX_HIP(launch_stft_mel(f, (hipStream_t)stream));
    return RFX_OK;
  }
  if (!d_mel_out) return fail(RFX_ERR_INVALID, "rfx_mel_from_waveform: this plan's forward path has no frame-major stage");
  float* mag = (float*)d_workspace;
  int rc = rfx_stft(plan, d_wave, B, Lw, mag, nullptr, stream);
  if (rc) return rc;
  MelArgs a;
  a.mag = mag;
  a.fbs = plan->d_melfb_slots;
  a.kblocks = plan->d_kblocks;
  a.n_kblocks = plan->n_kblocks;
  a.out = d_mel_out;
  a.M = plan->p.n_mels;
  a.Mp = plan->melfb_cols;
  a.T = 1 + Lw / kHop;
  a.N = B * a.T;
  RFX_HIP(launch_mel_gemm(a, (hipStream_t)stream));
  return RFX_OK;
}

int rfx_mel_from_waveform(const rfx_plan* plan, const float* d_wave, int B, int Lw, float* d_mel_out, void* d_workspace,
                          size_t workspace_bytes, void* stream) {
  if (!d_mel_out) return fail(RFX_ERR_INVALID, "rfx_mel_from_waveform: null argument");
  return mel_forward(plan, d_wave, B, Lw, d_mel_out, d_workspace, workspace_bytes, stream, nullptr, nullptr, 1, nullptr);
}

// ---- spectrogram_image_from_audio's device half (spectrogram_image_converter.py:30-51: spectrogram_from_audio, then
// image_util.image_from_spectrogram): waveforms -> mel amplitudes -> uint8 image without the (B, M, T) tensor in between
// the forward kernel leaves one key per workgroup: at most one workgroup per frame
static size_t image_keys_bytes(const rfx_plan* plan, int rows, int Lw) { return align_up((size_t)rows * stft_frames(plan, Lw) * sizeof(unsigned), 256); }

size_t rfx_image_from_waveform_workspace_bytes(const rfx_plan* plan, int N, int stereo, int Lw) {
  if (!plan || N <= 0) return 0;
  const int C = stereo ? 2 : 1;
  const size_t mel_ws = rfx_mel_workspace_bytes(plan, N * C, Lw);
  if (!mel_ws) return 0;
  size_t total = mel_ws + image_keys_bytes(plan, N * C, Lw);
  if (!forward_has_frame_major(plan)) total += align_up((size_t)N * C * plan->p.n_mels * stft_frames(plan, Lw) * sizeof(float), 256);
  return total;
}

int rfx_image_from_waveform(const rfx_plan* plan, const float* d_wave, int N, int stereo, int Lw, const float* d_thresholds255,
                            float* d_clip_max, uint8_t* d_img_out, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!plan || !d_wave || !d_thresholds255 || !d_clip_max || !d_img_out || !d_workspace || N <= 0)
    return fail(RFX_ERR_INVALID, "rfx_image_from_waveform: bad argument");
  if (!plan->d_melfb) return fail(RFX_ERR_INVALID, "rfx_image_from_waveform: plan was created without a mel filterbank");
  if (Lw <= plan->p.n_fft / 2) return fail(RFX_ERR_INVALID, "rfx_image_from_waveform: input too short");
  if (workspace_bytes < rfx_image_from_waveform_workspace_bytes(plan, N, stereo, Lw)) return fail(RFX_ERR_WORKSPACE, "rfx_image_from_waveform: workspace too small");
  RFX_ON_DEVICE(plan->device);
  const int C = stereo ? 2 : 1, B = N * C, T = stft_frames(plan, Lw), M = plan->p.n_mels;
  const size_t mel_ws = rfx_mel_workspace_bytes(plan, B, Lw);
  unsigned* keys = reinterpret_cast<unsigned*>((char*)d_workspace + mel_ws);
  if (!forward_has_frame_major(plan)) {  // (dense-GEMM fall-back of a non-banded bank: the two calls, the tensor in the workspace)
    float* mel = reinterpret_cast<float*>((char*)keys + image_keys_bytes(plan, B, Lw));
    if (int rc = rfx_mel_from_waveform(plan, d_wave, B, Lw, mel, d_workspace, mel_ws, stream)) return rc;
    return rfx_image_encode_u8(mel, N, M, T, stereo, d_thresholds255, d_clip_max, d_img_out, stream);
  }
  float* mel_tm = nullptr;
  int keys_per_row = 0;  // (round 6: one key per workgroup of the forward kernel, every one written by the launch: nothing to zero)
  if (int rc = mel_forward(plan, d_wave, B, Lw, nullptr, d_workspace, mel_ws, stream, &mel_tm, keys, C, &keys_per_row)) return rc;
  // (a kernel that does not take the maximum on the fly: one pass over the frame-major amplitudes; their padding columns are zero
  // and mel amplitudes are not negative)
  if (!keys_per_row) RFX_HIP(launch_clip_max(mel_tm, reinterpret_cast<float*>(keys), N, (size_t)C * T * plan->Mpad, false, (hipStream_t)stream));
  RFX_HIP(launch_image_encode_tm(mel_tm, keys_per_row ? keys : nullptr, C * keys_per_row, keys_per_row ? nullptr : reinterpret_cast<const float*>(keys),
                                 d_thresholds255, d_img_out, d_clip_max, N, M, plan->Mpad, T, C, (hipStream_t)stream));
  return RFX_OK;
}

size_t rfx_mel_scale_workspace_bytes(const rfx_plan* plan, int B, int T) {
  if (!plan || B <= 0 || T <= 0) return 0;
  if (plan->generic)
    return align_up((size_t)B * T * plan->gg.fs * sizeof(float), 256) + align_up((size_t)B * T * plan->Mpad * sizeof(float), 256);
  return align_up((size_t)B * T * kFrameStride * sizeof(float), 256);
}

int rfx_mel_scale(const rfx_plan* plan, const float* d_lin_bft, int B, int T, float* d_mel_out, void* d_workspace,
                  size_t workspace_bytes, void* stream) {
  if (!plan || !d_lin_bft || !d_mel_out || !d_workspace || B <= 0 || T <= 0) return fail(RFX_ERR_INVALID, "rfx_mel_scale: bad argument");
  if (!plan->d_melfb) return fail(RFX_ERR_INVALID, "rfx_mel_scale: plan was created without a mel filterbank");
  if (workspace_bytes < rfx_mel_scale_workspace_bytes(plan, B, T)) return fail(RFX_ERR_WORKSPACE, "rfx_mel_scale: workspace too small");
  RFX_ON_DEVICE(plan->device);
  if (plan->generic) {
    if (!plan->fwd_ok) return fail(RFX_ERR_UNSUPPORTED, "rfx_mel_scale: filterbank is not banded: " + plan->imel_why);
    float* mag = (float*)d_workspace;
    float* mel_tm = (float*)((char*)d_workspace + align_up((size_t)B * T * plan->gg.fs * sizeof(float), 256));
    RFX_HIP(launch_gen_pack(d_lin_bft, mag, false, B, plan->n_stft, T, plan->gg.fs, (hipStream_t)stream));
    RFX_HIP(launch_gen_mel(mag, mel_tm, plan->d_band_wt, plan->d_band_lo, plan->d_band_lo + plan->Mpad, (long long)B * T, plan->gg.fs,
                           plan->p.n_mels, plan->Mpad, plan->imel.f_lo, plan->imel.f_hi, (hipStream_t)stream));
    RFX_HIP(launch_mel_transpose(mel_tm, d_mel_out, B, T, plan->p.n_mels, plan->Mpad, (hipStream_t)stream));
    return RFX_OK;
  }
  float* mag = (float*)d_workspace;
  RFX_HIP(launch_pack_mag(d_lin_bft, mag, B, T, (hipStream_t)stream));
  MelArgs a;
  a.mag = mag;
  a.fbs = plan->d_melfb_slots;
  a.kblocks = plan->d_kblocks;
  a.n_kblocks = plan->n_kblocks;
  a.out = d_mel_out;
  a.M = plan->p.n_mels;
  a.Mp = plan->melfb_cols;
  a.T = T;
  a.N = B * T;
  RFX_HIP(launch_mel_gemm(a, (hipStream_t)stream));
  return RFX_OK;
}

size_t rfx_inverse_mel_workspace_bytes(const rfx_plan* plan, int B, int T) {
  if (!plan || B <= 0 || T <= 0) return 0;
  return align_up((size_t)B * T * plan->p.max_mel_iters * sizeof(float), 256) + align_up((size_t)(B + 1) * sizeof(int), 256) + range_table_bytes(B);
}

// can InverseMelScale write a row-family plan's frames straight in the family kernels' slot order?  (Every kernel that leaves
// through imel_emit_frame can: the output order is just its pos_bin table.  The general LDS kernel stores bin by bin.)
static bool imel_can_emit_fam_slots(const rfx_plan* plan) {
  return plan->generic && plan->fam_ok && plan->imel_ok && plan->d_fam_binof &&
         rfx::imel_kernel_choice(plan->imel, plan->p.n_mels, plan->p.max_mel_iters, plan->imel_variant) != 0;
}

static int inverse_mel_impl(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, const float* d_spec0,
                            uint64_t seed, float* d_mag_slots, void* d_workspace, size_t workspace_bytes, void* stream_, bool fam_slots,
                            const CallOpt& opt, float* d_gl_row_scale = nullptr);

int rfx_inverse_mel(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, const float* d_spec0,
                    uint64_t seed, float* d_mag_slots, void* d_workspace, size_t workspace_bytes, void* stream_) {
  return inverse_mel_impl(plan, d_mel, B, T, channels_per_clip, d_spec0, seed, d_mag_slots, d_workspace, workspace_bytes, stream_, false, CallOpt{});
}

int rfx_inverse_mel_ex(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, const float* d_spec0,
                       uint64_t seed, float* d_mag_slots, void* d_workspace, size_t workspace_bytes, void* stream_,
                       const rfx_call_options* options) {
  CallOpt opt;
  if (int rc = read_call_options(options, &opt, "rfx_inverse_mel_ex")) return rc;
  return inverse_mel_impl(plan, d_mel, B, T, channels_per_clip, d_spec0, seed, d_mag_slots, d_workspace, workspace_bytes, stream_, false, opt);
}

static int inverse_mel_impl(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, const float* d_spec0,
                            uint64_t seed, float* d_mag_slots, void* d_workspace, size_t workspace_bytes, void* stream_, bool fam_slots,
                            const CallOpt& opt, float* d_gl_row_scale) {
  if (!plan || !d_mel || !d_mag_slots || !d_workspace) return fail(RFX_ERR_INVALID, "rfx_inverse_mel: null argument");
  if (!plan->d_melfb) return fail(RFX_ERR_INVALID, "rfx_inverse_mel: plan was created without a mel filterbank");
  if (!plan->imel_ok) return fail(RFX_ERR_UNSUPPORTED, "rfx_inverse_mel: filterbank is not banded: " + plan->imel_why);
  if (B <= 0 || T <= 0 || channels_per_clip <= 0 || B % channels_per_clip)
    return fail(RFX_ERR_INVALID, "rfx_inverse_mel: batch must be a multiple of channels_per_clip");
  if (opt.row_base % (uint64_t)channels_per_clip)
    return fail(RFX_ERR_INVALID, "rfx_inverse_mel: rfx_call_options.row_base must be a multiple of channels_per_clip (clips are not split)");
  if (workspace_bytes < rfx_inverse_mel_workspace_bytes(plan, B, T)) return fail(RFX_ERR_WORKSPACE, "rfx_inverse_mel: workspace too small");
  RFX_ON_DEVICE(plan->device);
  hipStream_t stream = (hipStream_t)stream_;
  const int nclips = B / channels_per_clip;
  char* ws = (char*)d_workspace;
  float* hist = (float*)ws;
  int* it_stop = (int*)(ws + align_up((size_t)B * T * plan->p.max_mel_iters * sizeof(float), 256));
  int* any_early = it_stop + nclips;
  RFX_HIP(hipMemsetAsync(any_early, 0, sizeof(int), stream));
  // numeric range: the power of two each clip's SGD state is held in (and, for the fused call, the Griffin-Lim rows' factors),
  // from the caller's hint or the clip's largest mel amplitude
  float* clip_scale = (float*)(ws + align_up((size_t)B * T * plan->p.max_mel_iters * sizeof(float), 256) + align_up((size_t)(B + 1) * sizeof(int), 256));
  RFX_HIP(launch_range_scale(d_mel, (size_t)channels_per_clip * plan->p.n_mels * T, nclips, opt.magnitude_hint, (unsigned*)(clip_scale + 2 * (size_t)nclips),
                             clip_scale, d_gl_row_scale, channels_per_clip, 1, stream));
  ImelArgs a;
  a.clip_scale = clip_scale;
  a.sc = a.un = 0.f;
  a.tb = plan->imel;
  a.mel = d_mel;
  a.spec0 = d_spec0;
  a.out_slots = d_mag_slots;
  a.loss_hist = hist;
  a.it_limit = nullptr;
  a.B = B;
  a.M = plan->p.n_mels;
  a.T = T;
  a.C = channels_per_clip;
  a.n_stft = plan->n_stft;
  a.out_stride = plan->frame_stride;
  a.plain = plan->generic ? 1 : 0;
  if (fam_slots) {  // (imel_can_emit_fam_slots: the frame's positions are the family kernels' slots)
    a.tb.pos_bin = plan->d_fam_binof;
    a.out_stride = plan->fam.fsf;
  }
  a.max_iter = plan->p.max_mel_iters;
  a.lr = 0.1f;        // sgdargs=None -> {"lr": 0.1, "momentum": 0.9} (torchaudio 0.13 InverseMelScale)
  a.momentum = 0.9f;
  a.seed = seed;
  a.frame_base = opt.row_base * (uint64_t)T;
  if (a.max_iter <= 0) return fail(RFX_ERR_INVALID, "rfx_inverse_mel: max_mel_iters must be positive");
  RFX_HIP(launch_imel(a, plan->imel_variant, stream));
  // reproduce the reference's early exit (tolerance_loss 1e-5, tolerance_change 1e-8,
  // spectrogram_converter.py:94-95): scan the clip losses, then re-run stopped clips for it_stop steps
  RFX_HIP(launch_imel_scan(hist, it_stop, any_early, nclips, channels_per_clip, T, a.max_iter, 1e-5f, 1e-8f, stream));
  a.it_limit = it_stop;
  RFX_HIP(launch_imel(a, plan->imel_variant, stream));
  return RFX_OK;
}

// ---- SpectrogramConverter.waveform_from_mel_amplitudes in one call (spectrogram_converter.py:187-204: inverse_mel_scaler, then
// inverse_spectrogram_func).  rfx_inverse_mel into the head of the workspace, rfx_griffinlim from there: the same two launches
// sequences, the same seeds (seed for the SGD start, seed + 1 for the phases, as the Python layer always called them), the linear
// magnitudes never leave the library.
size_t rfx_waveform_from_mel_workspace_bytes(const rfx_plan* plan, int B, int T) {
  if (!plan || B <= 0 || T <= 0) return 0;
  const size_t imel = rfx_inverse_mel_workspace_bytes(plan, B, T), gl = rfx_griffinlim_workspace_bytes(plan, B, T);
  if (!imel || !gl) return 0;
  const size_t stride = imel_can_emit_fam_slots(plan) ? (size_t)plan->fam.fsf : (size_t)plan->frame_stride;
  return align_up((size_t)B * T * stride * sizeof(float), 256) + range_table_bytes(B) + (imel > gl ? imel : gl);
}

static int waveform_from_mel_impl(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, uint64_t seed, int n_iter,
                                 float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, void* stream, const CallOpt& opt);

int rfx_waveform_from_mel(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, uint64_t seed, int n_iter,
                          float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, void* stream) {
  return waveform_from_mel_impl(plan, d_mel, B, T, channels_per_clip, seed, n_iter, momentum, d_wave_out, d_workspace, workspace_bytes, stream, CallOpt{});
}

int rfx_waveform_from_mel_ex(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, uint64_t seed, int n_iter,
                             float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, void* stream,
                             const rfx_call_options* options) {
  CallOpt opt;
  if (int rc = read_call_options(options, &opt, "rfx_waveform_from_mel_ex")) return rc;
  return waveform_from_mel_impl(plan, d_mel, B, T, channels_per_clip, seed, n_iter, momentum, d_wave_out, d_workspace, workspace_bytes, stream, opt);
}

static int waveform_from_mel_impl(const rfx_plan* plan, const float* d_mel, int B, int T, int channels_per_clip, uint64_t seed, int n_iter,
                                 float momentum, float* d_wave_out, void* d_workspace, size_t workspace_bytes, void* stream, const CallOpt& opt) {
  if (!plan || !d_mel || !d_wave_out || !d_workspace || B <= 0 || T <= 0) return fail(RFX_ERR_INVALID, "rfx_waveform_from_mel: bad argument");
  const size_t need = rfx_waveform_from_mel_workspace_bytes(plan, B, T);
  if (!need) return fail(RFX_ERR_UNSUPPORTED, "rfx_waveform_from_mel: this plan cannot invert (see rfx_inverse_mel / rfx_griffinlim)");
  if (workspace_bytes < need) return fail(RFX_ERR_WORKSPACE, "rfx_waveform_from_mel: workspace too small");
  float* lin = reinterpret_cast<float*>(d_workspace);
  // a row-family plan's magnitudes go from the SGD kernel to the Griffin-Lim kernels in THEIR slot order: the once-per-call
  // re-ordering of plain frames (0.75 ms and 2.5 GB of traffic per 64 tiles at 48 kHz) exists only for callers of the two entry points
  const bool fam_slots = imel_can_emit_fam_slots(plan);
  const size_t lin_bytes = align_up((size_t)B * T * (fam_slots ? (size_t)plan->fam.fsf : (size_t)plan->frame_stride) * sizeof(float), 256);
  float* row_scale = reinterpret_cast<float*>((char*)d_workspace + lin_bytes);  // written by the SGD stage's range pass, read by Griffin-Lim
  const size_t head = lin_bytes + range_table_bytes(B);
  void* rest = (char*)d_workspace + head;
  if (int rc = inverse_mel_impl(plan, d_mel, B, T, channels_per_clip, nullptr, seed, lin, rest, workspace_bytes - head, stream, fam_slots, opt, row_scale)) return rc;
  if (!fam_slots) return griffinlim_impl(plan, lin, nullptr, seed + 1, B, T, n_iter, momentum, d_wave_out, rest, workspace_bytes - head, stream, nullptr, opt, row_scale);
  if (T < 2 || n_iter < 0 || (long long)B * T > 0x7fffffffLL || !(momentum >= 0.f && momentum < 1.f)) return fail(RFX_ERR_INVALID, "rfx_waveform_from_mel: bad shape");
  RFX_ON_DEVICE(plan->device);
  return gen_griffinlim(plan, lin, nullptr, seed + 1, B, T, n_iter, momentum, d_wave_out, rest, workspace_bytes - head, (hipStream_t)stream, nullptr, opt, true, row_scale);
}

int rfx_image_decode_u8(const uint8_t* d_img, int N, int H, int W, int stereo, const float* d_lut256, float* d_mel_out,
                        void* stream) {
  if (!d_img || !d_lut256 || !d_mel_out || N <= 0 || H <= 0 || W <= 0) return fail(RFX_ERR_INVALID, "rfx_image_decode_u8: bad argument");
  int dev;
  if (int rc = device_of(d_mel_out, &dev)) return rc;
  RFX_ON_DEVICE(dev);
  RFX_HIP(launch_image_decode(d_img, d_lut256, d_mel_out, N, H, W, stereo ? 2 : 1, (hipStream_t)stream));
  return RFX_OK;
}

int rfx_image_encode_u8(const float* d_mel, int N, int M, int T, int stereo, const float* d_thresholds255, float* d_clip_max,
                        uint8_t* d_img_out, void* stream) {
  if (!d_mel || !d_thresholds255 || !d_clip_max || !d_img_out || N <= 0 || M <= 0 || T <= 0)
    return fail(RFX_ERR_INVALID, "rfx_image_encode_u8: bad argument");
  int dev;
  if (int rc = device_of(d_img_out, &dev)) return rc;
  RFX_ON_DEVICE(dev);
  const int C = stereo ? 2 : 1;
  RFX_HIP(launch_clip_max(d_mel, d_clip_max, N, (size_t)C * M * T, false, (hipStream_t)stream));
  RFX_HIP(launch_image_encode(d_mel, d_clip_max, d_thresholds255, d_img_out, N, M, T, C, (hipStream_t)stream));
  return RFX_OK;
}

int rfx_pcm16(const float* d_wave, int N, int C, int L, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* stream) {
  if (!d_wave || !d_clip_peak || !d_pcm_out || N <= 0 || C <= 0 || L <= 0) return fail(RFX_ERR_INVALID, "rfx_pcm16: bad argument");
  int dev;
  if (int rc = device_of(d_pcm_out, &dev)) return rc;
  RFX_ON_DEVICE(dev);
  if (normalize) RFX_HIP(launch_clip_max(d_wave, d_clip_peak, N, (size_t)C * L, true, (hipStream_t)stream));
  RFX_HIP(launch_pcm16(d_wave, d_clip_peak, d_pcm_out, N, L, C, normalize, (hipStream_t)stream));
  return RFX_OK;
}

// ---- SpectrogramImageConverter.audio_from_spectrogram_image's device half in one call (spectrogram_image_converter.py:54-91:
// image_util.spectrogram_from_image, SpectrogramConverter.audio_from_spectrogram -> waveform_from_mel_amplitudes on the image's
// (C, n_mels, T) tensor, audio_util.audio_from_waveform): uint8 tiles in, int16 PCM out.  The three entry points it is made of,
// in their order, with the same seeds: same bytes.
size_t rfx_audio_from_image_workspace_bytes(const rfx_plan* plan, int N, int stereo, int T) {
  if (!plan || N <= 0 || T <= 0) return 0;
  const int C = stereo ? 2 : 1, B = N * C;
  const size_t inner = rfx_waveform_from_mel_workspace_bytes(plan, B, T);
  if (!inner) return 0;
  return align_up((size_t)B * plan->p.n_mels * T * sizeof(float), 256) + align_up((size_t)B * rfx_griffinlim_output_samples(plan, T) * sizeof(float), 256) + inner;
}

static int audio_from_image_impl(const rfx_plan* plan, const uint8_t* d_img, int N, int T, int stereo, const float* d_lut256, uint64_t seed,
                                int n_iter, float momentum, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* d_workspace,
                                size_t workspace_bytes, void* stream, const CallOpt& opt);

int rfx_audio_from_image_u8(const rfx_plan* plan, const uint8_t* d_img, int N, int T, int stereo, const float* d_lut256, uint64_t seed,
                            int n_iter, float momentum, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
  return audio_from_image_impl(plan, d_img, N, T, stereo, d_lut256, seed, n_iter, momentum, normalize, d_clip_peak, d_pcm_out, d_workspace,
                               workspace_bytes, stream, CallOpt{});
}

int rfx_audio_from_image_u8_ex(const rfx_plan* plan, const uint8_t* d_img, int N, int T, int stereo, const float* d_lut256, uint64_t seed,
                               int n_iter, float momentum, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* d_workspace,
                               size_t workspace_bytes, void* stream, const rfx_call_options* options) {
  CallOpt opt;
  if (int rc = read_call_options(options, &opt, "rfx_audio_from_image_u8_ex")) return rc;
  return audio_from_image_impl(plan, d_img, N, T, stereo, d_lut256, seed, n_iter, momentum, normalize, d_clip_peak, d_pcm_out, d_workspace,
                               workspace_bytes, stream, opt);
}

static int audio_from_image_impl(const rfx_plan* plan, const uint8_t* d_img, int N, int T, int stereo, const float* d_lut256, uint64_t seed,
                                int n_iter, float momentum, int normalize, float* d_clip_peak, int16_t* d_pcm_out, void* d_workspace,
                                size_t workspace_bytes, void* stream, const CallOpt& opt) {
  if (!plan || !d_img || !d_lut256 || !d_clip_peak || !d_pcm_out || !d_workspace || N <= 0 || T <= 0)
    return fail(RFX_ERR_INVALID, "rfx_audio_from_image_u8: bad argument");
  const size_t need = rfx_audio_from_image_workspace_bytes(plan, N, stereo, T);
  if (!need) return fail(RFX_ERR_UNSUPPORTED, "rfx_audio_from_image_u8: this plan cannot invert (see rfx_inverse_mel / rfx_griffinlim)");
  if (workspace_bytes < need) return fail(RFX_ERR_WORKSPACE, "rfx_audio_from_image_u8: workspace too small");
  const int C = stereo ? 2 : 1, B = N * C, M = plan->p.n_mels, L = rfx_griffinlim_output_samples(plan, T);
  float* mel = reinterpret_cast<float*>(d_workspace);
  const size_t mel_bytes = align_up((size_t)B * M * T * sizeof(float), 256), wave_bytes = align_up((size_t)B * L * sizeof(float), 256);
  float* wave = reinterpret_cast<float*>((char*)d_workspace + mel_bytes);
  void* rest = (char*)d_workspace + mel_bytes + wave_bytes;
  if (int rc = rfx_image_decode_u8(d_img, N, M, T, stereo, d_lut256, mel, stream)) return rc;
  // (a clip is one image: its channels share the SGD loss mean and the peak normalisation)
  if (int rc = waveform_from_mel_impl(plan, mel, B, T, C, seed, n_iter, momentum, wave, rest, workspace_bytes - mel_bytes - wave_bytes, stream, opt)) return rc;
  return rfx_pcm16(wave, N, C, L, normalize, d_clip_peak, d_pcm_out, stream);
}

}  // extern "C"
