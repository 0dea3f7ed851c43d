// Reproducer (no GPU needed): hipcc of ROCm 7.2 narrows a 64-bit raw buffer load whose result is bit-cast to a FLOAT pair and whose
// two halves are then bit-cast back to integers - the ISA holds ONE buffer_load_dword and the low half is used twice.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S tools/ubench/hipcc_narrowed_load.hip -o - | grep buffer_load
// k_u (integer pair kept as integers) and k_f (float pair used as floats): buffer_load_dwordx2.  k_fi: buffer_load_dword - wrong.
// Found in round 5 (the packed 16-bit tables of stft_mel2_kernel: the padding positions and the falling segment came out as copies of
// their neighbours); rfx_frame.hip.h::ld1u / ld2u load integer tables as integers for that reason.
#include <hip/hip_runtime.h>
using v2f = float __attribute__((ext_vector_type(2)));
using v2u = unsigned __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__global__ void k_u(const unsigned* tab, unsigned* out) {
  const v2u pp = __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(tab, 4096), threadIdx.x * 8u, 0, 0));
  out[threadIdx.x] = pp.x * 3 + pp.y;
}
__global__ void k_f(const unsigned* tab, float* out) {
  const v2f pp = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(tab, 4096), threadIdx.x * 8u, 0, 0));
  out[threadIdx.x] = pp.x * 3.f + pp.y;
}
__global__ void k_fi(const unsigned* tab, unsigned* out) {
  const v2f pp = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(tab, 4096), threadIdx.x * 8u, 0, 0));
  out[threadIdx.x] = __builtin_bit_cast(unsigned, pp.x) * 3 + __builtin_bit_cast(unsigned, pp.y);
}
