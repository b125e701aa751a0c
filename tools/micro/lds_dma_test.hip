// Micro-test: does `buffer_load_dwordx4 ... lds` write ZEROS for lanes whose
// offset fails the buffer range check?  (needed for padded conv taps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // odd lanes out of range
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(64 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (unsigned)(h.size() * 4), o);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; ++l) printf("lane %d: %x %x %x %x\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
  printf("beyond (untouched?) %x\n", r[300]);
  return 0;
}
