// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write ZEROS into LDS on gfx950,
// or does it leave the destination untouched?  (conv_v3 relies on zeros for the convolution halo.)
//   hipcc --offload-arch=gfx950 -O2 oob_lds_probe.hip -o oob_lds_probe && ./oob_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned* x, int nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* s = (unsigned*)smem;
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeefu;
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..63 out of range (two flavours: just past the end, and 0x7fffffff)
  int off = threadIdx.x * 16;
  if (threadIdx.x >= 32) off = (threadIdx.x & 1) ? 0x7ffffff0 : nbytes + (threadIdx.x - 32) * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)smem, 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
  unsigned h[128]; for (int i = 0; i < 128; i++) h[i] = 0x1000u + i;
  unsigned *dx, *dout; hipMalloc(&dx, 4096); hipMalloc(&dout, 2048);
  hipMemset(dx, 0x55, 4096); hipMemcpy(dx, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, 512, dout);
  unsigned o[512]; hipMemcpy(o, dout, 2048, hipMemcpyDeviceToHost);
  int ok_in = 1, zeros = 1, untouched = 1;
  for (int i = 0; i < 128; i++) ok_in &= (o[i] == 0x1000u + i);
  for (int i = 128; i < 256; i++) { zeros &= (o[i] == 0u); untouched &= (o[i] == 0xdeadbeefu); }
  printf("in-range lanes correct: %d; out-of-range lanes: zeros=%d untouched=%d (sample %08x %08x)\n", ok_in, zeros, untouched, o[128], o[132]);
  printf("rest untouched: %d\n", o[300] == 0xdeadbeefu);
  return !(ok_in && zeros);
}
