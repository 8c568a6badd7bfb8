// Probe: semantics of ds_read_b64_tr_b16 and global_load_lds (dwordx4) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const unsigned short* in, unsigned short* out2) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // lane L supplies the address of chunk L (4 consecutive elements) of a contiguous block
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
  __syncthreads();
  // LDS-DMA: lane i loads 16 B from global in[(63-i)*8 ..] ; where does it land?
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xFFFF;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(in + (63 - threadIdx.x) * 8),
                                   (void __attribute__((address_space(3)))*)(lds + 512), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out2[i] = lds[i];
}
int main() {
  short* out; unsigned short *in, *out2;
  hipMalloc(&out, 256 * 2); hipMalloc(&in, 4096 * 2); hipMalloc(&out2, 2048 * 2);
  unsigned short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)(10000 + i);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(out, in, out2);
  short r[256]; unsigned short r2[2048];
  hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost); hipMemcpy(r2, out2, sizeof(r2), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
  int first = -1, last = -1; for (int i = 0; i < 2048; ++i) if (r2[i] != 0xFFFF) { if (first < 0) first = i; last = i; }
  printf("lds-dma wrote elements [%d, %d]; lds[512..519]=%d..%d lds[520]=%d lds[1016]=%d\n", first, last, r2[512], r2[519], r2[520], r2[1016]);
  return 0;
}
