#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
__global__ void probe(uint8_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(i & 255);
  __syncthreads();
  const int lane = threadIdx.x;
  // lane p of each 16-lane group points at 8 bytes: row (p>>1) of stride `stride`, col byte (p&1)*8; group g offset g*1024
  const int g = lane >> 4, p = lane & 15;
  const int off = g * 1024 + (p >> 1) * stride + (p & 1) * 8;
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32(LDS_PTR(v2i, lds + off));
  uint32_t w0 = r[0], w1 = r[1];
  for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = (w0 >> (8 * j)) & 255; out[lane * 8 + 4 + j] = (w1 >> (8 * j)) & 255; }
}
int main() {
  uint8_t* d; hipMalloc(&d, 512);
  for (int stride : {16, 32}) {
    probe<<<1, 64>>>(d, stride);
    uint8_t h[512]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %4d", h[l * 8 + j] + ((l >> 4) * 1024 & 0)); printf("\n"); }
  }
  return 0;
}
