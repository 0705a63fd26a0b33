// GPU micro-probe: confirms the gfx950 lane layouts the GEMM kernels rely on.
// (1) ds_read_tr16_b64 per-lane-address semantics, (2) MFMA 32x32x16 / 16x16x32 bf16 operand+result maps,
// (3) global_load_lds destination = wave-uniform base + lane*16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__global__ void k_tr(short* out_lin, short* out_custom) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  // linear addressing: lane l -> 4 elems at l*4
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + l * 4));
  for (int j = 0; j < 4; j++) out_lin[l * 4 + j] = t[j];
  // custom: tile [k][n] row-major with pitch 64; group g=l>>4, p=l&15: addr = &tile[8g + (p>>2)][(p&3)*4]
  int g = l >> 4, p = l & 15;
  s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + (8 * g + (p >> 2)) * 64 + (p & 3) * 4));
  for (int j = 0; j < 4; j++) out_custom[l * 4 + j] = u[j];
}

__global__ void k_mfma(const __bf16* A32, const __bf16* B32, float* C32, const __bf16* A16, const __bf16* B16, float* C16) {
  int l = threadIdx.x;
  // 32x32x16: A [32][16] row-major, B [16][32] row-major (k-major)
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = A32[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B32[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; C32[row * 32 + col] = c[r]; }
  // 16x16x32: A [16][32], B [32][16]
  for (int j = 0; j < 8; j++) { a[j] = A16[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B16[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 d = {0};
  d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; r++) { int row = 4 * (l >> 4) + r, col = l & 15; C16[row * 16 + col] = d[r]; }
}

__global__ void k_f32mfma(const float* A, const float* B, float* C) {
  // 32x32x2 f32: A[32][2], B[2][32]
  int l = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; C[row * 32 + col] = c[r]; }
}

__global__ void k_glds(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) lds[i] = -1;
  __syncthreads();
  // each lane sources 8 shorts from a permuted location: lane l reads in[(63-l)*8 ..]
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + (63 - l) * 8),
                                   (__attribute__((address_space(3))) void*)(lds + 512), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 2048; i += 64) out[i] = lds[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz mem=%.1f GB\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.totalGlobalMem / 1e9);
  // ---- tr16
  short *d_lin, *d_cus; CK(hipMalloc(&d_lin, 512)); CK(hipMalloc(&d_cus, 512));
  k_tr<<<1, 64>>>(d_lin, d_cus); CK(hipDeviceSynchronize());
  short h_lin[256], h_cus[256]; CK(hipMemcpy(h_lin, d_lin, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_cus, d_cus, 512, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) { int exp = (l & 15) + j * 16 + (l >> 4) * 64; if (h_lin[l * 4 + j] != exp) bad++; }
  printf("TR16 linear hypothesis (lane l elem j = lds[(l&15)+16j+64(l>>4)]): %s\n", bad ? "FAIL" : "PASS");
  if (bad) { for (int l = 0; l < 64; l++) printf("  lane %2d: %d %d %d %d\n", l, h_lin[l*4], h_lin[l*4+1], h_lin[l*4+2], h_lin[l*4+3]); }
  bad = 0;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) { int g = l >> 4, i = l & 15; int exp = (8 * g + j) * 64 + i; if (h_cus[l * 4 + j] != exp) bad++; }
  printf("TR16 custom hypothesis (lane gets tile[8g+j][i], pitch 64): %s\n", bad ? "FAIL" : "PASS");
  if (bad) { for (int l = 0; l < 64; l++) printf("  lane %2d: %d %d %d %d\n", l, h_cus[l*4], h_cus[l*4+1], h_cus[l*4+2], h_cus[l*4+3]); }
  // ---- mfma
  float A32[32*16], B32[16*32], A16[16*32], B16[32*16];
  srand(1);
  for (auto& x : A32) x = (rand() % 7) - 3; for (auto& x : B32) x = (rand() % 5) - 2;
  for (auto& x : A16) x = (rand() % 7) - 3; for (auto& x : B16) x = (rand() % 5) - 2;
  unsigned short hA32[512], hB32[512], hA16[512], hB16[512];
  for (int i = 0; i < 512; i++) { hA32[i] = f2bf(A32[i]); hB32[i] = f2bf(B32[i]); hA16[i] = f2bf(A16[i]); hB16[i] = f2bf(B16[i]); }
  __bf16 *dA32, *dB32, *dA16, *dB16; float *dC32, *dC16;
  CK(hipMalloc(&dA32, 1024)); CK(hipMalloc(&dB32, 1024)); CK(hipMalloc(&dA16, 1024)); CK(hipMalloc(&dB16, 1024));
  CK(hipMalloc(&dC32, 4096)); CK(hipMalloc(&dC16, 1024));
  CK(hipMemcpy(dA32, hA32, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB32, hB32, 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(dA16, hA16, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB16, hB16, 1024, hipMemcpyHostToDevice));
  k_mfma<<<1, 64>>>(dA32, dB32, dC32, dA16, dB16, dC16); CK(hipDeviceSynchronize());
  float C32[1024], C16[256]; CK(hipMemcpy(C32, dC32, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(C16, dC16, 1024, hipMemcpyDeviceToHost));
  bad = 0;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float s = 0; for (int k = 0; k < 16; k++) s += A32[i*16+k] * B32[k*32+j]; if (s != C32[i*32+j]) bad++; }
  printf("MFMA 32x32x16 bf16 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
  bad = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int k = 0; k < 32; k++) s += A16[i*32+k] * B16[k*16+j]; if (s != C16[i*16+j]) bad++; }
  printf("MFMA 16x16x32 bf16 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
  // ---- f32 mfma
  float Af[64], Bf[64], Cf[1024]; for (auto& x : Af) x = (rand() % 7) - 3; for (auto& x : Bf) x = (rand() % 5) - 2;
  float *dAf, *dBf, *dCf; CK(hipMalloc(&dAf, 256)); CK(hipMalloc(&dBf, 256)); CK(hipMalloc(&dCf, 4096));
  CK(hipMemcpy(dAf, Af, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dBf, Bf, 256, hipMemcpyHostToDevice));
  k_f32mfma<<<1, 64>>>(dAf, dBf, dCf); CK(hipDeviceSynchronize()); CK(hipMemcpy(Cf, dCf, 4096, hipMemcpyDeviceToHost));
  bad = 0;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float s = Af[i*2]*Bf[j] + Af[i*2+1]*Bf[32+j]; if (s != Cf[i*32+j]) bad++; }
  printf("MFMA 32x32x2 f32 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
  // ---- glds
  short hin[512], hout[2048]; for (int i = 0; i < 512; i++) hin[i] = (short)i;
  short *din, *dout; CK(hipMalloc(&din, 1024)); CK(hipMalloc(&dout, 4096)); CK(hipMemcpy(din, hin, 1024, hipMemcpyHostToDevice));
  k_glds<<<1, 64>>>(din, dout); CK(hipDeviceSynchronize()); CK(hipMemcpy(hout, dout, 4096, hipMemcpyDeviceToHost));
  bad = 0;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) if (hout[512 + l * 8 + j] != (63 - l) * 8 + j) bad++;
  for (int i = 0; i < 512; i++) if (hout[i] != -1) bad++;
  for (int i = 1024; i < 2048; i++) if (hout[i] != -1) bad++;
  printf("global_load_lds (dst = uniform base + lane*16, src per-lane): %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
  return 0;
}
