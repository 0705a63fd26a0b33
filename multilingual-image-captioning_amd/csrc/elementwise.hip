// elementwise.hip — HBM-bound ops around the GEMMs: ViT im2col/assemble, decoder token embedding fwd/bwd,
// cross-entropy rows/backward, bias column sums, casts, dropout-mask materialisation, fused AdamW.
#include "common.h"
#include <stdarg.h>
#include <type_traits>

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
void mic_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mic_last_error(void) { return g_err; }
extern "C" int mic_version(void) { return 1; }

template <typename F>
static int dispatch_t(int dtype, F&& f) {
  if (dtype == MIC_BF16) f((uint16_t*)nullptr);
  else if (dtype == MIC_F32) f((float*)nullptr);
  else { mic_set_error("bad dtype %d", dtype); return MIC_EINVAL; }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mic_set_error("launch failed: %s", hipGetErrorString(e)); return MIC_ELAUNCH; }
  return MIC_OK;
}
#define TYPE_OF(tag) typename std::remove_pointer<decltype(tag)>::type

// ------------------------------------------------------------------ im2col (K1 gather)
// patches[(b*g+pi)*g+pj][(u*ps+v)*3+c] = pixels[b][pi*ps+u][pj*ps+v][c]; one row of a patch = ps*3 contiguous floats.
template <typename T>
__global__ void im2col_kernel(int B, int img, int ps, const float* __restrict__ px, T* __restrict__ out, int ldp, int trunc) {
  const int g = img / ps;
  const int rowlen = ps * 3;
  const long total = (long)B * g * g * ps * rowlen;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int w = (int)(e % rowlen);
    long t = e / rowlen;
    const int u = (int)(t % ps); t /= ps;
    const int pj = (int)(t % g); t /= g;
    const int pi = (int)(t % g);
    const int b = (int)(t / g);
    float v = px[(((long)b * img + pi * ps + u) * img + pj * ps) * 3 + w];
    if (trunc) v = truncf(v);
    ElemT<T>::st(out + (size_t)((b * g + pi) * g + pj) * ldp + u * rowlen + w, v);
  }
}
// the same gather 8 elements per thread (two 16-B loads, one or two 16-B stores, 32-bit index arithmetic): the first kernel of every
// train step and of every encode — the element-wise form above spends 41-60 us on four 64-bit divisions per element
template <typename T>
__global__ __launch_bounds__(256) void im2col8_kernel(int B, int img, int ps, const float* __restrict__ px, T* __restrict__ out, int ldp, int trunc) {
  const int g = img / ps, chunks = ps * 3 / 8;            // 8-element chunks per patch row
  const int total = B * g * g * ps * chunks;               // < 2^31 (host check)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c = e % chunks;
    int t = e / chunks;
    const int u = t % ps; t /= ps;
    const int pj = t % g; t /= g;
    const int pi = t % g;
    const int b = t / g;
    const float* src = px + (((size_t)b * img + pi * ps + u) * img + pj * ps) * 3 + c * 8;
    float v[8];
    ld8(src, v);
    if (trunc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = truncf(v[i]);
    }
    st8(out + (size_t)((b * g + pi) * g + pj) * ldp + u * ps * 3 + c * 8, v);
  }
}
extern "C" int mic_im2col(int dtype, int B, int img, int ps, const float* pixels, void* patches, int ldp,
                          int trunc_int32, void* stream) {
  MIC_CHECK(B > 0 && img > 0 && ps > 0 && img % ps == 0 && pixels && patches, "mic_im2col: bad args");
  const long total = (long)B * img * img * 3;
  const bool vec = (ps * 3) % 8 == 0 && ldp % 8 == 0 && ((uintptr_t)pixels & 15) == 0 && ((uintptr_t)patches & 15) == 0 && (img * 3) % 4 == 0 && total < (1L << 31);
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    if (vec) {
      int nb = (int)((total / 8 + 255) / 256);
      if (nb > 8192) nb = 8192;
      hipLaunchKernelGGL(im2col8_kernel<T>, dim3(nb), dim3(256), 0, (hipStream_t)stream, B, img, ps, pixels, (T*)patches, ldp, trunc_int32);
    } else {
      hipLaunchKernelGGL(im2col_kernel<T>, dim3(2048), dim3(256), 0, (hipStream_t)stream, B, img, ps, pixels, (T*)patches, ldp, trunc_int32);
    }
  });
}

// ------------------------------------------------------------------ ViT assemble (class token + position embedding)
template <typename T>
__global__ void vit_assemble_kernel(int B, int S, int width, const T* __restrict__ patch, int ldp,
                                    const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ x) {
  const long total = (long)B * S * width;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % width);
    const long r = e / width;
    const int s = (int)(r % S), b = (int)(r / S);
    const float v = (s == 0 ? cls[c] : ElemT<T>::ld(patch + (size_t)(b * (S - 1) + s - 1) * ldp + c)) + pos[s * width + c];
    ElemT<T>::st(x + e, v);
  }
}
extern "C" int mic_vit_assemble(int dtype, int B, int S, int width, const void* patch_out, int ldp, const float* cls,
                                const float* pos, void* x, void* stream) {
  MIC_CHECK(B > 0 && S > 1 && width > 0 && patch_out && cls && pos && x, "mic_vit_assemble: bad args");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    hipLaunchKernelGGL(vit_assemble_kernel<T>, dim3(1024), dim3(256), 0, (hipStream_t)stream, B, S, width, (const T*)patch_out, ldp, cls, pos, (T*)x);
  });
}
// dpatch[b,p] = dx[b,1+p]; dpos[s] += sum_b dx[b,s]; dcls += sum_b dx[b,0].  One thread per (s, c) column sums over b.
template <typename T>
__global__ void vit_assemble_bwd_kernel(int B, int S, int width, const T* __restrict__ dx, T* __restrict__ dpatch, int ldp,
                                        float* __restrict__ dcls, float* __restrict__ dpos) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * width) return;
  const int s = e / width, c = e % width;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const T raw = dx[((size_t)b * S + s) * width + c];
    acc += ElemT<T>::ld(&raw);
    if (s > 0) dpatch[(size_t)(b * (S - 1) + s - 1) * ldp + c] = raw;
  }
  atomicAdd(dpos + e, acc);
  if (s == 0) atomicAdd(dcls + c, acc);
}
// bf16 storage: 8 columns per thread (16-B loads / stores), the batch cut into gridDim.y slices (one thread per column walking all
// B rows was a 64-deep chain on 150 blocks: 25 us on the tail of backward); the slices meet in dpos / dcls by fp32 atomics
__global__ __launch_bounds__(256) void vit_assemble_bwd8_kernel(int B, int S, int width, const uint16_t* __restrict__ dx, uint16_t* __restrict__ dpatch,
                                                                int ldp, float* __restrict__ dcls, float* __restrict__ dpos) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // (s, 8-column chunk)
  const int cw = width >> 3;
  if (e >= S * cw) return;
  const int s = e / cw, c = (e % cw) * 8;
  const int per = (B + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = b0; b < b1; ++b) {
    const u32x4 raw = *reinterpret_cast<const u32x4*>(dx + ((size_t)b * S + s) * width + c);
    float v[8];
    unpack8(raw, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += v[i];
    if (s > 0) *reinterpret_cast<u32x4*>(dpatch + (size_t)(b * (S - 1) + s - 1) * ldp + c) = raw;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    atomicAdd(dpos + (size_t)s * width + c + i, acc[i]);
    if (s == 0) atomicAdd(dcls + c + i, acc[i]);
  }
}
extern "C" int mic_vit_assemble_bwd(int dtype, int B, int S, int width, const void* dx, void* dpatch, int ldp,
                                    float* dcls, float* dpos, void* stream) {
  MIC_CHECK(B > 0 && S > 1 && width > 0 && dx && dpatch && dcls && dpos, "mic_vit_assemble_bwd: bad args");
  if (dtype == MIC_BF16 && width % 8 == 0 && ldp % 8 == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dpatch & 15) == 0) {
    const int slices = B >= 32 ? 8 : (B >= 8 ? 4 : 1);
    hipLaunchKernelGGL(vit_assemble_bwd8_kernel, dim3((S * (width / 8) + 255) / 256, slices), dim3(256), 0, (hipStream_t)stream, B, S, width,
                       (const uint16_t*)dx, (uint16_t*)dpatch, ldp, dcls, dpos);
    MIC_LAUNCH_CHECK();
    return MIC_OK;
  }
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    hipLaunchKernelGGL(vit_assemble_bwd_kernel<T>, dim3((S * width + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, S, width, (const T*)dx, (T*)dpatch, ldp, dcls, dpos);
  });
}

// ------------------------------------------------------------------ decoder token embedding (K8)
template <typename T>
__global__ void embed_fwd_kernel(int rows, int width, const int32_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
                                 const T* __restrict__ table, const float* __restrict__ pos_table, float scale,
                                 T* __restrict__ h) {
  const int nchunk = width >> 3;
  const long total = (long)rows * nchunk;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(e % nchunk);
    const int r = (int)(e / nchunk);
    float a[8], p[8], o[8];
    ld8(table + (size_t)ids[r] * width + ch * 8, a);
    ld8(pos_table + (size_t)(pos_ids[r] + 2) * width + ch * 8, p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = a[i] * scale + p[i];
    st8(h + (size_t)r * width + ch * 8, o);
  }
}
extern "C" int mic_embed_fwd(int dtype, int rows, int width, const int32_t* ids, const int32_t* pos_ids, const void* table,
                             const float* pos_table, float scale, void* h, void* stream) {
  MIC_CHECK(rows > 0 && width > 0 && width % 8 == 0 && ids && pos_ids && table && pos_table && h, "mic_embed_fwd: bad args");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    const long total = (long)rows * (width / 8);
    int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(embed_fwd_kernel<T>, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows, width, ids, pos_ids, (const T*)table, pos_table, scale, (T*)h);
  });
}
template <typename T>
__global__ void embed_bwd_kernel(int rows, int width, const int32_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
                                 const T* __restrict__ dh, float scale, float* __restrict__ dtable, float* __restrict__ dpos) {
  const long total = (long)rows * width;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % width);
    const int r = (int)(e / width);
    const float g = ElemT<T>::ld(dh + e);
    if (dtable) atomicAdd(dtable + (size_t)ids[r] * width + c, g * scale);
    if (dpos) atomicAdd(dpos + (size_t)(pos_ids[r] + 2) * width + c, g);
  }
}
extern "C" int mic_embed_bwd(int dtype, int rows, int width, const int32_t* ids, const int32_t* pos_ids, const void* dh,
                             float scale, float* dtable, float* dpos_table, void* stream) {
  MIC_CHECK(rows > 0 && width > 0 && dh && (dtable || dpos_table) && (!dtable || ids) && (!dpos_table || pos_ids), "mic_embed_bwd: bad args");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    const long total = (long)rows * width;
    int nb = (int)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(embed_bwd_kernel<T>, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows, width, ids, pos_ids, (const T*)dh, scale, dtable, dpos_table);
  });
}

// Deterministic form of the token-embedding scatter for the DATA-PARALLEL step: every rank adds the all-gathered (id, dh) rows of all
// ranks into its already all-reduced dense gradient — with fp32 atomics the order of the adds to one row differs from rank to rank
// and the replicas' embeddings drift apart (measured: 2.8e-9 after 5 steps).  Three small launches over a workspace of
// 3 V + 1 + DET_MAXM ints (owner[V] = INT_MAX, count[V] = 0, last[V] = -1, multi_count = 0 between calls: the kernels restore that state):
//   1. owner[id] = min index, last[id] = max index, count[id] = occurrences (integer atomics: order-free);
//   2. one block per source row j that IS its id's first occurrence: one or two occurrences (nearly all of the ~20 k distinct ids of
//      8 ranks x 4096 rows; first + last index) are added in place; an id with more goes on the multi list;
//   3. one block per listed id: it scans the id list for the occurrences (a bitmap in index order) and adds their rows by a FIXED tree
//      — the occurrences of bitmap word w go to thread group (w ^ (w >> 4)) % 16, index order inside a group, the sixteen partial rows combined in
//      group order — as the single writer of that table row.
// ids < 0 are skipped (padding rows behind a rank's valid ones).  n <= 65536.
#define DET_MAXM 32768  // (n <= 65536 rows hold at most 32768 ids with more than one occurrence: the list cannot overflow)
#define DET_G 16
__global__ void det_owner_kernel(int n, const int32_t* __restrict__ ids, int* __restrict__ owner, int* __restrict__ count, int* __restrict__ last) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int id = ids[i];
  if (id < 0) return;
  atomicMin(owner + id, i);
  atomicMax(last + id, i);
  atomicAdd(count + id, 1);
}
template <typename T>
__global__ __launch_bounds__(256) void det_single_kernel(int n, int width, const int32_t* __restrict__ ids, const T* __restrict__ dh, float scale,
                                                         float* __restrict__ dtable, int* __restrict__ owner, int* __restrict__ count,
                                                         int* __restrict__ last, int* __restrict__ multi_count, int* __restrict__ multi_list) {
  // one WAVE per source row (no block-wide barrier: a wave's loads of owner / count / last precede its lane 0's reset in program order)
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n) return;
  const int id = ids[j];
  if (id < 0 || owner[id] != j) return;  // (wave-uniform; a later occurrence of the id reads INT_MAX or the first index: never its own)
  const int c = count[id], j2 = last[id];
  if (c <= 2) {  // one occurrence, or two (random collisions: ~900 ids at 8 ranks x 4096 rows): first + last index say it all
    for (int col = lane * 4; col < width; col += 256) {
      float* dst = dtable + (size_t)id * width + col;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = ElemT<T>::ld(dh + (size_t)j * width + col + e);
        if (c == 2) v += ElemT<T>::ld(dh + (size_t)j2 * width + col + e);
        dst[e] += v * scale;
      }
    }
    if (lane == 0) { owner[id] = 0x7fffffff; count[id] = 0; last[id] = -1; }
  } else if (lane == 0) {
    const int k = atomicAdd(multi_count, 1);
    if (k < DET_MAXM) multi_list[k] = j;  // (the order of the list does not matter: ids are independent of each other)
  }
}
template <typename T>
__global__ __launch_bounds__(256) void det_multi_kernel(int n, int width, const int32_t* __restrict__ ids, const T* __restrict__ dh, float scale,
                                                        float* __restrict__ dtable, int* __restrict__ owner, int* __restrict__ count,
                                                        int* __restrict__ last, const int* __restrict__ multi_count, const int* __restrict__ multi_list) {
  extern __shared__ __attribute__((aligned(16))) char det_smem[];
  uint32_t* bits = reinterpret_cast<uint32_t*>(det_smem);              // [words]
  const int words = (n + 31) >> 5;
  float* red = reinterpret_cast<float*>(det_smem + (size_t)words * 4);   // [DET_G][1024]
  const int tid = threadIdx.x;
  const int nm = min(*multi_count, DET_MAXM);
  for (int b = blockIdx.x; b < nm; b += gridDim.x) {
    const int j = multi_list[b], id = ids[j];
    for (int w = tid; w < words; w += 256) bits[w] = 0u;
    __syncthreads();
    {
      // the scan: four independent 16-B loads per thread and trip (one id per load and trip was a 128-deep latency chain: 90 us)
      const int4* ids4 = reinterpret_cast<const int4*>(ids);
      const int nvec = n >> 2;
      for (int v0 = tid; v0 < nvec; v0 += 1024) {
        int4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = v0 + u * 256 < nvec ? ids4[v0 + u * 256] : make_int4(-1, -1, -1, -1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i0 = (v0 + u * 256) << 2;
          const int xe[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (xe[e] == id) atomicOr(&bits[(i0 + e) >> 5], 1u << ((i0 + e) & 31));
        }
      }
      for (int i = (nvec << 2) + tid; i < n; i += 256)
        if (ids[i] == id) atomicOr(&bits[i >> 5], 1u << (i & 31));
    }
    __syncthreads();
    const int g = tid >> 4, lane = tid & 15;  // 16 groups x 16 lanes; a lane owns 64 columns of a 1024-column block
    for (int cb = 0; cb < width; cb += 1024) {
      const int c0 = cb + lane * 64;
      float a[64];
#pragma unroll
      for (int e = 0; e < 64; ++e) a[e] = 0.f;
      // this group's occurrences (those in bitmap words w with w % 16 == g, index order), four rows in flight at a time: the adds keep
      // their order
      int k = 0, np = 0, pend[4];
      auto flush = [&](int cnt) __attribute__((always_inline)) {
        u32x4 raw[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < cnt) {
#pragma unroll
            for (int q = 0; q < 8; ++q) raw[u][q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dh) + ((size_t)pend[u] * width + c0) * sizeof(T) + q * 16);
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < cnt) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                float v[8];
                unpack8(raw[u][q], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[q * 8 + e] += v[e];
              }
            } else {  // fp32 rows: 64 columns = 16 pieces of 16 B; the eight loaded above are the first 32 columns
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const float* f = reinterpret_cast<const float*>(&raw[u][q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[q * 4 + e] += f[e];
              }
#pragma unroll
              for (int e = 32; e < 64; ++e) a[e] += ElemT<T>::ld(dh + (size_t)pend[u] * width + c0 + e);
            }
          }
      };
      // this group's words: one of every 16 consecutive ones, rotated by the block of 16 (sequence starts sit 64 rows apart = every
      // other word: a plain w % 16 would leave half the groups idle).  (Every thread walking all 1024 words was a 100-us LDS chain.)
      for (int wb = 0; wb < words; wb += DET_G) {
        const int w = wb + ((g ^ (wb >> 4)) & (DET_G - 1));
        if (w >= words) continue;
        uint32_t m = bits[w];
        while (m) {
          const int i = (w << 5) + __builtin_ctz(m);
          m &= m - 1;
          if (c0 < width) {
            pend[np++] = i;
            if (np == 4) { flush(4); np = 0; }
          }
        }
      }
      if (np) flush(np);
      (void)k;
#pragma unroll
      for (int e = 0; e < 64; ++e) red[g * 1024 + lane * 64 + e] = a[e];
      __syncthreads();
      for (int c = tid; c < 1024 && cb + c < width; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < DET_G; ++q) t += red[q * 1024 + c];
        dtable[(size_t)id * width + cb + c] += t * scale;
      }
      __syncthreads();
    }
    if (tid == 0) { owner[id] = 0x7fffffff; count[id] = 0; last[id] = -1; }
  }
}
__global__ void det_reset_kernel(int* multi_count) { *multi_count = 0; }
extern "C" int mic_embed_rows_add_det(int dtype, int n, int width, int vocab, const int32_t* ids, const void* dh, float scale, float* dtable,
                                      int32_t* ws, void* stream) {
  MIC_CHECK(n > 0 && n <= 65536 && width > 0 && width % 64 == 0 && vocab > 0 && ids && dh && dtable && ws && ((uintptr_t)dh & 15) == 0 && ((uintptr_t)ids & 15) == 0,
            "mic_embed_rows_add_det: bad args (n <= 65536, width %% 64 == 0, 16-B aligned rows, workspace of mic_embed_rows_add_det_ws(vocab) ints)");
  int* owner = ws; int* count = ws + vocab; int* last = ws + 2 * (size_t)vocab; int* mcount = ws + 3 * (size_t)vocab; int* mlist = mcount + 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(det_owner_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, ids, owner, count, last);
  const size_t lds = (size_t)((n + 31) / 32) * 4 + (size_t)DET_G * 1024 * 4;  // <= 8 + 64 KiB
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&det_multi_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 + DET_G * 4096);
      attr_set[dev_ & 63] = true;
    }
    hipLaunchKernelGGL(det_single_kernel<T>, dim3((n + 3) / 4), dim3(256), 0, st, n, width, ids, (const T*)dh, scale, dtable, owner, count, last, mcount, mlist);
    hipLaunchKernelGGL(det_multi_kernel<T>, dim3(512), dim3(256), lds, st, n, width, ids, (const T*)dh, scale, dtable, owner, count, last, mcount, mlist);
    hipLaunchKernelGGL(det_reset_kernel, dim3(1), dim3(1), 0, st, mcount);
  });
}
// ints of the workspace mic_embed_rows_add_det needs for a table of `vocab` rows; the caller initialises it ONCE: ints [0, vocab) to
// 0x7fffffff, [vocab, 2 vocab) to 0, [2 vocab, 3 vocab) to -1, the rest to 0
extern "C" int64_t mic_embed_rows_add_det_ws(int vocab) { return (int64_t)3 * vocab + 1 + DET_MAXM; }

// ------------------------------------------------------------------ cross-entropy over materialised logits (K13)
// One 256-thread block per row; 16-B vector loads; online (max, sum-exp) per thread, combined through LDS.
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { s = 0.f; m = mn; return; }
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_kernel(int V, const T* __restrict__ logits, int ld,
                                                      const int32_t* __restrict__ labels, const int32_t* __restrict__ mask,
                                                      float ls, float* __restrict__ row_lse, float* __restrict__ row_loss) {
  __shared__ float sm[256], ss[256], st[256];
  const int row = blockIdx.x, tid = threadIdx.x;
  const T* lr = logits + (size_t)row * ld;
  float m = -INFINITY, s = 0.f, tot = 0.f;
  const int nchunk = (V + 7) >> 3;
  for (int ch = tid; ch < nchunk; ch += 256) {
    float v[8];
    ld8(lr + ch * 8, v);  // ld is padded to a multiple of 8 columns
    float cm = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (ch * 8 + i < V) cm = fmaxf(cm, v[i]);
    const float mn = fmaxf(m, cm);
    float add = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (ch * 8 + i < V) { add += __expf(v[i] - mn); tot += v[i]; }
    s = s * __expf(m - mn) + add;
    m = mn;
  }
  sm[tid] = m; ss[tid] = s; st[tid] = tot;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      float a = sm[tid], b = ss[tid];
      online_merge(a, b, sm[tid + o], ss[tid + o]);
      sm[tid] = a; ss[tid] = b; st[tid] += st[tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float lse = sm[0] + logf(ss[0]);
    row_lse[row] = lse;
    const float xl = ElemT<T>::ld(lr + labels[row]);
    const float nll = lse - xl;
    float loss = nll;
    if (ls > 0.f) {  // main.py:666-675
      const float conf = 1.0f - ls, low = ls / (float)(V - 1);
      const float norm = -(conf * logf(conf) + (float)(V - 1) * low * logf(low + 1e-20f));
      const float sum_neg_logp = (float)V * lse - st[0];  // -sum_v logp_v
      loss = conf * nll + low * (sum_neg_logp - nll) - norm;
    }
    row_loss[row] = loss;
    (void)mask;
  }
}
extern "C" int mic_ce_rows(int dtype, int rows, int V, const void* logits, int ld, const int32_t* labels,
                           const int32_t* mask, float label_smoothing, float* row_lse, float* row_loss, void* stream) {
  MIC_CHECK(rows > 0 && V > 1 && ld >= V && ld % 8 == 0 && logits && labels && row_lse && row_loss, "mic_ce_rows: bad args");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    hipLaunchKernelGGL(ce_rows_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, V, (const T*)logits, ld, labels, mask, label_smoothing, row_lse, row_loss);
  });
}
// Cross-entropy forward from the head GEMM's per-tile softmax partials (mic_gemm_args.rowstat): one wave per row merges the
// ceil(V / 64) (max, sum exp) pairs into the row's log-sum-exp and reads ONE logit (the label's) — no pass over the
// [rows][250 054] logits.  Plain NLL only (label_smoothing == 0 needs no sum of the logits).
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_tiles_kernel(int rows, const T* __restrict__ logits, int ld, const float2* __restrict__ stat,
                                                           int stat_ld, int ntiles, const int32_t* __restrict__ labels,
                                                           float* __restrict__ row_lse, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2* sr = stat + (size_t)row * stat_ld;
  float m = -INFINITY, s = 0.f;
  if ((stat_ld & 1) == 0 && ((uintptr_t)stat & 15) == 0) {
    // four pairs per trip (two 16-B loads in flight, one rescale of the running sum): one pair per trip was a 61-deep chain of
    // load -> two exps per lane, 39 us for 76 MB at 2.4 k rows
    for (int t = lane * 4; t < ntiles; t += 256) {
      float4 a = *reinterpret_cast<const float4*>(sr + t);
      float4 b = t + 2 < ntiles ? *reinterpret_cast<const float4*>(sr + t + 2) : make_float4(-INFINITY, 0.f, -INFINITY, 0.f);
      if (t + 1 >= ntiles) { a.z = -INFINITY; a.w = 0.f; }
      if (t + 3 >= ntiles) { b.z = -INFINITY; b.w = 0.f; }
      const float mn = fmaxf(fmaxf(m, fmaxf(a.x, a.z)), fmaxf(b.x, b.z));
      if (mn != -INFINITY)
        s = s * __expf(m - mn) + ((a.y * __expf(a.x - mn) + a.w * __expf(a.z - mn)) + (b.y * __expf(b.x - mn) + b.w * __expf(b.z - mn)));
      m = mn;
    }
  } else {
    for (int t = lane; t < ntiles; t += 64) {
      const float2 p = sr[t];
      const float mn = fmaxf(m, p.x);
      s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + p.y * __expf(p.x - mn);
      m = mn;
    }
  }
  const float M = wave_max(m);
  s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - M));
  if (lane == 0) {
    const float lse = M + logf(s);
    row_lse[row] = lse;
    row_loss[row] = lse - ElemT<T>::ld(logits + (size_t)row * ld + labels[row]);
  }
}
extern "C" int mic_ce_rows_tiles(int dtype, int rows, int V, const void* logits, int ld, const float* rowstat, int stat_ld,
                                 const int32_t* labels, float* row_lse, float* row_loss, void* stream) {
  const int ntiles = (V + 63) / 64;
  MIC_CHECK(rows > 0 && V > 1 && ld >= V && logits && rowstat && labels && row_lse && row_loss && stat_ld >= ntiles, "mic_ce_rows_tiles: bad args");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    hipLaunchKernelGGL(ce_rows_tiles_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, (const T*)logits, ld, (const float2*)rowstat, stat_ld, ntiles, labels, row_lse, row_loss);
  });
}
__global__ __launch_bounds__(256) void ce_reduce_kernel(int rows, const float* __restrict__ row_loss,
                                                        const int32_t* __restrict__ mask, float* loss_out, float* denom_out) {
  __shared__ float sl[256], sd[256];
  float l = 0.f, d = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) { const float mk = (float)mask[r]; l += row_loss[r] * mk; d += mk; }
  sl[threadIdx.x] = l; sd[threadIdx.x] = d;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sd[threadIdx.x] += sd[threadIdx.x + o]; } __syncthreads(); }
  if (threadIdx.x == 0) { denom_out[0] = sd[0]; loss_out[0] = sl[0] / sd[0]; }
}
extern "C" int mic_ce_reduce(int rows, const float* row_loss, const int32_t* mask, float* loss_out, float* denom_out, void* stream) {
  MIC_CHECK(rows > 0 && row_loss && mask && loss_out && denom_out, "mic_ce_reduce: bad args");
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, row_loss, mask, loss_out, denom_out);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
// dlogits = mask/denom * loss_scale * (softmax - soft_label); soft_label = conf at label, low elsewhere.
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(int V, int Vpad, T* __restrict__ logits, int ld,
                                                     const int32_t* __restrict__ labels, const int32_t* __restrict__ mask,
                                                     float ls, const float* __restrict__ row_lse, const float* __restrict__ denom,
                                                     float loss_scale) {
  const int row = blockIdx.x;
  T* lr = logits + (size_t)row * ld;
  const float w = mask[row] ? loss_scale / denom[0] : 0.f;
  const float lse = row_lse[row];
  const int label = labels[row];
  const float conf = 1.0f - ls, low = ls > 0.f ? ls / (float)(V - 1) : 0.f;
  const int nchunk = Vpad >> 3;
  constexpr int U = 4;  // chunks per trip, their loads issued together (one 16-B load in flight per thread left HBM at 3.8 TB/s)
  for (int ch0 = threadIdx.x; ch0 < nchunk; ch0 += 256 * U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ch0 + u * 256 < nchunk) ld8(lr + (ch0 + u * 256) * 8, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ch = ch0 + u * 256;
      if (ch >= nchunk) break;
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = ch * 8 + i;
        o[i] = c < V ? w * (__expf(v[u][i] - lse) - (c == label ? conf : low)) : 0.f;
      }
      st8(lr + ch * 8, o);
    }
  }
}
extern "C" int mic_ce_bwd(int dtype, int rows, int V, int Vpad, void* logits, int ld, const int32_t* labels,
                          const int32_t* mask, float label_smoothing, const float* row_lse, const float* denom,
                          float loss_scale, void* stream) {
  MIC_CHECK(rows > 0 && V > 1 && Vpad >= V && Vpad % 8 == 0 && ld >= Vpad && logits && labels && mask && row_lse && denom, "mic_ce_bwd: bad args");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    hipLaunchKernelGGL(ce_bwd_kernel<T>, dim3(rows), dim3(256), 0, (hipStream_t)stream, V, Vpad, (T*)logits, ld, labels, mask, label_smoothing, row_lse, denom, loss_scale);
  });
}

// ------------------------------------------------------------------ 64 x 512 tile transpose (bf16), optionally behind the CE backward
// The LM head's two backward GEMMs reduce over the vocabulary (dX = dlogits E) and over the rows (dE = dlogits^T h): with the
// tensors as forward leaves them, one operand of each is k-major.  The four-wave LDS-DMA kernel (gemm_w4.hip) is an NT kernel —
// LDS-DMA cannot transpose and k-major fragments cost it two transposing reads each — so the head's backward gets k-contiguous
// copies instead: dlogits^T [Vpad][Kp] written by THIS kernel beside the in-place dlogits (one more 1.2 GB write instead of a
// second pass), h^T and E^T [d][Vpad] by the plain transpose below.
// One block = 64 source rows (k) x 512 source columns (x): phase 1 streams the tile row by row (one wave = one row of 512
// columns per trip, 16-B per lane, all 16 rows of a wave requested before the first is used), applies OP, writes the result back
// in place (CE) and into a swizzled LDS image [64 k][64 chunks of 16 B] (chunk c of row k at (c & ~7) | ((c ^ (k >> 3)) & 7):
// the row-wise ds_write_b128 and the column-wise 2-byte reads below are both conflict-free); phase 2 reads 8 consecutive k of one
// x per lane and stores 16 B of dst[x][k0 ..]: one 128-B line per x and block.  Rows k >= rows (up to the grid's 64-row
// multiple) are written as zeros — the padding of the GEMM's reduction dimension.
struct TransposeArgs {
  const uint16_t* src; uint16_t* src_rw; int ld_src;   // source [rows][ld_src]; src_rw != NULL: OP's result is written back in place
  uint16_t* dst; int ld_dst;                            // dst [cols][ld_dst]
  int rows, cols;
  // CE backward (OP = 1): see ce_bwd_kernel
  int V; const int32_t* labels; const int32_t* mask; float ls; const float* row_lse; const float* denom; float loss_scale;
  float* colsum;  // += column sums of the stored dlogits (fp32 atomics; the final_logits_bias gradient), or NULL
};
template <int OP>
__global__ __launch_bounds__(256) void tile_transpose_kernel(TransposeArgs a) {
  __shared__ __attribute__((aligned(16))) char img[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 512, r0 = blockIdx.y * 64;
  const int col = c0 + lane * 8;
  const bool col_ok = col < a.cols;  // (cols % 8 == 0: a chunk is inside or outside as a whole)
  uint4 q[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = r0 + wave * 16 + rr;
    q[rr] = make_uint4(0u, 0u, 0u, 0u);
    if (row < a.rows && col_ok) q[rr] = *reinterpret_cast<const uint4*>(a.src + (size_t)row * a.ld_src + col);
  }
  float inv_denom = 0.f;
  if constexpr (OP == 1) inv_denom = a.loss_scale / a.denom[0];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int k = wave * 16 + rr, row = r0 + k;
    uint4 u = q[rr];
    if constexpr (OP == 1) {
      if (row < a.rows && col_ok) {
        const float w = a.mask[row] ? inv_denom : 0.f;
        const float lse = a.row_lse[row];
        const int label = a.labels[row];
        const float conf = 1.0f - a.ls, low = a.ls > 0.f ? a.ls / (float)(a.V - 1) : 0.f;
        const uint32_t wd[4] = {u.x, u.y, u.z, u.w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = __uint_as_float((i & 1) ? (wd[i >> 1] & 0xffff0000u) : (wd[i >> 1] << 16));
          const int c = col + i;
          o[i] = c < a.V ? w * (__expf(x - lse) - (c == label ? conf : low)) : 0.f;
        }
        u.x = f2bf_pk(o[0], o[1]); u.y = f2bf_pk(o[2], o[3]); u.z = f2bf_pk(o[4], o[5]); u.w = f2bf_pk(o[6], o[7]);
        *reinterpret_cast<uint4*>(a.src_rw + (size_t)row * a.ld_src + col) = u;
      }
    }
    const int pos = (lane & ~7) | ((lane ^ (k >> 3)) & 7);
    *reinterpret_cast<uint4*>(img + k * 1024 + pos * 16) = u;
  }
  __syncthreads();
  const int kc = lane & 7, xx = lane >> 3;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int xl = wave * 128 + it * 8 + xx;  // column of the tile
    const int c = xl >> 3;
    const int pos = (c & ~7) | ((c ^ kc) & 7);
    const char* base = img + (kc * 8) * 1024 + pos * 16 + (xl & 7) * 2;
    uint16_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = *reinterpret_cast<const uint16_t*>(base + i * 1024);
    const int x = c0 + xl;
    if (x < a.cols) {
      uint4 u;
      u.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16); u.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
      u.z = (uint32_t)e[4] | ((uint32_t)e[5] << 16); u.w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
      *reinterpret_cast<uint4*>(a.dst + (size_t)x * a.ld_dst + r0 + kc * 8) = u;
    }
    if constexpr (OP == 1) {
      if (a.colsum != nullptr) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sacc += bf2f(e[i]);
        sacc = group8_sum(sacc);  // the 8 lanes kc = 0..7 of this x
        if (kc == 0 && x < a.cols && sacc != 0.f) atomicAdd(a.colsum + x, sacc);
      }
    }
  }
}
// dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < rows_pad (a multiple of 64; 0 = rows rounded up to 64).  bf16 only.
extern "C" int mic_transpose_bf16(int rows, int rows_pad, int cols, const void* src, int ld_src, void* dst, int ld_dst, void* stream) {
  if (rows_pad <= 0) rows_pad = (rows + 63) / 64 * 64;
  MIC_CHECK(rows > 0 && cols > 0 && cols % 8 == 0 && src && dst && ld_src >= cols && ld_src % 8 == 0 && rows_pad >= rows && rows_pad % 64 == 0 &&
                ld_dst >= rows_pad && ld_dst % 8 == 0,
            "mic_transpose_bf16: bad args (cols %% 8 == 0, ld_src %% 8 == 0, rows_pad %% 64 == 0, ld_dst >= rows_pad and %% 8 == 0)");
  MIC_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "mic_transpose_bf16: 16-B aligned operands");
  TransposeArgs a{};
  a.src = (const uint16_t*)src; a.ld_src = ld_src; a.dst = (uint16_t*)dst; a.ld_dst = ld_dst; a.rows = rows; a.cols = cols;
  hipLaunchKernelGGL(tile_transpose_kernel<0>, dim3((cols + 511) / 512, rows_pad / 64), dim3(256), 0, (hipStream_t)stream, a);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
// mic_ce_bwd (bf16) that ALSO writes dlogits^T [Vpad][ld_t] (columns rows .. rows_pad zero) and, when `colsum` is
// given, adds the column sums of the stored dlogits to it (the final_logits_bias gradient, main.py:178's parameter)
extern "C" int mic_ce_bwd_t(int rows, int V, int Vpad, void* logits, int ld, const int32_t* labels, const int32_t* mask,
                            float label_smoothing, const float* row_lse, const float* denom, float loss_scale, void* dlogits_t,
                            int ld_t, int rows_pad, float* colsum, void* stream) {
  if (rows_pad <= 0) rows_pad = (rows + 63) / 64 * 64;
  MIC_CHECK(rows_pad >= rows && rows_pad % 64 == 0, "mic_ce_bwd_t: rows_pad >= rows, a multiple of 64");
  MIC_CHECK(rows > 0 && V > 1 && Vpad >= V && Vpad % 8 == 0 && ld >= Vpad && ld % 8 == 0 && logits && labels && mask && row_lse && denom && dlogits_t,
            "mic_ce_bwd_t: bad args");
  MIC_CHECK(ld_t >= rows_pad && ld_t % 8 == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits_t & 15) == 0,
            "mic_ce_bwd_t: ld_t >= rows rounded up to 64, ld_t %% 8 == 0, 16-B aligned operands");
  TransposeArgs a{};
  a.src = (const uint16_t*)logits; a.src_rw = (uint16_t*)logits; a.ld_src = ld; a.dst = (uint16_t*)dlogits_t; a.ld_dst = ld_t;
  a.rows = rows; a.cols = Vpad; a.V = V; a.labels = labels; a.mask = mask; a.ls = label_smoothing; a.row_lse = row_lse; a.denom = denom;
  a.loss_scale = loss_scale; a.colsum = colsum;
  hipLaunchKernelGGL(tile_transpose_kernel<1>, dim3((Vpad + 511) / 512, rows_pad / 64), dim3(256), 0, (hipStream_t)stream, a);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ CE backward with the gradient leaving as e5m2 bytes (fp8 LM head)
// dlogits = w (softmax - soft_label), w = mask * loss_scale / denom.  Every entry of g = mask (softmax - soft_label) lies in [-1, 1]:
// this tensor's scale is known in closed form — q = fp8(g * FMAX), dequantisation factor state[1] = loss_scale / (denom * FMAX) — so
// there is no amax history, no saturation and no first-pass special case.  (A delayed amax was tried first: the label entries
// -w (1 - p) are the large ones and early in training they are all ~ -w, so wherever w * scale fell between two e5m2 grid points
// EVERY token's gradient was rounded the same way, up to 12 %, and a previous batch of confident tokens clipped the next batch's
// unconfident ones; the 200-step loss curve left the bf16 one by 10 %.  With FMAX <-> 1 the early label entries are exact.)
// The bytes are written ONCE, q8->q [rows][ldq], instead of in place: the head's two backward GEMMs read that one copy — dX = dlogits
// E^T-copy (k-contiguous) and dE = dlogits^T h (k-major on both sides: ds_read_b64_tr_b8 fragments) — so the 1.2 GB in-place write
// AND the 1.2 GB transposed copy of mic_ce_bwd_t become one 0.6 GB write.  Same 64-row x 512-column blocks as the transposing kernel
// (16 row loads per lane in flight); the column sums of the fp32 gradient (final_logits_bias, modeling:178) are added per block:
// registers over a wave's 16 rows, LDS over the four waves, one fp32 atomic per column.  Padding columns V .. Vpad are zero bytes.
// label_coef: the ONE entry per row that carries at least half of the row's gradient energy, w (p_label - conf), does not go through
// two mantissa bits: it leaves as an fp32 coefficient and its two products are added exactly by head_label_terms_kernel below
// (measured on the 200-step curve: with the label entries in e5m2 the run ended 2-5 % above bf16, without them at the level of the
// fp8 run whose head is bf16).
struct CeQ8Args {
  const uint16_t* logits; int ld; int rows, V, Vpad;
  const int32_t* labels; const int32_t* mask; float ls; const float* row_lse; const float* denom; float loss_scale;
  float* colsum; uint8_t* q; int ldq; float* state; int fmt;
  float* label_coef;  // != NULL: the label entry of every row leaves the byte matrix (a zero byte there) as the fp32 gradient label_coef[row]
};
constexpr int CEQ8_RG = 4;  // 64-row groups per block: a column's sum meets its fp32 atomic once per 256 rows (38 per column at 2404 rows cost 0.1 ms)
__global__ __launch_bounds__(256) void ce_bwd_q8_kernel(CeQ8Args a) {
  __shared__ __attribute__((aligned(16))) float cs[4][512];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 512;
  const int col = c0 + lane * 8;
  const bool col_ok = col < a.Vpad;  // (Vpad % 8 == 0: a chunk is inside or outside as a whole)
  const float fmax = a.fmt == MIC_E4M3 ? 448.0f : 57344.0f;
  const float inv_denom = a.loss_scale / a.denom[0];
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { a.state[0] = fmax / inv_denom; a.state[1] = inv_denom / fmax; }
  const float conf = 1.0f - a.ls, low = a.ls > 0.f ? a.ls / (float)(a.V - 1) : 0.f;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int rg = 0; rg < CEQ8_RG; ++rg) {
    const int r0 = (blockIdx.y * CEQ8_RG + rg) * 64;
    if (r0 >= a.rows) break;
    uint4 q[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = r0 + wave * 16 + rr;
      q[rr] = make_uint4(0u, 0u, 0u, 0u);
      if (row < a.rows && col_ok) q[rr] = *reinterpret_cast<const uint4*>(a.logits + (size_t)row * a.ld + col);
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = r0 + wave * 16 + rr;
      if (row < a.rows && col_ok) {
        const float w = a.mask[row] ? 1.0f : 0.f;
        const float lse = a.row_lse[row];
        const int label = a.labels[row];
        // (a label outside [0, V) matches no column: no block would write the row's coefficient — the block of column 0 clears it)
        if (a.label_coef != nullptr && col == 0 && (label < 0 || label >= a.V)) a.label_coef[row] = 0.f;
        const uint32_t wd[4] = {q[rr].x, q[rr].y, q[rr].z, q[rr].w};
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = __uint_as_float((i & 1) ? (wd[i >> 1] & 0xffff0000u) : (wd[i >> 1] << 16));
          const int c = col + i;
          const float g = c < a.V ? w * (__expf(x - lse) - (c == label ? conf : low)) : 0.f;
          csum[i] += g;
          o[i] = fminf(fmaxf(g * fmax, -fmax), fmax);  // (|g| <= 1 up to the rounding of lse)
          if (c == label && a.label_coef != nullptr) {
            a.label_coef[row] = g * inv_denom;
            o[i] = 0.f;
          }
        }
        *reinterpret_cast<uint2*>(a.q + (size_t)row * a.ldq + col) = make_uint2(cvt4_fp8(o, a.fmt), cvt4_fp8(o + 4, a.fmt));
      }
    }
  }
  if (a.colsum != nullptr) {
    *reinterpret_cast<float4*>(&cs[wave][lane * 8]) = make_float4(csum[0], csum[1], csum[2], csum[3]);
    *reinterpret_cast<float4*>(&cs[wave][lane * 8 + 4]) = make_float4(csum[4], csum[5], csum[6], csum[7]);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = tid + h * 256;
      const float s4 = (cs[0][c] + cs[1][c]) + (cs[2][c] + cs[3][c]);
      if (c0 + c < a.Vpad && s4 != 0.f) atomicAdd(a.colsum + c0 + c, s4 * inv_denom);
    }
  }
}
extern "C" int mic_ce_bwd_q8(int rows, int V, int Vpad, const void* logits, int ld, const int32_t* labels, const int32_t* mask,
                             float label_smoothing, const float* row_lse, const float* denom, float loss_scale, const mic_fp8_out* q8,
                             float* colsum, float* label_coef, void* stream) {
  MIC_CHECK(rows > 0 && V > 1 && Vpad >= V && Vpad % 8 == 0 && ld >= Vpad && ld % 8 == 0 && logits && labels && mask && row_lse && denom && q8,
            "mic_ce_bwd_q8: bad args");
  MIC_CHECK(q8->q && q8->state && q8->ldq >= Vpad && q8->ldq % 8 == 0 && ((uintptr_t)q8->q & 7) == 0 && ((uintptr_t)logits & 15) == 0 &&
                (q8->fmt == MIC_E4M3 || q8->fmt == MIC_E5M2),
            "mic_ce_bwd_q8: fp8 output [rows][ldq >= Vpad] (ldq %% 8 == 0, 8-B aligned), a scale state, 16-B aligned bf16 logits");
  CeQ8Args a{};
  a.logits = (const uint16_t*)logits; a.ld = ld; a.rows = rows; a.V = V; a.Vpad = Vpad; a.labels = labels; a.mask = mask; a.ls = label_smoothing;
  a.row_lse = row_lse; a.denom = denom; a.loss_scale = loss_scale; a.colsum = colsum;
  a.q = (uint8_t*)q8->q; a.ldq = q8->ldq; a.state = q8->state; a.fmt = q8->fmt; a.label_coef = label_coef;
  hipLaunchKernelGGL(ce_bwd_q8_kernel, dim3((Vpad + 511) / 512, (rows + 64 * CEQ8_RG - 1) / (64 * CEQ8_RG)), dim3(256), 0, (hipStream_t)stream, a);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// The label entries of dlogits, kept out of the fp8 matrix (mic_ce_bwd_q8 label_coef): row m of dlogits has coef[m] at column
// labels[m].  dX[m][:] += coef[m] E[labels[m]][:] goes out as one more fp32 slab for mic_sum_slabs; dE[labels[m]][:] += coef[m] h[m][:]
// by fp32 atomics into the gradient the dE GEMM has just written (rows that share a label — eos, the language ids — meet there).
// One block per row.
__global__ __launch_bounds__(256) void head_label_terms_kernel(int rows, int width, const int32_t* __restrict__ labels, const float* __restrict__ coef,
                                                               const uint16_t* __restrict__ E, int lde, const uint16_t* __restrict__ h, int ldh,
                                                               float* __restrict__ dx_slab, int ldx, float* __restrict__ dE, int ldde) {
  const int m = blockIdx.x;
  const float c = coef[m];
  const int lab = labels[m];
  // consecutive lanes = consecutive columns: a wave's atomic instruction covers two whole 128-B lines of the gradient row (8 columns
  // per lane put 16 lines under every instruction: 79 us for 2404 rows against 20)
  for (int j = threadIdx.x; j < width; j += 256) {
    float o = 0.f;
    if (c != 0.f) {
      o = c * bf2f(E[(size_t)lab * lde + j]);
      atomicAdd(dE + (size_t)lab * ldde + j, c * bf2f(h[(size_t)m * ldh + j]));
    }
    dx_slab[(size_t)m * ldx + j] = o;
  }
}
extern "C" int mic_head_label_terms(int rows, int width, const int32_t* labels, const float* coef, const void* E, int lde, const void* h, int ldh,
                                    float* dx_slab, int ldx, float* dE, int ldde, void* stream) {
  MIC_CHECK(rows > 0 && width > 0 && width % 8 == 0 && labels && coef && E && h && dx_slab && dE && lde % 8 == 0 && ldh % 8 == 0 && ldx % 4 == 0,
            "mic_head_label_terms: bad args (width, lde, ldh multiples of 8; bf16 E and h)");
  hipLaunchKernelGGL(head_label_terms_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, rows, width, labels, coef, (const uint16_t*)E, lde,
                     (const uint16_t*)h, ldh, dx_slab, ldx, dE, ldde);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

// ------------------------------------------------------------------ column sums (bias gradients)
// Table-driven so the 4..7 bias gradients of a layer go out as ONE launch (each alone is a ~9 us launch for ~2 us of
// work).  Each thread owns 8 consecutive columns (one 16-B load per row); a block = 32 column-chunks x 8 row lanes
// covers 256 columns and walks its row slice with stride 8; the 8 row lanes combine through LDS, one fp32 atomic per
// column (outputs live in the pre-zeroed atomic region, or are memset by the single-item entry point).
#define COLSUM_MAX 8
struct ColsumItem { const void* x; float* out; int rows, cols, ld, gx, gy, block_begin; };
struct ColsumTable { int count; ColsumItem it[COLSUM_MAX]; };

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(ColsumTable tab) {
  __shared__ float red[8][256 + 8];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < COLSUM_MAX; ++i)
    if (i < tab.count && (int)blockIdx.x >= tab.it[i].block_begin) pi = i;
  const ColsumItem& I = tab.it[pi];
  const int local = blockIdx.x - I.block_begin;
  const int bx = local % I.gx, by = local / I.gx;
  const T* __restrict__ x = (const T*)I.x;
  const int rows = I.rows, cols = I.cols, ld = I.ld;
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = bx * 256 + cc * 8;
  const int rows_per = (rows + I.gy - 1) / I.gy;
  const int r0 = by * rows_per, r1 = min(rows, r0 + rows_per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 + 8 <= cols) {
    for (int r = r0 + rl; r < r1; r += 8) {
      float v[8];
      ld8(x + (size_t)r * ld + c0, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  } else if (c0 < cols) {
    for (int r = r0 + rl; r < r1; r += 8)
      for (int i = 0; i < cols - c0; ++i) acc[i] += ElemT<T>::ld(x + (size_t)r * ld + c0 + i);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = acc[i];
  __syncthreads();
  const int c = bx * 256 + threadIdx.x;
  if (c < cols) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += red[j][threadIdx.x];
    atomicAdd(I.out + c, s);
  }
}
static int colsum_launch(int dtype, const mic_colsum_item* items, int count, void* stream) {
  ColsumTable tab;
  tab.count = count;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const mic_colsum_item& a = items[i];
    MIC_CHECK(a.rows > 0 && a.cols > 0 && a.x && a.out, "mic_colsum: bad args");
    MIC_CHECK(a.ld % 8 == 0 && ((uintptr_t)a.x & 15) == 0, "mic_colsum: rows must be 16-B aligned");
    ColsumItem& t = tab.it[i];
    t.x = a.x; t.out = a.out; t.rows = a.rows; t.cols = a.cols; t.ld = a.ld;
    t.gx = (a.cols + 255) / 256;
    int gy = (a.rows + 63) / 64;
    const int cap = (1024 + t.gx - 1) / t.gx;
    if (gy > cap) gy = cap;
    t.gy = gy < 1 ? 1 : gy;
    t.block_begin = blocks;
    blocks += t.gx * t.gy;
  }
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    hipLaunchKernelGGL(colsum_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab);
  });
}
// the same for fp8 tensors (fused emission, BASELINE configs[4]): out[c] += scale_inv * sum_r fp8(x[r][c]) — the bias gradient of an fp8
// projection from the e5m2 bytes its dy exists as (no bf16 copy is written any more).  Thread = 8 columns (one 8-B load per row).
struct Colsum8Item { const uint8_t* x; float* out; const float* scale_inv; int rows, cols, ld, fmt, gx, gy, block_begin; };
struct Colsum8Table { int count; Colsum8Item it[COLSUM_MAX]; };
typedef float f32x2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void colsum_q8_kernel(Colsum8Table tab) {
  __shared__ float red[8][256 + 8];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < COLSUM_MAX; ++i)
    if (i < tab.count && (int)blockIdx.x >= tab.it[i].block_begin) pi = i;
  const Colsum8Item& I = tab.it[pi];
  const int local = blockIdx.x - I.block_begin;
  const int bx = local % I.gx, by = local / I.gx;
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = bx * 256 + cc * 8;
  const int rows_per = (I.rows + I.gy - 1) / I.gy;
  const int r0 = by * rows_per, r1 = min(I.rows, r0 + rows_per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < I.cols) {  // cols % 8 == 0 (host check)
    for (int r = r0 + rl; r < r1; r += 8) {
      const uint2 u = *reinterpret_cast<const uint2*>(I.x + (size_t)r * I.ld + c0);
      const int w[2] = {(int)u.x, (int)u.y};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x2v lo, hi;
        if (I.fmt == MIC_E4M3) { lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[h], false); hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[h], true); }
        else { lo = __builtin_amdgcn_cvt_pk_f32_bf8(w[h], false); hi = __builtin_amdgcn_cvt_pk_f32_bf8(w[h], true); }
        acc[4 * h + 0] += lo[0]; acc[4 * h + 1] += lo[1]; acc[4 * h + 2] += hi[0]; acc[4 * h + 3] += hi[1];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = acc[i];
  __syncthreads();
  const int c = bx * 256 + threadIdx.x;
  if (c < I.cols) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][threadIdx.x];
    atomicAdd(I.out + c, t * I.scale_inv[0]);
  }
}
extern "C" int mic_colsum_q8_grouped(const mic_colsum_q8_item* items, int count, void* stream) {
  MIC_CHECK(items && count >= 1, "mic_colsum_q8_grouped: bad args");
  for (int i0 = 0; i0 < count; i0 += COLSUM_MAX) {
    Colsum8Table tab;
    tab.count = count - i0 < COLSUM_MAX ? count - i0 : COLSUM_MAX;
    int blocks = 0;
    for (int i = 0; i < tab.count; ++i) {
      const mic_colsum_q8_item& a = items[i0 + i];
      MIC_CHECK(a.rows > 0 && a.cols > 0 && a.x && a.out && a.scale_inv, "mic_colsum_q8_grouped: bad item %d", i0 + i);
      MIC_CHECK(a.cols % 8 == 0 && a.ld % 8 == 0 && ((uintptr_t)a.x & 7) == 0 && (a.fmt == MIC_E4M3 || a.fmt == MIC_E5M2), "mic_colsum_q8_grouped: cols, ld multiples of 8; fmt");
      Colsum8Item& t = tab.it[i];
      t.x = (const uint8_t*)a.x; t.out = a.out; t.scale_inv = a.scale_inv; t.rows = a.rows; t.cols = a.cols; t.ld = a.ld; t.fmt = a.fmt;
      t.gx = (a.cols + 255) / 256;
      int gy = (a.rows + 63) / 64;
      const int cap = (1024 + t.gx - 1) / t.gx;
      if (gy > cap) gy = cap;
      t.gy = gy < 1 ? 1 : gy;
      t.block_begin = blocks;
      blocks += t.gx * t.gy;
    }
    hipLaunchKernelGGL(colsum_q8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab);
    MIC_LAUNCH_CHECK();
  }
  return MIC_OK;
}
extern "C" int mic_colsum(int dtype, int rows, int cols, const void* x, int ld, float* out, int accumulate, void* stream) {
  MIC_CHECK(rows > 0 && cols > 0 && x && out, "mic_colsum: bad args");
  if (!accumulate) hipMemsetAsync(out, 0, sizeof(float) * cols, (hipStream_t)stream);
  mic_colsum_item it = {x, out, rows, cols, ld};
  return colsum_launch(dtype, &it, 1, stream);
}
extern "C" int mic_colsum_grouped(int dtype, const mic_colsum_item* items, int count, void* stream) {
  MIC_CHECK(items && count >= 1, "mic_colsum_grouped: bad args");
  for (int i = 0; i < count; i += COLSUM_MAX) {
    const int n = count - i < COLSUM_MAX ? count - i : COLSUM_MAX;
    if (int rc = colsum_launch(dtype, items + i, n, stream)) return rc;
  }
  return MIC_OK;
}

// ------------------------------------------------------------------ dropout mask, casts
__global__ void dropout_mask_kernel(uint8_t* out, long n, uint32_t thr, uint32_t seed) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = dropout_keep(seed, (uint32_t)e, thr) ? 1 : 0;
}
extern "C" int mic_dropout_mask(uint8_t* out, int64_t n, float p, uint32_t seed, void* stream) {
  MIC_CHECK(out && n > 0 && p >= 0.f && p < 1.f, "mic_dropout_mask: bad args");
  const uint32_t thr = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, (long)n, thr, seed);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
template <typename S, typename D>
__global__ void cast2d_kernel(int rows, int cols, const S* __restrict__ src, long ld_src, D* __restrict__ dst, long ld_dst) {
  const long total = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / cols, c = e % cols;
    ElemT<D>::st(dst + r * ld_dst + c, ElemT<S>::ld(src + r * ld_src + c));
  }
}
extern "C" int mic_cast2d(int src_dtype, int dst_dtype, int rows, int cols, const void* src, int ld_src, void* dst,
                          int ld_dst, void* stream) {
  MIC_CHECK(rows > 0 && cols > 0 && src && dst, "mic_cast2d: bad args");
  const long total = (long)rows * cols;
  int nb = (int)((total + 255) / 256); if (nb > 4096) nb = 4096;
  dim3 grid(nb), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == MIC_F32 && dst_dtype == MIC_BF16) hipLaunchKernelGGL((cast2d_kernel<float, uint16_t>), grid, block, 0, s, rows, cols, (const float*)src, (long)ld_src, (uint16_t*)dst, (long)ld_dst);
  else if (src_dtype == MIC_BF16 && dst_dtype == MIC_F32) hipLaunchKernelGGL((cast2d_kernel<uint16_t, float>), grid, block, 0, s, rows, cols, (const uint16_t*)src, (long)ld_src, (float*)dst, (long)ld_dst);
  else if (src_dtype == MIC_F32 && dst_dtype == MIC_F32) hipLaunchKernelGGL((cast2d_kernel<float, float>), grid, block, 0, s, rows, cols, (const float*)src, (long)ld_src, (float*)dst, (long)ld_dst);
  else if (src_dtype == MIC_BF16 && dst_dtype == MIC_BF16) hipLaunchKernelGGL((cast2d_kernel<uint16_t, uint16_t>), grid, block, 0, s, rows, cols, (const uint16_t*)src, (long)ld_src, (uint16_t*)dst, (long)ld_dst);
  else MIC_CHECK(false, "mic_cast2d: bad dtypes");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
// ------------------------------------------------------------------ split-K second half: sum fp32 slabs
template <typename D>
__global__ __launch_bounds__(256) void sum_slabs_kernel(int n_slabs, long slab_stride, int rows, int cols, const float* __restrict__ src,
                                                        long ld_src, D* __restrict__ dst, long ld_dst) {
  const int cv = cols >> 3;  // 8-column groups (cols % 8 == 0)
  const long total = (long)rows * cv;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / cv, c = (e % cv) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* p = src + r * ld_src + c;
    for (int s = 0; s < n_slabs; ++s) {
      float v[8];
      ld8(p + (long)s * slab_stride, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
    st8(dst + r * ld_dst + c, acc);
  }
}
extern "C" int mic_sum_slabs(int dst_dtype, int n_slabs, long long slab_stride, int rows, int cols, const float* src, int ld_src,
                             void* dst, int ld_dst, void* stream) {
  MIC_CHECK(n_slabs > 0 && rows > 0 && cols > 0 && cols % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 && slab_stride % 4 == 0 && src && dst,
            "mic_sum_slabs: bad args");
  const long total = (long)rows * (cols >> 3);
  int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
  if (dst_dtype == MIC_BF16)
    hipLaunchKernelGGL(sum_slabs_kernel<uint16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n_slabs, (long)slab_stride, rows, cols, src, (long)ld_src, (uint16_t*)dst, (long)ld_dst);
  else if (dst_dtype == MIC_F32)
    hipLaunchKernelGGL(sum_slabs_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n_slabs, (long)slab_stride, rows, cols, src, (long)ld_src, (float*)dst, (long)ld_dst);
  else MIC_CHECK(false, "mic_sum_slabs: bad dtype");
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_zero(void* p, int64_t bytes, void* stream) {
  MIC_CHECK(p && bytes > 0, "mic_zero: bad args");
  hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) { mic_set_error("mic_zero: %s", hipGetErrorString(e)); return MIC_ELAUNCH; }
  return MIC_OK;
}

// A stream whose kernels run on bits [first_cu, first_cu + n_cus) of the device's CU mask only.  On MI355X consecutive mask bits go
// round the 8 XCDs (bit i -> XCD i % 8), so 64 consecutive bits are 8 CUs on every XCD (tools/probe_cu_mask.hip).  The optimizer's per-bucket
// launches use such a stream: bandwidth-bound work on a few dedicated CUs beside the backward GEMMs, whose blocks own a CU's
// whole register file and therefore never share one with anything else.
extern "C" int mic_stream_create_cu_masked(int first_cu, int n_cus, void** stream) {
  MIC_CHECK(stream != nullptr && first_cu >= 0 && n_cus > 0, "mic_stream_create_cu_masked: bad args");
  int dev = 0, total = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) { mic_set_error("mic_stream_create_cu_masked: %s", hipGetErrorString(e)); return MIC_ELAUNCH; }
  MIC_CHECK(first_cu + n_cus <= total && n_cus < total && total <= 1024, "mic_stream_create_cu_masked: CUs [%d, %d) of %d", first_cu, first_cu + n_cus, total);
  uint32_t mask[32] = {};
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
  if (e != hipSuccess) { mic_set_error("mic_stream_create_cu_masked: %s", hipGetErrorString(e)); return MIC_ELAUNCH; }
  *stream = (void*)s;
  return MIC_OK;
}

extern "C" int mic_stream_destroy(void* stream) {
  MIC_CHECK(stream != nullptr, "mic_stream_destroy: null stream");
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { mic_set_error("mic_stream_destroy: %s", hipGetErrorString(e)); return MIC_ELAUNCH; }
  return MIC_OK;
}

// ------------------------------------------------------------------ emulated collective (bench.py --emulate-comm)
__global__ __launch_bounds__(256) void comm_emulate_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16, unsigned long long ticks) {
  // copy src -> dst (local HBM reads and writes, as the reduce / gather steps of a ring all-reduce make them) for AS LONG AS the
  // projected duration allows — the copy is cut when the time is up, the kernel never runs longer than asked — then hold the CUs
  const unsigned long long t0 = wall_clock64();
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int pass = 0; pass < 2; ++pass) {
    for (; i + 3 * stride < n16; i += 4 * stride) {  // four 16-B loads in flight per lane
      if (wall_clock64() - t0 >= ticks) return;
      const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
      dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  }
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);  // bounded by `ticks` (checked on the host side)
}

extern "C" int mic_comm_emulate(const void* src, void* dst, int64_t bytes, float micros, int blocks, void* stream) {
  MIC_CHECK(src && dst && bytes >= 0 && blocks > 0 && blocks <= 4096, "mic_comm_emulate: bad args");
  MIC_CHECK(micros >= 0.f && micros <= 200000.f, "mic_comm_emulate: micros=%g (0..200000)", (double)micros);
  MIC_CHECK((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "mic_comm_emulate: 16-B aligned buffers");
  int dev = 0, khz = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (e != hipSuccess || khz <= 0) khz = 100000;  // 100 MHz constant clock on MI300-class parts
  const unsigned long long ticks = (unsigned long long)((double)micros * 1e-3 * (double)khz);
  hipLaunchKernelGGL(comm_emulate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, (long)(bytes / 16), ticks);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
  MIC_CHECK(n > 0 && n < (1LL << 40), "mic_cast: bad n");
  // split into rows of <= 2^20 so the 2-D kernel's int shape holds
  const int64_t cols = n < (1 << 20) ? n : (1 << 20);
  const int64_t rows = n / cols;
  int rc = mic_cast2d(src_dtype, dst_dtype, (int)rows, (int)cols, src, (int)cols, dst, (int)cols, stream);
  if (rc) return rc;
  const int64_t rem = n - rows * cols;
  if (rem > 0) {
    const size_t ss = src_dtype == MIC_F32 ? 4 : 2, ds = dst_dtype == MIC_F32 ? 4 : 2;
    return mic_cast2d(src_dtype, dst_dtype, 1, (int)rem, (const char*)src + rows * cols * ss, (int)rem, (char*)dst + rows * cols * ds, (int)rem, stream);
  }
  return MIC_OK;
}

// ------------------------------------------------------------------ row gather / scatter
// dst[dst_idx ? dst_idx[i] : i][:] = src[src_idx ? src_idx[i] : i][:]  for i < n   (16-B vectors)
template <typename T>
__global__ void copy_rows_kernel(int n, int width, const T* __restrict__ src, int ld_src, const int32_t* __restrict__ src_idx,
                                 T* __restrict__ dst, int ld_dst, const int32_t* __restrict__ dst_idx) {
  constexpr int VEC = 16 / sizeof(T);
  const int nchunk = width / VEC;
  const long total = (long)n * nchunk;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(e % nchunk);
    const int i = (int)(e / nchunk);
    const int rs = src_idx ? src_idx[i] : i, rd = dst_idx ? dst_idx[i] : i;
    *reinterpret_cast<uint4*>(dst + (size_t)rd * ld_dst + ch * VEC) = *reinterpret_cast<const uint4*>(src + (size_t)rs * ld_src + ch * VEC);
  }
}
extern "C" int mic_copy_rows(int dtype, int n, int width, const void* src, int ld_src, const int32_t* src_idx, void* dst, int ld_dst,
                             const int32_t* dst_idx, void* stream) {
  MIC_CHECK(n > 0 && width > 0 && src && dst, "mic_copy_rows: bad args");
  const int vec = dtype == MIC_BF16 ? 8 : 4;
  MIC_CHECK(width % vec == 0 && ld_src % vec == 0 && ld_dst % vec == 0, "mic_copy_rows: rows must be 16-B multiples");
  return dispatch_t(dtype, [&](auto* tag) {
    using T = TYPE_OF(tag);
    const long total = (long)n * (width / vec);
    int nb = (int)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, width, (const T*)src, ld_src, src_idx, (T*)dst, ld_dst, dst_idx);
  });
}

// ------------------------------------------------------------------ fused AdamW over the flat parameter buffer (K15)
// ROWS: the buffer is [rows][width] and only the rows with (row_flag[row] != 0) == want are updated — the tied embedding's
// AdamW in two passes (mic_adamw_rows): AdamW is elementwise, so the split is exact.
template <bool ROWS>
__global__ __launch_bounds__(256) void adamw_kernel(long n, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ g, uint16_t* __restrict__ p_lp,
                                                    const float* __restrict__ hyper, float b1, float b2, float omb1, float omb2,
                                                    float eps, float wd, float gscale, const uint8_t* __restrict__ row_flag,
                                                    int row_f4, int want) {
  const float lr = hyper[0], t = hyper[1];
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const long n4 = n >> 2;
  // ROWS: a block walks whole rows (one flag test per row and block, uniform), its threads the row's float4s
  const long outer_n = ROWS ? n4 / row_f4 : n4;
  const long outer_0 = ROWS ? blockIdx.x : (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long outer_step = ROWS ? gridDim.x : (long)gridDim.x * blockDim.x;
  for (long o = outer_0; o < outer_n; o += outer_step) {
    if constexpr (ROWS) {
      if ((row_flag[o] != 0) != (want != 0)) continue;
    }
    for (long e = ROWS ? o * row_f4 + threadIdx.x : o; e < (ROWS ? (o + 1) * row_f4 : o + 1); e += ROWS ? blockDim.x : 1) {
    float4 pp = reinterpret_cast<float4*>(p)[e], mm = reinterpret_cast<float4*>(m)[e], vv = reinterpret_cast<float4*>(v)[e];
    const float4 gg = reinterpret_cast<const float4*>(g)[e];
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ga[4] = {gg.x * gscale, gg.y * gscale, gg.z * gscale, gg.w * gscale};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ma[i] = b1 * ma[i] + omb1 * ga[i];
      va[i] = b2 * va[i] + omb2 * ga[i] * ga[i];
      const float upd = (ma[i] / bc1) / (sqrtf(va[i] / bc2) + eps) + wd * pa[i];
      pa[i] -= lr * upd;
    }
    reinterpret_cast<float4*>(p)[e] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    reinterpret_cast<float4*>(m)[e] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    reinterpret_cast<float4*>(v)[e] = make_float4(va[0], va[1], va[2], va[3]);
    if (p_lp) {
      uint2 q;
      q.x = f2bf_pk(pa[0], pa[1]);
      q.y = f2bf_pk(pa[2], pa[3]);
      reinterpret_cast<uint2*>(p_lp)[e] = q;
    }
    }
  }
}
extern "C" int mic_adamw(int64_t n, float* p, float* m, float* v, const float* g, void* p_lp, const float* hyper, double b1,
                         double b2, double eps, double wd, float grad_scale, void* stream) {
  MIC_CHECK(n > 0 && n % 4 == 0 && p && m && v && g && hyper, "mic_adamw: bad args (n must be a multiple of 4)");
  long nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(adamw_kernel<false>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (long)n, p, m, v, g, (uint16_t*)p_lp, hyper, (float)b1, (float)b2,
                     (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, grad_scale, (const uint8_t*)nullptr, 1, 0);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

extern "C" int mic_adamw_rows(int64_t rows, int width, const uint8_t* row_flag, int want, float* p, float* m, float* v, const float* g,
                              void* p_lp, const float* hyper, double b1, double b2, double eps, double wd, float grad_scale, void* stream) {
  MIC_CHECK(rows > 0 && width > 0 && width % 4 == 0 && row_flag && p && m && v && g && hyper, "mic_adamw_rows: bad args (width must be a multiple of 4)");
  const long n = (long)rows * width;
  long nb = (n / 4 + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(adamw_kernel<true>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, n, p, m, v, g, (uint16_t*)p_lp, hyper, (float)b1, (float)b2,
                     (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, grad_scale, row_flag, width / 4, want);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}

__global__ __launch_bounds__(256) void row_flags_kernel(const int32_t* __restrict__ ids, int n_ids, uint8_t* __restrict__ flags, int n_rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_ids) {
    const int r = ids[i];
    if (r >= 0 && r < n_rows) flags[r] = 1;
  }
}
extern "C" int mic_row_flags(const int32_t* ids, int n_ids, uint8_t* flags, int n_rows, void* stream) {
  MIC_CHECK(ids && flags && n_ids > 0 && n_rows > 0, "mic_row_flags: bad args");
  hipError_t e = hipMemsetAsync(flags, 0, (size_t)n_rows, (hipStream_t)stream);
  if (e != hipSuccess) { mic_set_error("mic_row_flags: %s", hipGetErrorString(e)); return MIC_ELAUNCH; }
  hipLaunchKernelGGL(row_flags_kernel, dim3((n_ids + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, n_ids, flags, n_rows);
  MIC_LAUNCH_CHECK();
  return MIC_OK;
}
