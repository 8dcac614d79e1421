// Backward-pass support kernels (HBM-bound, NHWC, 16 bytes per lane where the layout allows):
// data-gradient weight packing, stride-2 scatter, ReLU masks, top-down (nearest 2x) backward, correlation backward,
// ROIAlign backward, GroupNorm(+ReLU) backward, naive strided data gradient for the two tiny 3x3/2 convs (P6/P7),
// weight-gradient unpacking (packed [Cout][R][S][Cin] fp32 -> OIHW, FrozenBN scale applied).
#include "osd_common.h"
#include "conv_params.h"

namespace {

constexpr int kMaxBlocks = 2048;
inline int grid_for(long long work, int threads) {
  long long b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Chunk<__bf16> {
  static constexpr int N = 8;
  float v[8];
  __device__ __forceinline__ void load(const __bf16* p) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ __forceinline__ void store(__bf16* p) const {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// Wd[ci][r'][s'][co] = w[co][ci][R-1-r'][S-1-s'] * scale[co]   (data gradient of a stride-1 conv = conv with these)
template <typename T>
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                         int cout, int cin, int R, int S, int rows, int cout_pad, int src_orsi) {
  const long long total = (long long)rows * R * S * cout_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout_pad);
    long long t = i / cout_pad;
    const int s = (int)(t % S); t /= S;
    const int r = (int)(t % R);
    const int ci = (int)(t / R);
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = src_orsi ? w[(((size_t)co * R + (R - 1 - r)) * S + (S - 1 - s)) * cin + ci]
                   : w[(((size_t)co * cin + ci) * R + (R - 1 - r)) * S + (S - 1 - s)];
      if (scale) v *= scale[co];
    }
    dst[i] = from_f32<T>(v);
  }
}

// ---- all weights of the model repacked by ONE launch per form (forward / data-gradient): table driven ----
struct PackEntry {
  long long src_off;     // floats into the flat fp32 master buffer ([cout][R][S][cin] order)
  long long dst_off;     // elements into the flat packed buffer
  long long scale_off;   // floats into the flat scale buffer, -1: none
  int cout, cin, R, S;
  int rows, kpad;        // packed geometry: forward [rows >= cout][R][S][kpad >= cin]; dgrad [rows >= cin][R][S][kpad >= cout]
  int first_block, n_blocks;
};

template <typename T>
__global__ void __launch_bounds__(256) pack_multi_kernel(const PackEntry* __restrict__ table, const int* __restrict__ block_entry,
                                                         const float* __restrict__ src, const float* __restrict__ scales,
                                                         T* __restrict__ dst, int dgrad) {
  const PackEntry e = table[block_entry[blockIdx.x]];
  const float* w = src + e.src_off;
  const float* sc = e.scale_off >= 0 ? scales + e.scale_off : nullptr;
  T* out = dst + e.dst_off;
  constexpr int EP = 16 / (int)sizeof(T);     // elements of one 16-byte output chunk
  if (!dgrad && e.cin % EP == 0 && e.kpad % EP == 0 && (e.src_off & 3) == 0) {
    // forward form, 16 bytes out per lane: a chunk lies entirely inside or entirely outside the real channels
    const int kch = e.kpad / EP;
    const int total = e.rows * e.R * e.S * kch;
    const int stride = e.n_blocks * blockDim.x;
    const int rs_c = e.R * e.S * kch;
    for (int i = (blockIdx.x - e.first_block) * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int row = i / rs_c, rem = i - row * rs_c;
      const int tap = rem / kch, k = (rem - tap * kch) * EP;
      float v[EP];
#pragma unroll
      for (int q = 0; q < EP; ++q) v[q] = 0.f;
      if (row < e.cout && k < e.cin) {
        const f32x4* sp = reinterpret_cast<const f32x4*>(w + ((size_t)row * e.R * e.S + tap) * e.cin + k);
        const float f = sc ? sc[row] : 1.f;
#pragma unroll
        for (int q = 0; q < EP; q += 4) {
          const f32x4 x = sp[q / 4];
          v[q] = x[0] * f; v[q + 1] = x[1] * f; v[q + 2] = x[2] * f; v[q + 3] = x[3] * f;
        }
      }
      if constexpr (sizeof(T) == 2) {
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (__bf16)v[q];
        *reinterpret_cast<bf16x8*>(out + (size_t)i * EP) = o;
      } else {
        *reinterpret_cast<f32x4*>(out + (size_t)i * EP) = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
    return;
  }
  if (!dgrad) {              // forward form: same [row = co][r][s][k = ci] order as the master -> a scaled, padded copy
    const int total = e.rows * e.R * e.S * e.kpad;
    const int stride = e.n_blocks * blockDim.x;
    const int rs_k = e.R * e.S * e.kpad;
    for (int i = (blockIdx.x - e.first_block) * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int row = i / rs_k, rem = i - row * rs_k;
      const int tap = rem / e.kpad, k = rem - tap * e.kpad;
      float v = 0.f;
      if (row < e.cout && k < e.cin) {
        v = w[((size_t)row * e.R * e.S + tap) * e.cin + k];
        if (sc) v *= sc[row];
      }
      out[i] = from_f32<T>(v);
    }
    return;
  }
  // data-gradient form [row = ci][r'][s'][k = co] with flipped taps: a transpose of the master per tap through LDS.
  // 64 x 64 tiles: 16-byte loads along ci, 16-byte stores along co
  if (e.cin % 4 == 0 && e.kpad % 64 == 0 && (e.src_off & 3) == 0) {
    __shared__ float big[64][65];
    const int tci = (e.rows + 63) / 64, tco = e.kpad / 64;
    const int ntiles = e.R * e.S * tci * tco;
    for (int t = blockIdx.x - e.first_block; t < ntiles; t += e.n_blocks) {
      const int cot = t % tco;
      int u = t / tco;
      const int cit = u % tci; u /= tci;
      const int ss = u % e.S, rr = u / e.S;
      const int co0 = cot * 64, ci0 = cit * 64;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + (threadIdx.x >> 4) + 16 * j, ci = ci0 + (threadIdx.x & 15) * 4;
        f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
        if (co < e.cout && ci < e.cin) {
          x = *reinterpret_cast<const f32x4*>(w + (((size_t)co * e.R + (e.R - 1 - rr)) * e.S + (e.S - 1 - ss)) * e.cin + ci);
          if (sc) { const float f = sc[co]; x[0] *= f; x[1] *= f; x[2] *= f; x[3] *= f; }
        }
        float* trow = &big[(threadIdx.x >> 4) + 16 * j][(threadIdx.x & 15) * 4];
        trow[0] = x[0]; trow[1] = x[1]; trow[2] = x[2]; trow[3] = x[3];
      }
      __syncthreads();
      constexpr int CPW = 64 / EP;                       // output chunks per ci row of the tile
#pragma unroll
      for (int j = 0; j < (64 * CPW) / 256; ++j) {
        const int idx = threadIdx.x + 256 * j;
        const int ci = ci0 + idx / CPW, cc = (idx % CPW) * EP;
        if (ci >= e.rows) continue;
        T* o = out + (((size_t)ci * e.R + rr) * e.S + ss) * e.kpad + co0 + cc;
        if constexpr (sizeof(T) == 2) {
          bf16x8 v;
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = (__bf16)big[cc + q][idx / CPW];
          *reinterpret_cast<bf16x8*>(o) = v;
        } else {
          *reinterpret_cast<f32x4*>(o) = f32x4{big[cc][idx / CPW], big[cc + 1][idx / CPW], big[cc + 2][idx / CPW], big[cc + 3][idx / CPW]};
        }
      }
    }
    return;
  }
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int tci = (e.rows + 31) / 32, tco = (e.kpad + 31) / 32;
  const int ntiles = e.R * e.S * tci * tco;
  for (int t = blockIdx.x - e.first_block; t < ntiles; t += e.n_blocks) {
    const int cot = t % tco;
    int u = t / tco;
    const int cit = u % tci; u /= tci;
    const int s = u % e.S, r = u / e.S;
    const int co0 = cot * 32, ci0 = cit * 32;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + ty + 8 * j, ci = ci0 + tx;
      float v = 0.f;
      if (co < e.cout && ci < e.cin) {
        v = w[(((size_t)co * e.R + (e.R - 1 - r)) * e.S + (e.S - 1 - s)) * e.cin + ci];
        if (sc) v *= sc[co];
      }
      tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = ci0 + ty + 8 * j, co = co0 + tx;
      if (ci < e.rows && co < e.kpad) out[(((size_t)ci * e.R + r) * e.S + s) * e.kpad + co] = from_f32<T>(tile[tx][ty + 8 * j]);
    }
  }
}

// ---- SGD with momentum over the flat master buffer, ONE launch (torch.optim.SGD semantics, dampening 0, no nesterov:
// g += wd*p; buf = first ? g : momentum*buf + g; p -= lr*buf), per-tensor lr multiplier and weight decay from a table
// (solver/build.py:8-26: biases get lr x2 and no weight decay) ----
struct SgdEntry {
  long long off, numel;
  float lr_mult, wd;
  int first_block, n_blocks;
};

__global__ void __launch_bounds__(256) sgd_multi_kernel(const SgdEntry* __restrict__ table, const int* __restrict__ block_entry,
                                                        float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ buf, float lr, float momentum, int first) {
  const SgdEntry e = table[block_entry[blockIdx.x]];
  const float step = lr * e.lr_mult;
  const long long stride = (long long)e.n_blocks * blockDim.x;
  const long long tid = (long long)(blockIdx.x - e.first_block) * blockDim.x + threadIdx.x;
  // 16 bytes per lane (3 loads + 2 stores of 16 B per 4 parameters) when the tensor starts on a 16-byte boundary
  const long long n4 = (e.off & 3) == 0 ? (e.numel >> 2) : 0;
  f32x4* p4 = reinterpret_cast<f32x4*>(p + e.off);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g + e.off);
  f32x4* b4 = reinterpret_cast<f32x4*>(buf + e.off);
  for (long long i = tid; i < n4; i += stride) {
    const f32x4 w = p4[i], gr = g4[i];
    f32x4 m;
    if (first) {
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = gr[k] + e.wd * w[k];
    } else {
      const f32x4 bb = b4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = momentum * bb[k] + (gr[k] + e.wd * w[k]);
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = w[k] - step * m[k];
    b4[i] = m;
    p4[i] = o;
  }
  for (long long i = n4 * 4 + tid; i < e.numel; i += stride) {
    const long long k = e.off + i;
    const float w = p[k];
    const float d = g[k] + e.wd * w;
    const float m = first ? d : momentum * buf[k] + d;
    buf[k] = m;
    p[k] = w - step * m;
  }
}

// ---- the same update that ALSO writes the forward-form packed weights of the conv tensors it updates (round 4): the repack
// of the forward form is a scaled, rounded, K-padded copy in the masters' own [cout][R][S][cin] order, so the thread that holds
// the four updated fp32 values stores them again as `T`, times the folded FrozenBN scale — the packed buffer's padding rows /
// columns were zeroed once and are never written.  One launch and one read of the masters less per bucket; the data-gradient
// form (a per-tap transpose) stays pack_multi_kernel's.  Entries with dst_off < 0 (biases, GroupNorm affine, Scale) only update.
struct SgdPackEntry {
  long long off, numel;
  float lr_mult, wd;
  int first_block, n_blocks;
  long long dst_off;     // elements into the flat packed buffer; -1: not a conv weight
  long long scale_off;   // floats into the flat scale buffer; -1: none
  int cin, rs, kpad, pad_;   // [cout][rs = R * S][cin] master -> [rows][rs][kpad] packed.  cin % 4 == 0 and off % 4 == 0 take the
                             // 16-byte path; anything else falls back to one value per lane IN THE KERNEL (n4 = 0).  The host
                             // (train_update._build_sgd_table) checks kpad >= cin and that the packed rows end inside the buffer;
                             // the packed buffer's padding is zero from the initial pack and is never written here
};

template <typename T>
__global__ void __launch_bounds__(256) sgd_pack_multi_kernel(const SgdPackEntry* __restrict__ table, const int* __restrict__ block_entry,
                                                             float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                             const float* __restrict__ scales, T* __restrict__ packed, float lr,
                                                             float momentum, int first, int zero_grads) {
  // zero_grads: the gradient is CONSUMED — zeros are stored behind the read, so the next step's weight-gradient kernels (which
  // accumulate atomically) find a clean buffer without a 236 MB memset on the critical path of the next forward pass
  const SgdPackEntry e = table[block_entry[blockIdx.x]];
  const float step = lr * e.lr_mult;
  const long long stride = (long long)e.n_blocks * blockDim.x;
  const long long tid = (long long)(blockIdx.x - e.first_block) * blockDim.x + threadIdx.x;
  const bool pack = e.dst_off >= 0;
  // four values per thread where the tensor starts on a 16-byte boundary (and, packed, a group of four stays inside one tap's
  // channel run); otherwise — the 2- and 4-row prediction convs behind an odd-sized bias — one value per thread
  const long long n4 = ((e.off & 3) == 0 && (!pack || (e.cin & 3) == 0)) ? (e.numel >> 2) : 0;
  f32x4* p4 = reinterpret_cast<f32x4*>(p + e.off);
  f32x4* g4 = reinterpret_cast<f32x4*>(g + e.off);
  f32x4* b4 = reinterpret_cast<f32x4*>(buf + e.off);
  const float* sc = e.scale_off >= 0 ? scales + e.scale_off : nullptr;
  const int cin4 = e.cin >> 2;
  const long long row4 = (long long)e.rs * cin4;            // groups of four per output row (co)
  for (long long i = tid; i < n4; i += stride) {
    const f32x4 w = p4[i], gr = g4[i];
    f32x4 m;
    if (first) {
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = gr[k] + e.wd * w[k];
    } else {
      const f32x4 bb = b4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = momentum * bb[k] + (gr[k] + e.wd * w[k]);
    }
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = w[k] - step * m[k];
    b4[i] = m;
    p4[i] = o;
    if (zero_grads) g4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pack) {
      const long long co = i / row4;
      const long long rem = i - co * row4;
      const int tap = (int)(rem / cin4), ci = ((int)(rem - (long long)tap * cin4)) << 2;
      const float f = sc ? sc[co] : 1.f;
      T* d = packed + e.dst_off + ((size_t)co * e.rs + tap) * e.kpad + ci;
      if constexpr (sizeof(T) == 2) {
        bf16x4 q;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (__bf16)(o[k] * f);
        *reinterpret_cast<bf16x4*>(d) = q;
      } else {
        *reinterpret_cast<f32x4*>(d) = f32x4{o[0] * f, o[1] * f, o[2] * f, o[3] * f};
      }
    }
  }
  const long long row1 = (long long)e.rs * e.cin;
  for (long long i = n4 * 4 + tid; i < e.numel; i += stride) {
    const long long k = e.off + i;
    const float w = p[k];
    const float d = g[k] + e.wd * w;
    const float m = first ? d : momentum * buf[k] + d;
    const float o = w - step * m;
    buf[k] = m;
    p[k] = o;
    if (zero_grads) g[k] = 0.f;
    if (pack) {
      const long long co = i / row1;
      const long long rem = i - co * row1;
      const int tap = (int)(rem / e.cin), ci = (int)(rem - (long long)tap * e.cin);
      packed[e.dst_off + ((size_t)co * e.rs + tap) * e.kpad + ci] = from_f32<T>(o * (sc ? sc[co] : 1.f));
    }
  }
}

// packed fp32 weight gradient [cout][R][S][cin] -> OIHW, times the folded FrozenBN scale (d/dw of conv(x, w*scale))
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, const float* __restrict__ scale, float* __restrict__ g,
                                    int cout, int cin, int R, int S, int accumulate) {
  const long long total = (long long)cout * cin * R * S;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(i % S);
    long long t = i / S;
    const int r = (int)(t % R); t /= R;
    const int ci = (int)(t % cin);
    const int co = (int)(t / cin);
    float v = dwp[(((size_t)co * R + r) * S + s) * cin + ci];
    if (scale) v *= scale[co];
    g[i] = accumulate ? g[i] + v : v;
  }
}

// dst[n, 2*ho, 2*wo, c] = mask > 0 ? src[n, ho, wo, c] : 0 ; every other position 0   (data gradient of a 1x1/2 conv)
template <typename T>
__global__ void scatter2x_kernel(const T* __restrict__ src, const T* __restrict__ mask, const T* __restrict__ addend,
                                 T* __restrict__ dst, int n, int h, int w, int ho, int wo, int c) {
  constexpr int E = Chunk<T>::N;
  const int cch = c / E;
  const long long total = (long long)n * h * w * cch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cch);
    long long t = i / cch;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    Chunk<T> v;
#pragma unroll
    for (int e = 0; e < E; ++e) v.v[e] = 0.f;
    if (!(x & 1) && !(y & 1) && (y >> 1) < ho && (x >> 1) < wo)
      v.load(src + (((size_t)b * ho + (y >> 1)) * wo + (x >> 1)) * c + cc * E);
    if (addend) {
      Chunk<T> a;
      a.load(addend + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) v.v[e] += a.v[e];
    }
    if (mask) {
      Chunk<T> m;
      m.load(mask + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) v.v[e] = m.v[e] > 0.f ? v.v[e] : 0.f;
    }
    v.store(dst + i * E);
  }
}

// out = (a [+ b]) masked by (mask > 0)      (ReLU backward, optionally summing two gradient branches)
template <typename T>
__global__ void add_mask_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ mask, T* __restrict__ out,
                                long long chunks) {
  constexpr int E = Chunk<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
    Chunk<T> v;
    v.load(a + i * E);
    if (b) {
      Chunk<T> w;
      w.load(b + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) v.v[e] += w.v[e];
    }
    if (mask) {
      Chunk<T> m;
      m.load(mask + i * E);
#pragma unroll
      for (int e = 0; e < E; ++e) v.v[e] = m.v[e] > 0.f ? v.v[e] : 0.f;
    }
    v.store(out + i * E);
  }
}

// nearest-2x upsample backward: top[n,y,x,c] (+)= sum_{i,j<2} inner[n,2y+i,2x+j,c]      (fpn.py:59-64)
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ inner, const T* __restrict__ prev, T* __restrict__ top, int n,
                                      int h, int w, int c) {
  constexpr int E = Chunk<T>::N;
  const int cch = c / E;
  const long long total = (long long)n * h * w * cch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cch);
    long long t = i / cch;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    Chunk<T> s;
    if (prev) s.load(prev + i * E);
    else {
#pragma unroll
      for (int e = 0; e < E; ++e) s.v[e] = 0.f;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        Chunk<T> v;
        v.load(inner + (((size_t)b * 2 * h + 2 * y + dy) * 2 * w + 2 * x + dx) * c + cc * E);
#pragma unroll
        for (int e = 0; e < E; ++e) s.v[e] += v.v[e];
      }
    s.store(top + i * E);
  }
}

// correlation backward wrt the query: dq[n][c] = sum_p g[n,p,c] * feat[n,p,c]   (one workgroup per (image, slab); atomics)
template <typename T>
__global__ void __launch_bounds__(256) correlate_bwd_q_kernel(const T* __restrict__ g, const T* __restrict__ feat,
                                                              float* __restrict__ dq, int hw, int c) {
  constexpr int E = Chunk<T>::N;
  const int cch = c / E;
  const int lanes = 256 / cch;
  const int img = blockIdx.y;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int per = (hw + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  if (pl < lanes && p0 < p1) {
    constexpr int U = 4;            // 8 x 16-byte loads in flight per thread (unconditional, clamped: no branch merges)
    for (int p = p0 + pl; p < p1; p += U * lanes) {
      Chunk<T> a[U], b[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const size_t off = ((size_t)img * hw + min(p + k * lanes, p1 - 1)) * c + cc * E;
        a[k].load(g + off);
        b[k].load(feat + off);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float live = p + k * lanes < p1 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += live * a[k].v[e] * b[k].v[e];
      }
    }
  }
  // fold the pixel lanes of the workgroup in LDS, then ONE atomic per channel and workgroup (the 8 x 256 destinations are
  // shared by every workgroup: fewer, block-reduced adds keep the contended-atomic cost negligible at 4x the workgroups)
  __shared__ float red[256 * 8];
#pragma unroll
  for (int e = 0; e < E; ++e) red[(pl * cch + cc) * E + e] = (pl < lanes) ? s[e] : 0.f;
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * c + ch];
    if (p0 < p1) atomicAdd(dq + (size_t)img * c + ch, t);
  }
}

// ROIAlign backward (csrc/cuda/ROIAlign_cuda.cu:125-254) on NHWC, fp32 gradient of the (tiny) query feature map.  Round 5: one
// WAVEFRONT per output cell (roi, ph, pw) like the forward (elementwise.hip): the sample grid and the four tap pixels / weights of
// every sample are wave-uniform, lane l adds the gradient of channels l, l + 64, ... — every atomic wave-instruction is one
// contiguous 256-byte run of a tap pixel's channels (the shape the memory-side atomic unit runs at full rate on; the
// one-thread-per-output mapping of the reference gives the same addresses but recomputes the geometry per channel).
__global__ void __launch_bounds__(64) roialign_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ rois, float* __restrict__ gx, int h,
                                                          int w, int c, int num_rois, float scale, int ph, int pw, int sampling) {
  const int cell = blockIdx.x;
  const int px = cell % pw, py = (cell / pw) % ph, r = cell / (pw * ph);
  const int lane = threadIdx.x;
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
  const float roi_w = fmaxf(rew - rsw, 1.f), roi_h = fmaxf(reh - rsh, 1.f);
  const float bin_h = roi_h / (float)ph, bin_w = roi_w / (float)pw;
  const int gh = sampling > 0 ? sampling : (int)ceilf(roi_h / ph);
  const int gw = sampling > 0 ? sampling : (int)ceilf(roi_w / pw);
  const float inv = (float)(gh * gw);
  float* img = gx + (size_t)b * h * w * c;
  const float* grow = gy + (size_t)cell * c;
  for (int iy = 0; iy < gh; ++iy) {
    const float yy = rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh;
    for (int ix = 0; ix < gw; ++ix) {
      const float xx = rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw;
      float yv = yy, xv = xx;
      if (yv < -1.0f || yv > (float)h || xv < -1.0f || xv > (float)w) continue;      // wave-uniform
      if (yv <= 0.f) yv = 0.f;
      if (xv <= 0.f) xv = 0.f;
      int yl = (int)yv, xl = (int)xv, yh, xh;
      if (yl >= h - 1) { yh = yl = h - 1; yv = (float)yl; } else { yh = yl + 1; }
      if (xl >= w - 1) { xh = xl = w - 1; xv = (float)xl; } else { xh = xl + 1; }
      const float ly = yv - yl, lx = xv - xl, hy = 1.f - ly, hx = 1.f - lx;
      float* t1 = img + ((size_t)yl * w + xl) * c, *t2 = img + ((size_t)yl * w + xh) * c;
      float* t3 = img + ((size_t)yh * w + xl) * c, *t4 = img + ((size_t)yh * w + xh) * c;
      for (int ch = lane; ch < c; ch += 64) {
        const float g = grow[ch] / inv;
        atomicAdd(t1 + ch, g * hy * hx);
        atomicAdd(t2 + ch, g * hy * lx);
        atomicAdd(t3 + ch, g * ly * hx);
        atomicAdd(t4 + ch, g * ly * lx);
      }
    }
  }
}

// ---- backward of the all-levels query pooling (elementwise.hip: query_pool_levels_kernel; round 6): per ROI the gradient of its target
// image's pooled vector / shots (shot_mean_bwd_kernel's value) scattered by roialign_bwd_kernel's rule into a zeroed fp32 map, all FPN
// levels in one launch; then one cast launch writes the maps in the engine's dtype.  Same expressions as the 4 x levels launches it
// replaces (every ROI adds into its own image, sample after sample).
constexpr int kQPoolLevels = 8;
struct QPoolBwdLevels {
  const float* dq[kQPoolLevels];      // [batch][c]
  float* gx[kQPoolLevels];            // [batch * shots][h][w][c] fp32, zeroed
  void* out[kQPoolLevels];            // the same maps in the engine's dtype
  long long begin[kQPoolLevels + 1];  // first element of each level in the concatenation of the maps (cast kernel)
  int h[kQPoolLevels], w[kQPoolLevels];
  float scale[kQPoolLevels];
  int n_levels;
};

__global__ void __launch_bounds__(64) query_pool_levels_bwd_kernel(QPoolBwdLevels L, const float* __restrict__ rois, int c, int shots, int sampling) {
  const int lvl = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
  const int h = L.h[lvl], w = L.w[lvl];
  const float scale = L.scale[lvl];
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
  const float roi_w = fmaxf(rew - rsw, 1.f), roi_h = fmaxf(reh - rsh, 1.f);
  const float bin_h = roi_h / 1.f, bin_w = roi_w / 1.f;
  const int gh = sampling > 0 ? sampling : (int)ceilf(roi_h / 1);
  const int gw = sampling > 0 ? sampling : (int)ceilf(roi_w / 1);
  const float inv = (float)(gh * gw);
  float* img = L.gx[lvl] + (size_t)b * h * w * c;
  const float* grow = L.dq[lvl] + (size_t)(r / shots) * c;
  for (int iy = 0; iy < gh; ++iy) {
    const float yy = rsh + 0 * bin_h + (iy + .5f) * bin_h / (float)gh;
    for (int ix = 0; ix < gw; ++ix) {
      const float xx = rsw + 0 * bin_w + (ix + .5f) * bin_w / (float)gw;
      float yv = yy, xv = xx;
      if (yv < -1.0f || yv > (float)h || xv < -1.0f || xv > (float)w) continue;      // wave-uniform
      if (yv <= 0.f) yv = 0.f;
      if (xv <= 0.f) xv = 0.f;
      int yl = (int)yv, xl = (int)xv, yh, xh;
      if (yl >= h - 1) { yh = yl = h - 1; yv = (float)yl; } else { yh = yl + 1; }
      if (xl >= w - 1) { xh = xl = w - 1; xv = (float)xl; } else { xh = xl + 1; }
      const float ly = yv - yl, lx = xv - xl, hy = 1.f - ly, hx = 1.f - lx;
      float* t1 = img + ((size_t)yl * w + xl) * c, *t2 = img + ((size_t)yl * w + xh) * c;
      float* t3 = img + ((size_t)yh * w + xl) * c, *t4 = img + ((size_t)yh * w + xh) * c;
      for (int ch = lane; ch < c; ch += 64) {
        const float g = (grow[ch] / (float)shots) / inv;
        atomicAdd(t1 + ch, g * hy * hx);
        atomicAdd(t2 + ch, g * hy * lx);
        atomicAdd(t3 + ch, g * ly * hx);
        atomicAdd(t4 + ch, g * ly * lx);
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) query_pool_levels_cast_kernel(QPoolBwdLevels L) {
  const long long total = L.begin[L.n_levels];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < L.n_levels && i >= L.begin[l + 1]) ++l;
    const long long j = i - L.begin[l];
    reinterpret_cast<T*>(L.out[l])[j] = from_f32<T>(L.gx[l][j]);
  }
}

__global__ void shot_mean_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int b, int shots, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * shots * c) return;
  const int ch = i % c, q = i / c;
  gx[i] = gy[(size_t)(q / shots) * c + ch] / (float)shots;
}

template <typename T> __global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(src[i]);
}

// ---- GroupNorm + ReLU backward.  Forward: z = u*a[n,c] + b[n,c], t = relu(z), a = gamma*rstd, b = beta - mean*a.
// With dz = dt*(z>0), xhat = (u-mean)*rstd and m = hw*cpg:  du = rstd*(dz*gamma - S1/m - xhat*S2/m),
// S1 = sum_g dz*gamma, S2 = sum_g dz*gamma*xhat; dgamma_c = sum dz*xhat, dbeta_c = sum dz.
// xhat comes from the saved per-(image, channel) normalisation xa = rstd, xb = -mean * rstd (never (z - beta) / gamma:
// gamma == 0 is a legal parameter value).  The kernels (all FPN levels of a tower layer per launch) follow further down.
constexpr int kGnSplits = kGnSlabs;      // conv_params.h: the conv epilogue that gathers the backward statistics uses the same slabs

// naive data gradient for strided convs (used for the two 3x3/2 convs P6, P7: M <= 1664 pixels):
// dx[n,hi,wi,ci] = sum_{r,s,co : (hi+pad-r) % stride == 0 ...} dy[n,ho,wo,co] * w[co][r][s][ci]  (forward-packed weights)
template <typename T>
__global__ void dgrad_naive_kernel(const T* __restrict__ dy, const T* __restrict__ wp, const T* __restrict__ mask,
                                   const T* __restrict__ addend, T* __restrict__ dx, int n, int h, int w, int cin, int ho,
                                   int wo, int cout, int R, int S, int stride, int pad, int ktot, int dy_stride) {
  const long long total = (long long)n * h * w * cin;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    long long t = i / cin;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    float acc = 0.f;
    for (int r = 0; r < R; ++r) {
      const int yy = y + pad - r;
      if (yy < 0 || yy % stride) continue;
      const int oy = yy / stride;
      if (oy >= ho) continue;
      for (int s = 0; s < S; ++s) {
        const int xx = x + pad - s;
        if (xx < 0 || xx % stride) continue;
        const int ox = xx / stride;
        if (ox >= wo) continue;
        const T* g = dy + (((size_t)b * ho + oy) * wo + ox) * dy_stride;
        const T* wr = wp + (size_t)(r * S + s) * cin + ci;
        for (int co = 0; co < cout; ++co) acc += to_f32(g[co]) * to_f32(wr[(size_t)co * ktot]);
      }
    }
    if (mask) acc = to_f32(mask[i]) > 0.f ? acc : 0.f;      // ReLU backward of the conv's own input ...
    if (addend) acc += to_f32(addend[i]);                   // ... then the other gradient branch of that tensor
    dx[i] = from_f32<T>(acc);
  }
}

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)
#define OSD_DISPATCH_DTYPE(dtype, CALL_F32, CALL_BF16)                            \
  do {                                                                           \
    if ((dtype) == OSD_F32) { CALL_F32; }                                        \
    else if ((dtype) == OSD_BF16) { CALL_BF16; }                                 \
    else return osd_fail(OSD_ERR_INVALID_ARG, "bad dtype %d", (int)(dtype));     \
  } while (0)

extern "C" int osd_pack_conv_weight_dgrad(const float* w, const float* scale, void* dst, int cout, int cin, int r, int s,
                                          int rows, int cout_pad, int src_orsi, int dtype, void* stream) {
  if (!w || !dst || rows < cin || cout_pad < cout) return osd_fail(OSD_ERR_INVALID_ARG, "pack_dgrad: bad args");
  const int g = grid_for((long long)rows * r * s * cout_pad, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(pack_dgrad_weight_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), w, scale, (float*)dst, cout, cin, r, s, rows, cout_pad, src_orsi),
      hipLaunchKernelGGL(pack_dgrad_weight_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), w, scale, (__bf16*)dst, cout, cin, r, s, rows, cout_pad, src_orsi));
  return osd_check_launch("pack_dgrad_weight");
}

extern "C" int osd_pack_multi(const void* table, const int32_t* block_entry, int n_blocks, const float* src,
                              const float* scales, void* dst, int dgrad, int dtype, void* stream) {
  if (!table || !block_entry || !src || !dst) return osd_fail(OSD_ERR_INVALID_ARG, "pack_multi: null argument");
  if (n_blocks <= 0) return OSD_OK;
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(n_blocks), dim3(256), 0, OSD_STREAM(stream), (const PackEntry*)table, block_entry, src, scales, (float*)dst, dgrad),
      hipLaunchKernelGGL(pack_multi_kernel<__bf16>, dim3(n_blocks), dim3(256), 0, OSD_STREAM(stream), (const PackEntry*)table, block_entry, src, scales, (__bf16*)dst, dgrad));
  return osd_check_launch("pack_multi");
}

extern "C" int osd_sgd_momentum_multi(const void* table, const int32_t* block_entry, int n_blocks, float* params,
                                      const float* grads, float* momentum_buf, float lr, float momentum, int first_step,
                                      void* stream) {
  if (!table || !block_entry || !params || !grads || !momentum_buf) return osd_fail(OSD_ERR_INVALID_ARG, "sgd: null argument");
  if (n_blocks <= 0) return OSD_OK;
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_blocks), dim3(256), 0, OSD_STREAM(stream), (const SgdEntry*)table, block_entry,
                     params, grads, momentum_buf, lr, momentum, first_step);
  return osd_check_launch("sgd_multi");
}

extern "C" int osd_sgd_momentum_pack_multi(const void* table, const int32_t* block_entry, int n_blocks, float* params,
                                           float* grads, float* momentum_buf, const float* scales, void* packed, int dtype,
                                           float lr, float momentum, int first_step, int zero_grads, void* stream) {
  if (!table || !block_entry || !params || !grads || !momentum_buf || !packed) return osd_fail(OSD_ERR_INVALID_ARG, "sgd_pack: null argument");
  if (n_blocks <= 0) return OSD_OK;
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(sgd_pack_multi_kernel<float>, dim3(n_blocks), dim3(256), 0, OSD_STREAM(stream), (const SgdPackEntry*)table, block_entry,
                         params, grads, momentum_buf, scales, (float*)packed, lr, momentum, first_step, zero_grads),
      hipLaunchKernelGGL(sgd_pack_multi_kernel<__bf16>, dim3(n_blocks), dim3(256), 0, OSD_STREAM(stream), (const SgdPackEntry*)table, block_entry,
                         params, grads, momentum_buf, scales, (__bf16*)packed, lr, momentum, first_step, zero_grads));
  return osd_check_launch("sgd_pack_multi");
}

extern "C" int osd_unpack_wgrad(const float* dw_packed, const float* scale, float* grad_oihw, int cout, int cin, int r, int s,
                                int accumulate, void* stream) {
  if (!dw_packed || !grad_oihw) return osd_fail(OSD_ERR_INVALID_ARG, "unpack_wgrad: null argument");
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for((long long)cout * cin * r * s, 256)), dim3(256), 0, OSD_STREAM(stream),
                     dw_packed, scale, grad_oihw, cout, cin, r, s, accumulate);
  return osd_check_launch("unpack_wgrad");
}

extern "C" int osd_scatter2x(const void* src, const void* mask, const void* addend, void* dst, int n, int h, int w, int ho,
                             int wo, int c, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!src || !dst || c % e) return osd_fail(OSD_ERR_INVALID_ARG, "scatter2x: bad args");
  const int g = grid_for((long long)n * h * w * (c / e), 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(scatter2x_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)src, (const float*)mask, (const float*)addend, (float*)dst, n, h, w, ho, wo, c),
      hipLaunchKernelGGL(scatter2x_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)src, (const __bf16*)mask, (const __bf16*)addend, (__bf16*)dst, n, h, w, ho, wo, c));
  return osd_check_launch("scatter2x");
}

extern "C" int osd_add_mask(const void* a, const void* b, const void* mask, void* out, int64_t numel, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!a || !out || numel % e) return osd_fail(OSD_ERR_INVALID_ARG, "add_mask: bad args");
  if (numel == 0) return OSD_OK;
  const int g = grid_for(numel / e, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(add_mask_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)a, (const float*)b, (const float*)mask, (float*)out, (long long)(numel / e)),
      hipLaunchKernelGGL(add_mask_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)a, (const __bf16*)b, (const __bf16*)mask, (__bf16*)out, (long long)(numel / e)));
  return osd_check_launch("add_mask");
}

extern "C" int osd_upsample2x_bwd(const void* inner, const void* prev, void* top, int n, int h, int w, int c, int dtype,
                                  void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!inner || !top || c % e) return osd_fail(OSD_ERR_INVALID_ARG, "upsample2x_bwd: bad args");
  const int g = grid_for((long long)n * h * w * (c / e), 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)inner, (const float*)prev, (float*)top, n, h, w, c),
      hipLaunchKernelGGL(upsample2x_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)inner, (const __bf16*)prev, (__bf16*)top, n, h, w, c));
  return osd_check_launch("upsample2x_bwd");
}

extern "C" int osd_correlate_bwd_query(const void* g, const void* feat, float* dq, int n, int hw, int c, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!g || !feat || !dq || c % e || 256 % (c / e)) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_bwd_query: bad args");
  hipError_t er = hipMemsetAsync(dq, 0, sizeof(float) * (size_t)n * c, OSD_STREAM(stream));
  if (er != hipSuccess) return osd_fail(OSD_ERR_LAUNCH, "correlate_bwd_query: memset failed");
  if (n == 0 || hw == 0) return OSD_OK;
  int slabs = (hw + 63) / 64;        // 64 pixels per workgroup: enough workgroups to keep 8 loads x 256 threads in flight per CU
  if (slabs > 512) slabs = 512;
  dim3 grid(slabs, n);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(correlate_bwd_q_kernel<float>, grid, dim3(256), 0, OSD_STREAM(stream), (const float*)g, (const float*)feat, dq, hw, c),
      hipLaunchKernelGGL(correlate_bwd_q_kernel<__bf16>, grid, dim3(256), 0, OSD_STREAM(stream), (const __bf16*)g, (const __bf16*)feat, dq, hw, c));
  return osd_check_launch("correlate_bwd_query");
}

// every FPN level in one launch: the same kernel per (level, image, slab) through a table of static-index selects
constexpr int kCorrQLevels = 6;
struct CorrQLevels {
  const void* g[kCorrQLevels]; const void* feat[kCorrQLevels]; float* dq[kCorrQLevels];
  int hw[kCorrQLevels], slabs[kCorrQLevels], begin[kCorrQLevels];
  int n_levels;
};
namespace {
template <typename T>
__global__ void __launch_bounds__(256) correlate_bwd_q_levels_kernel(CorrQLevels L, int c) {
  constexpr int E = Chunk<T>::N;
  const int b = blockIdx.x;
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < kCorrQLevels; ++i)
    if (i < L.n_levels && b >= L.begin[i]) lvl = i;
  const void* gv = L.g[0]; const void* fv = L.feat[0]; float* dq = L.dq[0];
  int hw = L.hw[0], slabs = L.slabs[0], beg = L.begin[0];
#pragma unroll
  for (int i = 1; i < kCorrQLevels; ++i)
    if (lvl == i) { gv = L.g[i]; fv = L.feat[i]; dq = L.dq[i]; hw = L.hw[i]; slabs = L.slabs[i]; beg = L.begin[i]; }
  const T* g = reinterpret_cast<const T*>(gv);
  const T* feat = reinterpret_cast<const T*>(fv);
  const int local = b - beg;
  const int img = local / slabs, slab = local - img * slabs;
  const int cch = c / E;
  const int lanes = 256 / cch;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int per = (hw + slabs - 1) / slabs;
  const int p0 = slab * per, p1 = min(hw, p0 + per);
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  if (pl < lanes && p0 < p1) {
    constexpr int U = 4;
    for (int p = p0 + pl; p < p1; p += U * lanes) {
      Chunk<T> a[U], bb[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const size_t off = ((size_t)img * hw + min(p + k * lanes, p1 - 1)) * c + cc * E;
        a[k].load(g + off);
        bb[k].load(feat + off);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float live = p + k * lanes < p1 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += live * a[k].v[e] * bb[k].v[e];
      }
    }
  }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int e = 0; e < E; ++e) red[(pl * cch + cc) * E + e] = (pl < lanes) ? s[e] : 0.f;
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * c + ch];
    if (p0 < p1) atomicAdd(dq + (size_t)img * c + ch, t);
  }
}
}  // namespace

extern "C" int osd_correlate_bwd_query_levels(int n_levels, const void* const* gs, const void* const* feats,
                                              float* const* dqs, const int32_t* hws, int n, int c, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (n_levels < 1 || n_levels > kCorrQLevels || !gs || !feats || !dqs || !hws || c % e || 256 % (c / e) || c > 2048)
    return osd_fail(OSD_ERR_INVALID_ARG, "correlate_bwd_query_levels: bad args");
  if (dtype != OSD_F32 && dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_bwd_query_levels: bad dtype");
  if (n == 0) return OSD_OK;
  CorrQLevels L;
  L.n_levels = 0;
  int blocks = 0;
  for (int i = 0; i < n_levels; ++i)
    if (!dqs[i]) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_bwd_query_levels: null dq at level %d", i);
  // the sums are accumulated atomically: clear them first — one fill per run of adjacent level buffers (the host side
  // hands over the rows of ONE [levels, n, c] tensor: a single fill launch instead of five on the step's main chain)
  for (int i = 0; i < n_levels;) {
    int j = i + 1;
    while (j < n_levels && dqs[j] == dqs[j - 1] + (size_t)n * c) ++j;
    hipError_t er = hipMemsetAsync(dqs[i], 0, sizeof(float) * (size_t)n * c * (j - i), OSD_STREAM(stream));
    if (er != hipSuccess) return osd_fail(OSD_ERR_LAUNCH, "correlate_bwd_query_levels: memset failed");
    i = j;
  }
  for (int i = 0; i < n_levels; ++i) {
    if (hws[i] <= 0) continue;
    if (!gs[i] || !feats[i]) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_bwd_query_levels: null tensor at level %d", i);
    int slabs = (hws[i] + 63) / 64;
    if (slabs > 512) slabs = 512;
    const int k = L.n_levels++;
    L.g[k] = gs[i]; L.feat[k] = feats[i]; L.dq[k] = dqs[i]; L.hw[k] = hws[i]; L.slabs[k] = slabs; L.begin[k] = blocks;
    blocks += slabs * n;
  }
  if (L.n_levels == 0) return OSD_OK;
  for (int k = L.n_levels; k < kCorrQLevels; ++k) { L.g[k] = L.g[0]; L.feat[k] = L.feat[0]; L.dq[k] = L.dq[0]; L.hw[k] = 0; L.slabs[k] = 1; L.begin[k] = 0x7fffffff; }
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(correlate_bwd_q_levels_kernel<float>, dim3(blocks), dim3(256), 0, OSD_STREAM(stream), L, c),
      hipLaunchKernelGGL(correlate_bwd_q_levels_kernel<__bf16>, dim3(blocks), dim3(256), 0, OSD_STREAM(stream), L, c));
  return osd_check_launch("correlate_bwd_query_levels");
}

extern "C" int osd_roialign_bwd(const float* gy, const float* rois, float* gx, int b, int h, int w, int c, int num_rois,
                                float spatial_scale, int ph, int pw, int sampling_ratio, void* stream) {
  if (!gy || !rois || !gx) return osd_fail(OSD_ERR_INVALID_ARG, "roialign_bwd: null argument");
  hipError_t er = hipMemsetAsync(gx, 0, sizeof(float) * (size_t)b * h * w * c, OSD_STREAM(stream));
  if (er != hipSuccess) return osd_fail(OSD_ERR_LAUNCH, "roialign_bwd: memset failed");
  if (num_rois == 0) return OSD_OK;
  const long long cells = (long long)num_rois * ph * pw;      // one wavefront per output cell
  if (cells > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "roialign_bwd: too many cells");
  hipLaunchKernelGGL(roialign_bwd_kernel, dim3((unsigned)cells), dim3(64), 0, OSD_STREAM(stream),
                     gy, rois, gx, h, w, c, num_rois, spatial_scale, ph, pw, sampling_ratio);
  return osd_check_launch("roialign_bwd");
}

extern "C" int osd_query_pool_levels_bwd(int n_levels, const float* const* dqs, const int32_t* hs, const int32_t* ws, const float* scales,
                                         const float* rois, int batch, int shots, int c, int sampling_ratio, float* gx32,
                                         void* const* outs, int dtype, void* stream) {
  if (n_levels < 1 || n_levels > kQPoolLevels || !dqs || !hs || !ws || !scales || !rois || !gx32 || !outs || shots < 1 || c < 1)
    return osd_fail(OSD_ERR_INVALID_ARG, "query_pool_levels_bwd: bad arguments");
  if (batch == 0) return OSD_OK;
  QPoolBwdLevels L;
  L.n_levels = n_levels;
  long long off = 0;
  for (int l = 0; l < kQPoolLevels; ++l) {
    const int j = l < n_levels ? l : 0;
    if (!dqs[j] || !outs[j] || hs[j] < 1 || ws[j] < 1) return osd_fail(OSD_ERR_INVALID_ARG, "query_pool_levels_bwd: bad level %d", j);
    L.dq[l] = dqs[j]; L.out[l] = outs[j]; L.h[l] = hs[j]; L.w[l] = ws[j]; L.scale[l] = scales[j];
    L.begin[l] = off;
    L.gx[l] = gx32 + (l < n_levels ? off : 0);
    if (l < n_levels) off += (long long)batch * shots * hs[j] * ws[j] * c;
  }
  for (int l = n_levels; l <= kQPoolLevels; ++l) L.begin[l] = off;
  hipError_t er = hipMemsetAsync(gx32, 0, sizeof(float) * (size_t)off, OSD_STREAM(stream));
  if (er != hipSuccess) return osd_fail(OSD_ERR_LAUNCH, "query_pool_levels_bwd: memset failed");
  hipLaunchKernelGGL(query_pool_levels_bwd_kernel, dim3(batch * shots, n_levels), dim3(64), 0, OSD_STREAM(stream), L, rois, c, shots, sampling_ratio);
  int rc = osd_check_launch("query_pool_levels_bwd");
  if (rc) return rc;
  const int g = grid_for(off, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(query_pool_levels_cast_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), L),
      hipLaunchKernelGGL(query_pool_levels_cast_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), L));
  return osd_check_launch("query_pool_levels_cast");
}

extern "C" int osd_shot_mean_bwd(const float* gy, float* gx, int b, int shots, int c, void* stream) {
  if (!gy || !gx || shots < 1) return osd_fail(OSD_ERR_INVALID_ARG, "shot_mean_bwd: bad args");
  hipLaunchKernelGGL(shot_mean_bwd_kernel, dim3(cdiv(b * shots * c, 256)), dim3(256), 0, OSD_STREAM(stream), gy, gx, b, shots, c);
  return osd_check_launch("shot_mean_bwd");
}

extern "C" int osd_cast_f32(const float* src, void* dst, int64_t numel, int dtype, void* stream) {
  if (!src || !dst) return osd_fail(OSD_ERR_INVALID_ARG, "cast_f32: null argument");
  if (numel == 0) return OSD_OK;
  const int g = grid_for(numel, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), src, (float*)dst, (long long)numel),
      hipLaunchKernelGGL(cast_from_f32_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), src, (__bf16*)dst, (long long)numel));
  return osd_check_launch("cast_f32");
}

// fp32 <-> bf16 in 16-byte-per-lane pieces (the gradient buckets on their way to and from a bf16 all-reduce); numel % 8 == 0
__global__ void __launch_bounds__(256) f32_to_bf16_vec_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    bf16x8 o;
    o[0] = (__bf16)a[0]; o[1] = (__bf16)a[1]; o[2] = (__bf16)a[2]; o[3] = (__bf16)a[3];
    o[4] = (__bf16)b[0]; o[5] = (__bf16)b[1]; o[6] = (__bf16)b[2]; o[7] = (__bf16)b[3];
    *reinterpret_cast<bf16x8*>(dst + i * 8) = o;
  }
}
__global__ void __launch_bounds__(256) bf16_to_f32_vec_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long long n8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i * 8);
    *reinterpret_cast<f32x4*>(dst + i * 8) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  }
}

extern "C" int osd_grad_wire_cast(const void* src, void* dst, int64_t numel, int to_wire, void* stream) {
  if (!src || !dst) return osd_fail(OSD_ERR_INVALID_ARG, "grad_wire_cast: null argument");
  if (numel % 8) return osd_fail(OSD_ERR_INVALID_ARG, "grad_wire_cast: numel must be a multiple of 8");
  if (numel == 0) return OSD_OK;
  const long long n8 = numel / 8;
  const int g = (int)(n8 + 255) / 256 > 2048 ? 2048 : (int)((n8 + 255) / 256);
  if (to_wire)
    hipLaunchKernelGGL(f32_to_bf16_vec_kernel, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)src, (__bf16*)dst, n8);
  else
    hipLaunchKernelGGL(bf16_to_f32_vec_kernel, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)src, (float*)dst, n8);
  return osd_check_launch("grad_wire_cast");
}

extern "C" int osd_conv2d_dgrad_naive(const osd_conv_desc* d, const void* dy, const void* w_fwd_packed, const void* mask,
                                      const void* addend, void* dx, void* stream) {
  if (!d || !dy || !w_fwd_packed || !dx) return osd_fail(OSD_ERR_INVALID_ARG, "dgrad_naive: null argument");
  const int g = grid_for((long long)d->n * d->h * d->w * d->cin, 256);
  const int ktot = d->r * d->s * d->cin;
  OSD_DISPATCH_DTYPE(d->dtype,
      hipLaunchKernelGGL(dgrad_naive_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)dy, (const float*)w_fwd_packed, (const float*)mask, (const float*)addend, (float*)dx, d->n, d->h, d->w, d->cin, d->ho, d->wo, d->cout, d->r, d->s, d->stride_h, d->pad_h, ktot, d->out_stride),
      hipLaunchKernelGGL(dgrad_naive_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)dy, (const __bf16*)w_fwd_packed, (const __bf16*)mask, (const __bf16*)addend, (__bf16*)dx, d->n, d->h, d->w, d->cin, d->ho, d->wo, d->cout, d->r, d->s, d->stride_h, d->pad_h, ktot, d->out_stride));
  return osd_check_launch("dgrad_naive");
}

// ====================================================================================================================
// GroupNorm + ReLU over ALL FPN levels of one tower layer in two launches (forward) / two launches (backward).
// The five levels share gamma/beta but are separate tensors; blockIdx.z = level, blockIdx.y = image, blockIdx.x = slab.
// The per-(image, channel) scale/shift (forward) and the per-(image, group) sums (backward) are finalised INSIDE the
// apply kernels from the slab partials (each workgroup redoes the tiny reduction for its image in LDS), so there is no
// finalize / reduce launch.
// ====================================================================================================================
namespace {

constexpr int kGnL = 8;
struct GnLevels {
  const void* x[kGnL];      // forward: conv output u; backward: u
  const void* dy[kGnL];     // backward: dt
  void* y[kGnL];            // forward: t = relu(gn(u)); backward: du
  int hw[kGnL];
  int n_levels;
};

template <typename T>
__global__ void __launch_bounds__(256) gnl_stats_kernel(GnLevels L, float* __restrict__ ws, int n, int c, int groups, unsigned fused_mask) {
  constexpr int E = Chunk<T>::N;
  const int lvl = blockIdx.z, img = blockIdx.y, split = blockIdx.x;
  if ((fused_mask >> lvl) & 1u) return;      // this level's sums were gathered by the conv that wrote x (conv_params.h: ConvGnb)
  const int hw = L.hw[lvl];
  const T* x = reinterpret_cast<const T*>(L.x[lvl]);
  const int cch = c / E, lanes = 256 / cch;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int per = (hw + kGnSplits - 1) / kGnSplits;
  const int p0 = split * per, p1 = min(hw, p0 + per);
  float s = 0.f, ss = 0.f;
  if (pl < lanes) {
    constexpr int U = 4;        // pixels in flight per thread (one 16-byte load each): a P3 slab is 200 pixels on 2 workgroups per CU
    for (int p = p0 + pl; p < p1; p += U * lanes) {
      Chunk<T> v[U];
#pragma unroll
      for (int k = 0; k < U; ++k)       // unconditional (clamped) loads, masked in the arithmetic
        v[k].load(x + ((size_t)img * hw + min(p + k * lanes, p1 - 1)) * c + cc * E);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float m = (p + k * lanes < p1) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) { const float t = v[k].v[e] * m; s += t; ss += t * t; }
      }
    }
  }
  __shared__ float red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  const int cpg_chunks = (c / groups) / E > 0 ? (c / groups) / E : 1;
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    float ts = 0.f, tss = 0.f;
    for (int l = 0; l < lanes; ++l)
      for (int k = 0; k < cpg_chunks; ++k) {
        ts += red[0][l * cch + g * cpg_chunks + k];
        tss += red[1][l * cch + g * cpg_chunks + k];
      }
    float* o = ws + ((((size_t)lvl * n + img) * kGnSplits + split) * groups + g) * 2;
    o[0] = ts;
    o[1] = tss;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gnl_apply_kernel(GnLevels L, const float* __restrict__ ws, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ ab, int n, int c,
                                                        int groups, float eps) {
  constexpr int E = Chunk<T>::N;
  const int lvl = blockIdx.z, img = blockIdx.y;
  const int hw = L.hw[lvl];
  __shared__ float sm[2][64];        // mean, rstd per group
  __shared__ float sa[512], sb[512];
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    double s = 0.0, ss = 0.0;
    const float* o = ws + (((size_t)lvl * n + img) * kGnSplits) * groups * 2;
    for (int k = 0; k < kGnSplits; ++k) {
      s += o[((size_t)k * groups + g) * 2];
      ss += o[((size_t)k * groups + g) * 2 + 1];
    }
    const double cnt = (double)hw * (c / groups);
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    sm[0][g] = (float)mean;
    sm[1][g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    const int g = ch / (c / groups);
    const float av = gamma[ch] * sm[1][g];
    const float bv = beta[ch] - sm[0][g] * av;
    sa[ch] = av;
    sb[ch] = bv;
    if (blockIdx.x == 0) {          // saved for the backward pass: ab[level][4][n][c] = a, b (y = a u + b) and the
      // normalisation itself, xhat = xa u + xb, so that the backward pass never divides by gamma (gamma == 0 is legal)
      ab[(((size_t)lvl * 4 + 0) * n + img) * c + ch] = av;
      ab[(((size_t)lvl * 4 + 1) * n + img) * c + ch] = bv;
      ab[(((size_t)lvl * 4 + 2) * n + img) * c + ch] = sm[1][g];
      ab[(((size_t)lvl * 4 + 3) * n + img) * c + ch] = -sm[0][g] * sm[1][g];
    }
  }
  __syncthreads();
  const T* x = reinterpret_cast<const T*>(L.x[lvl]);
  T* y = reinterpret_cast<T*>(L.y[lvl]);
  const int cch = c / E;
  const long long chunks = (long long)hw * cch;
  const long long per = (chunks + gridDim.x - 1) / gridDim.x;
  const long long i0 = blockIdx.x * per, i1 = min(chunks, i0 + per);
  // blockDim.x is a multiple of cch (checked by the launcher), so a thread's channel chunk never changes
  const int cc = (int)((i0 + threadIdx.x) % cch);
  float av[E], bv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { av[e] = sa[cc * E + e]; bv[e] = sb[cc * E + e]; }
  constexpr int U = 4;
  const size_t base = (size_t)img * hw * c;
  for (long long i = i0 + threadIdx.x; i < i1; i += (long long)U * blockDim.x) {
    Chunk<T> v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {        // unconditional loads; an out-of-range slot re-reads this thread's first chunk
      const long long ik = i + (long long)k * blockDim.x;
      v[k].load(x + base + (ik < i1 ? ik : i) * E);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const long long ik = i + (long long)k * blockDim.x;
      if (ik >= i1) break;
#pragma unroll
      for (int e = 0; e < E; ++e) v[k].v[e] = fmaxf(fmaf(v[k].v[e], av[e], bv[e]), 0.f);
      v[k].store(y + base + ik * E);
    }
  }
}

template <typename T, bool SX>
__global__ void __launch_bounds__(256) gnl_bwd_stats_kernel(GnLevels L, const float* __restrict__ ab, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ ws,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int n, int c,
                                                            int groups, unsigned fused_mask, float* __restrict__ sxw) {
  // sxw (optional): per-(level, image, slab) partial sums of xhat per channel, [levels][n][kGnSplits][c] — with them the apply
  // kernel knows sum_px du per channel without a pass over du: the bias gradient of the conv that produced u (see there)
  constexpr int E = Chunk<T>::N;
  const int lvl = blockIdx.z, img = blockIdx.y, split = blockIdx.x;
  if ((fused_mask >> lvl) & 1u) return;      // this level's sums were gathered by the conv that wrote dt (conv_params.h: ConvGnb)
  const int hw = L.hw[lvl];
  const T* u = reinterpret_cast<const T*>(L.x[lvl]);
  const T* dt = reinterpret_cast<const T*>(L.dy[lvl]);
  const int cch = c / E, lanes = 256 / cch;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int per = (hw + kGnSplits - 1) / kGnSplits;
  const int p0 = split * per, p1 = min(hw, p0 + per);
  float av[E], bv[E], gm[E], xa[E], xb[E], s1 = 0.f, s2 = 0.f, dg[E], db[E], sx[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    av[e] = ab[(((size_t)lvl * 4 + 0) * n + img) * c + cc * E + e];
    bv[e] = ab[(((size_t)lvl * 4 + 1) * n + img) * c + cc * E + e];
    xa[e] = ab[(((size_t)lvl * 4 + 2) * n + img) * c + cc * E + e];
    xb[e] = ab[(((size_t)lvl * 4 + 3) * n + img) * c + cc * E + e];
    gm[e] = gamma[cc * E + e];
    dg[e] = 0.f; db[e] = 0.f; sx[e] = 0.f;
  }
  if (pl < lanes) {
    constexpr int U = 4;          // pixels in flight per thread: 2 * U 16-byte loads issued before the arithmetic
    for (int p = p0 + pl; p < p1; p += U * lanes) {
      Chunk<T> uu[U], gg[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {      // unconditional (clamped) loads: a branch around a load costs a vmcnt(0) at its merge
        const int pk = min(p + k * lanes, p1 - 1);
        const size_t off = ((size_t)img * hw + pk) * c + cc * E;
        uu[k].load(u + off);
        gg[k].load(dt + off);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (p + k * lanes >= p1) break;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float z = fmaf(uu[k].v[e], av[e], bv[e]);
          const float dz = z > 0.f ? gg[k].v[e] : 0.f;
          const float xhat = fmaf(uu[k].v[e], xa[e], xb[e]);
          s1 += dz * gm[e];
          s2 += dz * gm[e] * xhat;
          dg[e] += dz * xhat;
          db[e] += dz;
          if constexpr (SX) sx[e] += xhat;
        }
      }
    }
  }
  __shared__ float red[3][256];
  __shared__ float redc[3][512];
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  const int cpg_chunks = (c / groups) / E > 0 ? (c / groups) / E : 1;
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    float t1 = 0.f, t2 = 0.f;
    for (int l = 0; l < lanes; ++l)
      for (int k = 0; k < cpg_chunks; ++k) {
        t1 += red[0][l * cch + g * cpg_chunks + k];
        t2 += red[1][l * cch + g * cpg_chunks + k];
      }
    float* o = ws + ((((size_t)lvl * n + img) * kGnSplits + split) * groups + g) * 2;
    o[0] = t1;
    o[1] = t2;
  }
  for (int e = 0; e < E; ++e) {
    __syncthreads();
    red[0][threadIdx.x] = dg[e];
    red[1][threadIdx.x] = db[e];
    if constexpr (SX) red[2][threadIdx.x] = sx[e];
    __syncthreads();
    if (threadIdx.x < cch) {
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
      for (int l = 0; l < lanes; ++l) {
        t1 += red[0][l * cch + threadIdx.x];
        t2 += red[1][l * cch + threadIdx.x];
        if constexpr (SX) t3 += red[2][l * cch + threadIdx.x];
      }
      redc[0][threadIdx.x * E + e] = t1;
      redc[1][threadIdx.x * E + e] = t2;
      if constexpr (SX) redc[2][threadIdx.x * E + e] = t3;
    }
  }
  __syncthreads();
  // per-(level, image, slab) partial d gamma / d beta: plain stores (2,560 workgroups adding atomically into the same
  // 2 x c words ran at the contended-atomic rate and cost more than the whole data pass); summed by the apply kernel
  float* pw = ws + (size_t)gridDim.z * n * kGnSplits * groups * 2 +
              ((((size_t)lvl * n + img) * kGnSplits + split) * 2) * c;
  float* px = (SX && sxw) ? sxw + ((((size_t)lvl * n + img) * kGnSplits + split)) * c : nullptr;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    pw[ch] = redc[0][ch];
    pw[c + ch] = redc[1][ch];
    if (px) px[ch] = redc[2][ch];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gnl_bwd_apply_kernel(GnLevels L, const float* __restrict__ ab, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ ws,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int n,
                                                            int c, int groups, const float* __restrict__ sxw,
                                                            float* __restrict__ conv_db) {
  constexpr int E = Chunk<T>::N;
  const int lvl = blockIdx.z, img = blockIdx.y;
  const int hw = L.hw[lvl];
  __shared__ float ssum[2][64];
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    const float* o = ws + (((size_t)lvl * n + img) * kGnSplits) * groups * 2;
    for (int k = 0; k < kGnSplits; ++k) {
      s1 += o[((size_t)k * groups + g) * 2];
      s2 += o[((size_t)k * groups + g) * 2 + 1];
    }
    ssum[0][g] = s1;
    ssum[1][g] = s2;
  }
  // one workgroup per (level, image) folds the slabs' d gamma / d beta partials (while the first threads sum the group partials).
  // The bias gradient of the conv that produced u (fcos.py:29-37: Conv2d(bias=True) -> GroupNorm) is sum_px du, and
  //   sum_px du = rstd (gamma sum_px dz - N c1 - c2 sum_px xhat)
  // follows from sums this pass has (sum dz = d beta; c1, c2 = the group means) plus sum_px xhat per channel from the statistics
  // pass: one atomic per channel and (level, image) instead of a d-bias column sum inside the tower's weight-gradient launch, which
  // cost it 11 % (610 vs 550 us, tools/sk_bias_cost.py).  It is the sum of the fp32 du, not of their bf16 roundings
  // The first four workgroups of a (level, image) share the fold in 64-channel chunks: thread (part = t / 64, channel t % 64) adds every
  // fourth slab (a wave reads 64 consecutive channels: coalesced), LDS adds the four parts.  One workgroup folding all 256 channels x
  // 64 slabs x 3 planes (192 loads per thread) held the launch ~5 us longer.
  constexpr int kFoldWgs = 4, kFoldW = 64;              // 64-channel chunks, dealt round-robin to the first four workgroups
  __shared__ float fold[3][256];
  __syncthreads();                                      // ssum
  if ((int)blockIdx.x < kFoldWgs) {
    const int nfold = min((int)gridDim.x, kFoldWgs);
    const float* pw = ws + (size_t)gridDim.z * n * kGnSplits * groups * 2 + (((size_t)lvl * n + img) * kGnSplits * 2) * c;
    const float* px = sxw ? sxw + (((size_t)lvl * n + img) * kGnSplits) * c : nullptr;
    const int cpg0 = c / groups;
    const float inv_m0 = 1.f / ((float)hw * cpg0);
    for (int chunk = blockIdx.x; chunk * kFoldW < c; chunk += nfold) {
      const int ch = chunk * kFoldW + (threadIdx.x % kFoldW), part = threadIdx.x / kFoldW;      // 4 parts x 16 slabs
      float t1 = 0.f, t2 = 0.f, t3 = 0.f;
      if (ch < c) {
#pragma unroll 8
        for (int k = part; k < kGnSplits; k += 256 / kFoldW) {
          t1 += pw[(size_t)k * 2 * c + ch];
          t2 += pw[(size_t)k * 2 * c + c + ch];
        }
        if (px != nullptr) {
#pragma unroll 8
          for (int k = part; k < kGnSplits; k += 256 / kFoldW) t3 += px[(size_t)k * c + ch];
        }
      }
      __syncthreads();                                  // the previous chunk's sums have been read
      fold[0][threadIdx.x] = t1; fold[1][threadIdx.x] = t2; fold[2][threadIdx.x] = t3;
      __syncthreads();
      if (threadIdx.x < kFoldW && ch < c) {
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int q = 0; q < 256 / kFoldW; ++q) { a1 += fold[0][q * kFoldW + threadIdx.x]; a2 += fold[1][q * kFoldW + threadIdx.x]; a3 += fold[2][q * kFoldW + threadIdx.x]; }
        atomicAdd(dgamma + ch, a1);
        atomicAdd(dbeta + ch, a2);
        if (conv_db != nullptr) {
          const int g0 = ch / cpg0;
          const float rstd = ab[(((size_t)lvl * 4 + 2) * n + img) * c + ch];
          atomicAdd(conv_db + ch, rstd * (gamma[ch] * a2 - (float)hw * (ssum[0][g0] * inv_m0) - (ssum[1][g0] * inv_m0) * a3));
        }
      }
    }
  }
  const T* u = reinterpret_cast<const T*>(L.x[lvl]);
  const T* dt = reinterpret_cast<const T*>(L.dy[lvl]);
  T* du = reinterpret_cast<T*>(L.y[lvl]);
  const int cch = c / E, cpg = c / groups;
  const float inv_m = 1.f / ((float)hw * cpg);
  const long long chunks = (long long)hw * cch;
  const long long per = (chunks + gridDim.x - 1) / gridDim.x;
  const long long i0 = blockIdx.x * per, i1 = min(chunks, i0 + per);
  // blockDim.x is a multiple of cch, so a thread's channel chunk never changes: per-channel constants live in registers
  const int cc = (int)((i0 + threadIdx.x) % cch);
  const int g = (cc * E) / cpg;
  const float c1 = ssum[0][g] * inv_m, c2 = ssum[1][g] * inv_m;
  float av[E], bv[E], gm[E], xa[E], xb[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int ch = cc * E + e;
    av[e] = ab[(((size_t)lvl * 4 + 0) * n + img) * c + ch];
    bv[e] = ab[(((size_t)lvl * 4 + 1) * n + img) * c + ch];
    xa[e] = ab[(((size_t)lvl * 4 + 2) * n + img) * c + ch];      // rstd of the channel's group
    xb[e] = ab[(((size_t)lvl * 4 + 3) * n + img) * c + ch];
    gm[e] = gamma[ch];
  }
  constexpr int U = 4;
  const size_t base = (size_t)img * hw * c;
  for (long long i = i0 + threadIdx.x; i < i1; i += (long long)U * blockDim.x) {
    Chunk<T> uu[U], gg[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {        // unconditional loads; an out-of-range slot re-reads this thread's first chunk
      const long long ik = i + (long long)k * blockDim.x;
      const long long il = ik < i1 ? ik : i;
      uu[k].load(u + base + il * E);
      gg[k].load(dt + base + il * E);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const long long ik = i + (long long)k * blockDim.x;
      if (ik >= i1) break;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float z = fmaf(uu[k].v[e], av[e], bv[e]);
        const float dz = z > 0.f ? gg[k].v[e] : 0.f;
        const float xhat = fmaf(uu[k].v[e], xa[e], xb[e]);
        uu[k].v[e] = xa[e] * (dz * gm[e] - c1 - xhat * c2);
      }
      uu[k].store(du + base + ik * E);
    }
  }
}

// workgroups per (image, level) of the apply kernels (they split the chunks evenly, no workspace layout depends on it)
int gn_apply_blocks() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("OSD_GN_APPLY_BLOCKS"); v = e ? atoi(e) : 32; if (v < 1) v = 32; }   // 32: measured best of 24..512 (tools/gn_bench.py: every workgroup re-reduces the slab partials)
  return v;
}

int gn_levels_fill(GnLevels& L, int n_levels, const void* const* xs, const void* const* dys, void* const* ys, const int32_t* hws) {
  if (n_levels < 1 || n_levels > kGnL || !xs || !ys || !hws) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_levels: bad arguments");
  L.n_levels = n_levels;
  for (int i = 0; i < kGnL; ++i) {
    const int j = i < n_levels ? i : 0;
    L.x[i] = xs[j]; L.dy[i] = dys ? dys[j] : nullptr; L.y[i] = ys[j]; L.hw[i] = hws[j];
  }
  return OSD_OK;
}

}  // namespace

// ws: n_levels * n * OSD_GN_SPLITS * groups * 2 floats; ab (out): [n_levels][2][n][c] fp32 scale / shift per image, channel
// fused_mask: bit l set = level l's slab sums in ws were accumulated by osd_conv2d_fwd_multi_gn (forward statistics) into ZEROED
// memory; the statistics pass skips those levels (and is not launched when every level is fused)
extern "C" int osd_groupnorm_relu_fwd_levels_fused(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                                   const float* gamma, const float* beta, float* ab, float* ws, int n, int c,
                                                   int groups, float eps, int dtype, uint32_t fused_mask, void* stream);

extern "C" int osd_groupnorm_relu_fwd_levels(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                             const float* gamma, const float* beta, float* ab, float* ws, int n, int c,
                                             int groups, float eps, int dtype, void* stream) {
  return osd_groupnorm_relu_fwd_levels_fused(n_levels, xs, ys, hws, gamma, beta, ab, ws, n, c, groups, eps, dtype, 0u, stream);
}

extern "C" int osd_groupnorm_relu_fwd_levels_fused(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                                   const float* gamma, const float* beta, float* ab, float* ws, int n, int c,
                                                   int groups, float eps, int dtype, uint32_t fused_mask, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!gamma || !beta || !ab || !ws || c % e != 0 || c > 512 || c / e > 256 || 256 % (c / e) != 0 || groups > 64 ||
      c % groups != 0 || (c / groups) % e != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_levels: unsupported shape c=%d groups=%d", c, groups);
  GnLevels L;
  int rc = gn_levels_fill(L, n_levels, xs, nullptr, ys, hws);
  if (rc) return rc;
  dim3 g1(kGnSplits, n, n_levels), g2(gn_apply_blocks(), n, n_levels);
#ifdef OSD_GN_DIAG      // diagnostic build: OSD_GN_SKIP=1 (forward) / 2 (backward) / 3 leaves the statistics launches out (garbage results)
  static int skip = -1;
  if (skip < 0) { const char* e = getenv("OSD_GN_SKIP"); skip = e ? atoi(e) : 0; }
  if (!(skip & 1))
#endif
  if ((fused_mask & ((1u << n_levels) - 1u)) != ((1u << n_levels) - 1u))
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(gnl_stats_kernel<float>, g1, dim3(256), 0, OSD_STREAM(stream), L, ws, n, c, groups, fused_mask),
      hipLaunchKernelGGL(gnl_stats_kernel<__bf16>, g1, dim3(256), 0, OSD_STREAM(stream), L, ws, n, c, groups, fused_mask));
  rc = osd_check_launch("gnl_stats");
  if (rc) return rc;
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(gnl_apply_kernel<float>, g2, dim3(256), 0, OSD_STREAM(stream), L, ws, gamma, beta, ab, n, c, groups, eps),
      hipLaunchKernelGGL(gnl_apply_kernel<__bf16>, g2, dim3(256), 0, OSD_STREAM(stream), L, ws, gamma, beta, ab, n, c, groups, eps));
  return osd_check_launch("gnl_apply");
}

// fused_mask: bit l set = level l's slab sums in ws (both parts) were accumulated by osd_conv2d_fwd_multi_gn into ZEROED memory;
// the statistics pass skips those levels (and is not launched when every level is fused)

static int gn_bwd_levels_impl(int n_levels, const void* const* us, const void* const* dts, void* const* dus, const int32_t* hws,
                              const float* ab, const float* gamma, const float* beta, float* ws, float* dgamma, float* dbeta,
                              float* conv_dbias, int n, int c, int groups, int dtype, uint32_t fused_mask, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!gamma || !beta || !ab || !ws || !dgamma || !dbeta || !dts || c % e != 0 || c > 512 || c / e > 256 ||
      256 % (c / e) != 0 || groups > 64 || c % groups != 0 || (c / groups) % e != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_bwd_levels: unsupported shape c=%d groups=%d", c, groups);
  if (conv_dbias != nullptr && fused_mask != 0u)
    return osd_fail(OSD_ERR_UNSUPPORTED, "groupnorm_bwd_levels: the conv bias gradient needs this launch's own statistics pass on every level");
  GnLevels L;
  int rc = gn_levels_fill(L, n_levels, us, dts, dus, hws);
  if (rc) return rc;
  // the per-slab sums of xhat (for conv_dbias) follow the group sums and the d gamma / d beta partials in ws
  float* sxw = conv_dbias ? ws + (size_t)n_levels * n * kGnSplits * (groups * 2 + 2 * c) : nullptr;
  dim3 g1(kGnSplits, n, n_levels), g2(gn_apply_blocks(), n, n_levels);
#ifdef OSD_GN_DIAG
  static int skip = -1;
  if (skip < 0) { const char* e = getenv("OSD_GN_SKIP"); skip = e ? atoi(e) : 0; }
  if (!(skip & 2))
#endif
  if ((fused_mask & ((1u << n_levels) - 1u)) != ((1u << n_levels) - 1u))
  {
    if (sxw) {
      OSD_DISPATCH_DTYPE(dtype,
          hipLaunchKernelGGL((gnl_bwd_stats_kernel<float, true>), g1, dim3(256), 0, OSD_STREAM(stream), L, ab, gamma, beta, ws, dgamma, dbeta, n, c, groups, fused_mask, sxw),
          hipLaunchKernelGGL((gnl_bwd_stats_kernel<__bf16, true>), g1, dim3(256), 0, OSD_STREAM(stream), L, ab, gamma, beta, ws, dgamma, dbeta, n, c, groups, fused_mask, sxw));
    } else {
      OSD_DISPATCH_DTYPE(dtype,
          hipLaunchKernelGGL((gnl_bwd_stats_kernel<float, false>), g1, dim3(256), 0, OSD_STREAM(stream), L, ab, gamma, beta, ws, dgamma, dbeta, n, c, groups, fused_mask, sxw),
          hipLaunchKernelGGL((gnl_bwd_stats_kernel<__bf16, false>), g1, dim3(256), 0, OSD_STREAM(stream), L, ab, gamma, beta, ws, dgamma, dbeta, n, c, groups, fused_mask, sxw));
    }
  }
  rc = osd_check_launch("gnl_bwd_stats");
  if (rc) return rc;
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(gnl_bwd_apply_kernel<float>, g2, dim3(256), 0, OSD_STREAM(stream), L, ab, gamma, beta, ws, dgamma, dbeta, n, c, groups, sxw, conv_dbias),
      hipLaunchKernelGGL(gnl_bwd_apply_kernel<__bf16>, g2, dim3(256), 0, OSD_STREAM(stream), L, ab, gamma, beta, ws, dgamma, dbeta, n, c, groups, sxw, conv_dbias));
  return osd_check_launch("gnl_bwd_apply");
}

extern "C" int osd_groupnorm_relu_bwd_levels(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                             const int32_t* hws, const float* ab, const float* gamma, const float* beta,
                                             float* ws, float* dgamma, float* dbeta, int n, int c, int groups, int dtype,
                                             void* stream) {
  return gn_bwd_levels_impl(n_levels, us, dts, dus, hws, ab, gamma, beta, ws, dgamma, dbeta, nullptr, n, c, groups, dtype, 0u, stream);
}

extern "C" int osd_groupnorm_relu_bwd_levels_fused(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                                   const int32_t* hws, const float* ab, const float* gamma, const float* beta,
                                                   float* ws, float* dgamma, float* dbeta, int n, int c, int groups, int dtype,
                                                   uint32_t fused_mask, void* stream) {
  return gn_bwd_levels_impl(n_levels, us, dts, dus, hws, ab, gamma, beta, ws, dgamma, dbeta, nullptr, n, c, groups, dtype, fused_mask, stream);
}

extern "C" int osd_groupnorm_relu_bwd_levels_convbias(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                                      const int32_t* hws, const float* ab, const float* gamma, const float* beta,
                                                      float* ws, float* dgamma, float* dbeta, float* conv_dbias, int n, int c,
                                                      int groups, int dtype, void* stream) {
  if (!conv_dbias) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_bwd_levels_convbias: null conv_dbias");
  return gn_bwd_levels_impl(n_levels, us, dts, dus, hws, ab, gamma, beta, ws, dgamma, dbeta, conv_dbias, n, c, groups, dtype, 0u, stream);
}
