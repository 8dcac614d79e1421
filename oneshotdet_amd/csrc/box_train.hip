// box_train.hip — training path of the second-stage few-shot ROI box head: what is not a convolution.
//   * proposal <-> ground-truth matching, fg / bg labels, BalancedPositiveNegativeSampler, BoxCoder.encode
//       modeling/roi_heads/box_head/loss.py:44-141,234-301 (match_targets_to_proposals, prepare_targets, subsample),
//       modeling/matcher.py:52-83, modeling/balanced_positive_negative_sampler.py:19-62, structures/boxlist_ops.py:221-256,
//       modeling/box_coder.py:21-50
//   * the loss: softmax cross-entropy over the 2 classes + smooth-L1 on the positives' class deltas, with the weights 5 / 2.5
//       of box_head.py:193-194 (loss.py:306-381, layers/smooth_l1_loss.py:5-15), values and gradient w.r.t. the predictor
//   * backward of GroupNorm(32) + LeakyReLU(0.2) over ROI maps (box_head.py:43-66) and of the level-routed 7x7 ROIAlign
//       (modeling/poolers.py:93-124, csrc/cuda/ROIAlign_cuda.cu:178-254)
// The convolutions / fully connected layers run forward, data gradient and weight gradient on the implicit-GEMM kernels.
#include "osd_common.h"

namespace {

constexpr int kMaxProps = 8192;     // proposals per image the matcher keeps in LDS (training: 4000 + ground truth)

// IoU of boxlist_ops.py:221-256 in its float32 operation order ("+1" areas); contraction off so thresholds compare alike
__device__ __forceinline__ float iou_plus1(const float* a, const float* b) {
  const float area_a = __fmul_rn(__fadd_rn(__fsub_rn(a[2], a[0]), 1.f), __fadd_rn(__fsub_rn(a[3], a[1]), 1.f));
  const float area_b = __fmul_rn(__fadd_rn(__fsub_rn(b[2], b[0]), 1.f), __fadd_rn(__fsub_rn(b[3], b[1]), 1.f));
  const float lx = fmaxf(a[0], b[0]), ly = fmaxf(a[1], b[1]), rx = fminf(a[2], b[2]), ry = fminf(a[3], b[3]);
  const float w = fmaxf(__fadd_rn(__fsub_rn(rx, lx), 1.f), 0.f), h = fmaxf(__fadd_rn(__fsub_rn(ry, ly), 1.f), 0.f);
  const float inter = __fmul_rn(w, h);
  return __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
}

// One workgroup per image.  labels: -1 ignored (past the count), 0 background (best IoU < thresh), else the matched
// ground truth's label (IoU >= thresh: high == low threshold, allow_low_quality_matches False).  Sampling: the positives
// / negatives with the smallest keys (ties: lower index), at most num_pos_max positives and `batch` in all — a uniformly
// random subset when the keys are uniform randoms, i.e. positive[randperm(n)[:k]] with randperm = argsort(keys).
// The sampled rows come out in ascending proposal order (torch.nonzero(pos | neg), loss.py:292).
__global__ void __launch_bounds__(1024) box_match_sample_kernel(
    const float* __restrict__ boxes, const int32_t* __restrict__ counts, const float* __restrict__ gt,
    const int32_t* __restrict__ gt_count, const int32_t* __restrict__ gt_labels, const float* __restrict__ keys, int P, int G,
    int batch, int num_pos_max, float thresh, float wx, float wy, float ww, float wh, float* __restrict__ s_boxes,
    int32_t* __restrict__ s_labels, float* __restrict__ s_targets, int32_t* __restrict__ s_index, int32_t* __restrict__ s_count,
    int32_t* __restrict__ all_labels, int32_t* __restrict__ all_matched) {
  __shared__ int lab[kMaxProps];
  __shared__ float keyv[kMaxProps];
  __shared__ int part[1024];
  __shared__ int tot[2];
  const int img = blockIdx.x, t = threadIdx.x;
  const int cnt = min(counts ? counts[img] : P, P);
  const int ng = min(gt_count[img], G);
  const float* bx = boxes + (size_t)img * P * 4;
  const float* gb = gt + (size_t)img * G * 4;
  if (t < 2) tot[t] = 0;
  __syncthreads();
  int npos = 0, nneg = 0;
  for (int i = t; i < P; i += 1024) {
    int l = -1, m = -1;
    if (i < cnt && ng > 0) {
      float best = -1.f;
      int arg = 0;
      for (int g = 0; g < ng; ++g) {
        const float v = iou_plus1(gb + g * 4, bx + (size_t)i * 4);
        if (v > best) { best = v; arg = g; }              // first maximum, as Tensor.max(dim=0)
      }
      if (best < thresh) { l = 0; m = -1; }                 // Matcher.BELOW_LOW_THRESHOLD
      else { l = gt_labels ? gt_labels[(size_t)img * G + arg] : 1; m = arg; }
    }
    lab[i] = l;
    keyv[i] = keys[(size_t)img * P + i];
    if (all_labels) all_labels[(size_t)img * P + i] = l;
    if (all_matched) all_matched[(size_t)img * P + i] = m;
    npos += l >= 1;
    nneg += l == 0;
  }
  atomicAdd(&tot[0], npos);
  atomicAdd(&tot[1], nneg);
  __syncthreads();
  const int num_pos = min(tot[0], num_pos_max);
  const int num_neg = min(tot[1], batch - num_pos);
  // The class quota takes the elements with the smallest (key, index).  Instead of ranking every element against every
  // other one (P^2 / 1024 compares per thread: 2.2 ms per step at 4000 proposals), find the quota-th smallest key of either
  // class by building its order-preserving bit pattern from the top bit down (32 counting rounds over the thread's own
  // <= 8 elements), then take everything below it and, in index order, as many of the elements EQUAL to it as still fit.
  constexpr int kPer = kMaxProps / 1024;
  unsigned uk[kPer];
  int cls[kPer];                           // 0 positive, 1 negative, 2 neither
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = t + k * 1024;
    cls[k] = 2;
    uk[k] = 0u;
    if (i < P) {
      const int l = lab[i];
      cls[k] = l >= 1 ? 0 : (l == 0 ? 1 : 2);
      const unsigned bits = __float_as_uint(keyv[i]);
      uk[k] = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);      // float order -> unsigned order
    }
  }
  __shared__ int cntb[3][2];             // rotating: a round adds into its buffer and clears the next round's (last read two rounds ago)
  unsigned T[2] = {0u, 0u};                // quota-th smallest key (mapped) of the positives / negatives
  const int quota[2] = {num_pos, num_neg};
  if (t < 6) cntb[t >> 1][t & 1] = 0;
  __syncthreads();
  for (int bit = 31; bit >= 0; --bit) {
    const int buf = bit % 3, clr = (bit + 2) % 3;      // bit counts down: the NEXT round uses (bit - 1) % 3 == (bit + 2) % 3
    const unsigned c0 = T[0] | (1u << bit), c1 = T[1] | (1u << bit);
    int n0 = 0, n1 = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      n0 += (cls[k] == 0 && uk[k] < c0) ? 1 : 0;
      n1 += (cls[k] == 1 && uk[k] < c1) ? 1 : 0;
    }
    int packed = n0 | (n1 << 16);          // <= 8 per thread, <= 512 per wave: no carry between the halves
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) packed += __shfl_xor(packed, d, 64);
    if ((t & 63) == 0) { atomicAdd(&cntb[buf][0], packed & 0xffff); atomicAdd(&cntb[buf][1], packed >> 16); }
    if (t < 2) cntb[clr][t] = 0;           // read for the last time two rounds ago, i.e. before the previous barrier
    __syncthreads();
    if (cntb[buf][0] < quota[0]) T[0] = c0;           // fewer than quota keys below the candidate: the answer has this bit
    if (cntb[buf][1] < quota[1]) T[1] = c1;
  }
  __syncthreads();
  // flags into keyv[]: 1 = below the threshold key (selected), 2 + class = equal to it (resolved in index order below)
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = t + k * 1024;
    if (i < P) {
      float f = 0.f;
      if (cls[k] < 2 && quota[cls[k]] > 0) {
        if (uk[k] < T[cls[k]]) f = 1.f;
        else if (uk[k] == T[cls[k]]) f = 2.f + (float)cls[k];
      }
      keyv[i] = f;
    }
  }
  __syncthreads();
  // thread t owns a contiguous chunk: counts of (sure, tied positive, tied negative) -> exclusive scans over the threads
  const int per = (P + 1023) / 1024;
  const int lo = min(t * per, P), hi = min(lo + per, P);
  __shared__ int part2[1024];
  __shared__ int sure[2];
  if (t < 2) sure[t] = 0;
  int ns0 = 0, ns1 = 0, ne0 = 0, ne1 = 0;
  for (int i = lo; i < hi; ++i) {
    const float f = keyv[i];
    const bool pos = lab[i] >= 1;
    ns0 += (f == 1.f && pos); ns1 += (f == 1.f && !pos);
    ne0 += f == 2.f; ne1 += f == 3.f;
  }
  part[t] = ne0;
  part2[t] = ne1;
  __syncthreads();
  if (ns0) atomicAdd(&sure[0], ns0);
  if (ns1) atomicAdd(&sure[1], ns1);
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0, v2 = t >= d ? part2[t - d] : 0;
    __syncthreads();
    part[t] += v;
    part2[t] += v2;
    __syncthreads();
  }
  int e0 = part[t] - ne0, e1 = part2[t] - ne1;         // tied elements of either class before this chunk
  const int room0 = num_pos - sure[0], room1 = num_neg - sure[1];
  __syncthreads();
  int c = 0;
  for (int i = lo; i < hi; ++i) {
    const float f = keyv[i];
    float s = f == 1.f ? 1.f : 0.f;
    if (f == 2.f) { s = e0 < room0 ? 1.f : 0.f; ++e0; }
    if (f == 3.f) { s = e1 < room1 ? 1.f : 0.f; ++e1; }
    keyv[i] = s;
    c += s != 0.f;
  }
  part[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int o = part[t] - c;
  const int S = batch;
  for (int i = lo; i < hi; ++i) {
    if (keyv[i] == 0.f) continue;
    if (o < S) {
      const size_t row = (size_t)img * S + o;
      const float* b = bx + (size_t)i * 4;
      const int l = lab[i];
      // matched_targets = target[matched_idxs.clamp(min=0)] (loss.py:70): background rows are encoded against box 0
      int m = 0;
      if (l >= 1) {
        float best = -1.f;
        for (int g = 0; g < ng; ++g) {
          const float v = iou_plus1(gb + g * 4, b);
          if (v > best) { best = v; m = g; }
        }
      }
      const float* r = gb + m * 4;
      // BoxCoder.encode (box_coder.py:21-50), float32 in its order
      const float ew = __fadd_rn(__fsub_rn(b[2], b[0]), 1.f), eh = __fadd_rn(__fsub_rn(b[3], b[1]), 1.f);
      const float ecx = __fadd_rn(b[0], __fmul_rn(0.5f, ew)), ecy = __fadd_rn(b[1], __fmul_rn(0.5f, eh));
      const float gw = __fadd_rn(__fsub_rn(r[2], r[0]), 1.f), gh = __fadd_rn(__fsub_rn(r[3], r[1]), 1.f);
      const float gcx = __fadd_rn(r[0], __fmul_rn(0.5f, gw)), gcy = __fadd_rn(r[1], __fmul_rn(0.5f, gh));
      float* tg = s_targets + row * 4;
      tg[0] = __fdiv_rn(__fmul_rn(wx, __fsub_rn(gcx, ecx)), ew);
      tg[1] = __fdiv_rn(__fmul_rn(wy, __fsub_rn(gcy, ecy)), eh);
      tg[2] = __fmul_rn(ww, logf(__fdiv_rn(gw, ew)));
      tg[3] = __fmul_rn(wh, logf(__fdiv_rn(gh, eh)));
      float* sb = s_boxes + row * 4;
      sb[0] = b[0]; sb[1] = b[1]; sb[2] = b[2]; sb[3] = b[3];
      s_labels[row] = l;
      s_index[row] = i;
    }
    ++o;
  }
  __syncthreads();
  const int kept = min(part[1023], S);
  for (int j = kept + t; j < S; j += 1024) {        // rows past the count: zero boxes, label -1
    const size_t row = (size_t)img * S + j;
    for (int q = 0; q < 4; ++q) { s_boxes[row * 4 + q] = 0.f; s_targets[row * 4 + q] = 0.f; }
    s_labels[row] = -1;
    s_index[row] = -1;
  }
  if (t == 0) s_count[img] = kept;
}

// Loss of the sampled ROIs (loss.py:306-381 with gt_label == -1, 'ce_loss', class-specific regression; weights of
// box_head.py:193-194 folded in) and its gradient w.r.t. the predictor's output.  One workgroup; fixed summation order.
template <typename T>
__global__ void __launch_bounds__(1024) box_loss_kernel(const T* __restrict__ pred, const int32_t* __restrict__ labels,
                                                        const float* __restrict__ targets, const int32_t* __restrict__ s_count,
                                                        int n_img, int S, int pstride, float w_cls, float w_box,
                                                        float* __restrict__ losses, T* __restrict__ d_pred, int gstride) {
  __shared__ float red[2][1024];
  __shared__ int nval, bad_label;
  const int t = threadIdx.x;
  if (t == 0) {
    int n = 0;
    for (int i = 0; i < n_img; ++i) n += min(s_count[i], S);
    nval = n;
    bad_label = 0;
  }
  __syncthreads();
  const int M = n_img * S;
  const float inv_n = nval > 0 ? 1.f / (float)nval : 0.f;
  float lc = 0.f, lb = 0.f;
  for (int r = t; r < M; r += 1024) {
    const int img = r / S, ri = r - img * S;
    const bool valid = ri < min(s_count[img], S);
    T* g = d_pred ? d_pred + (size_t)r * gstride : nullptr;
    if (g) for (int q = 0; q < gstride; ++q) g[q] = from_f32<T>(0.f);
    if (!valid) continue;
    const T* p = pred + (size_t)r * pstride;
    // two classes (background / the queried object: ROI_BOX_HEAD.NUM_CLASSES = 2 in the config of record); a row is
    // 2 logits + 2 x 4 deltas.  A label > 1 has no columns in the row: nothing is read or written for it and the
    // losses come back NaN (no host synchronisation to report it any other way)
    if (labels[r] > 1) { bad_label = 1; continue; }
    const int l = labels[r] > 0 ? labels[r] : 0;
    const float x0 = to_f32(p[0]), x1 = to_f32(p[1]);
    const float m = fmaxf(x0, x1);
    const float e0 = expf(x0 - m), e1 = expf(x1 - m);
    const float lse = m + logf(e0 + e1);
    lc += lse - (l == 1 ? x1 : x0);
    if (g) {
      const float s0 = e0 / (e0 + e1), s1 = e1 / (e0 + e1);
      g[0] = from_f32<T>(w_cls * inv_n * (s0 - (l == 0 ? 1.f : 0.f)));
      g[1] = from_f32<T>(w_cls * inv_n * (s1 - (l == 1 ? 1.f : 0.f)));
    }
    if (l >= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = to_f32(p[2 + 4 * l + k]) - targets[(size_t)r * 4 + k];
        const float n = fabsf(d);
        lb += n < 1.f ? 0.5f * n * n : n - 0.5f;                       // smooth_l1_loss(beta = 1), summed
        if (g) g[2 + 4 * l + k] = from_f32<T>(w_box * inv_n * (n < 1.f ? d : (d > 0.f ? 1.f : -1.f)));
      }
    }
  }
  red[0][t] = lc;
  red[1][t] = lb;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
    __syncthreads();
  }
  if (t == 0) {
    const float poison = bad_label ? __builtin_nanf("") : 0.f;
    losses[0] = w_cls * red[0][0] * inv_n + poison;     // 5 * F.cross_entropy(class_logits, labels)
    losses[1] = w_box * red[1][0] * inv_n + poison;     // 2.5 * smooth_l1(sum) / labels.numel()
    losses[2] = (float)nval;
  }
}

__device__ __forceinline__ void ld2(const float* p, float& a, float& b) {
  const f32x2 t = *reinterpret_cast<const f32x2*>(p);
  a = t[0]; b = t[1];
}
__device__ __forceinline__ void ld2(const __bf16* p, float& a, float& b) {
  const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
  a = __uint_as_float(u << 16);
  b = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ void st2(float* p, float a, float b) {
  f32x2 t = {a, b};
  *reinterpret_cast<f32x2*>(p) = t;
}
__device__ __forceinline__ void st2(__bf16* p, float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 t = {(__bf16)a, (__bf16)b};
  *reinterpret_cast<bf16x2*>(p) = t;
}

// Backward of y = LeakyReLU(GroupNorm(x [+ addend])) for one ROI map per workgroup (the mirror of gn_act_rois_kernel:
// thread t owns channels 2t, 2t+1 of all HW pixels in registers; statistics recomputed from x, exact two-pass).
// dx = rstd * (dxh - mean_g(dxh) - xh * mean_g(dxh * xh)), dxh = dz * gamma, dz = dy * (z >= 0 ? 1 : slope).
// part[sample][0][c] = sum_p dz * xh (d gamma), part[sample][1][c] = sum_p dz (d beta): folded by gn_rois_param_reduce.
template <typename T, int HW>
__global__ __launch_bounds__(256) void gn_act_rois_bwd_kernel(const T* __restrict__ x, const T* __restrict__ addend,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const T* __restrict__ dy, T* __restrict__ dx,
                                                              float* __restrict__ part, int c, int groups, float eps,
                                                              float slope, int rois_per_add, int add_stride, int add_offset) {
  const int sample = blockIdx.x;
  const int t = threadIdx.x;                 // blockDim.x == c / 2
  const int cpg = c / groups, lanes = cpg / 2;
  const T* xs = x + (size_t)sample * HW * c + 2 * t;
  float v0[HW], v1[HW];
#pragma unroll
  for (int p = 0; p < HW; ++p) ld2(xs + (size_t)p * c, v0[p], v1[p]);
  if (addend) {
    const T* as = addend + ((size_t)(sample / rois_per_add) * add_stride + add_offset) * HW * c + 2 * t;
#pragma unroll
    for (int p = 0; p < HW; ++p) {
      float a0, a1;
      ld2(as + (size_t)p * c, a0, a1);
      v0[p] += a0;
      v1[p] += a1;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < HW; ++p) s += v0[p] + v1[p];
  for (int m = 1; m < lanes; m <<= 1) s += __shfl_xor(s, m);
  const float inv_n = 1.f / (float)(HW * cpg);
  const float mean = s * inv_n;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < HW; ++p) {
    const float d0 = v0[p] - mean, d1 = v1[p] - mean;
    q += d0 * d0 + d1 * d1;
  }
  for (int m = 1; m < lanes; m <<= 1) q += __shfl_xor(q, m);
  const float rstd = rsqrtf(q * inv_n + eps);
  const float g0 = gamma[2 * t], g1 = gamma[2 * t + 1], b0 = beta[2 * t], b1 = beta[2 * t + 1];
  const T* ds = dy + (size_t)sample * HW * c + 2 * t;
  float s1 = 0.f, s2 = 0.f, dg0 = 0.f, dg1 = 0.f, db0 = 0.f, db1 = 0.f;
  // v becomes xhat, then dxhat is kept in w
  float w0[HW], w1[HW];
#pragma unroll
  for (int p = 0; p < HW; ++p) {
    const float xh0 = (v0[p] - mean) * rstd, xh1 = (v1[p] - mean) * rstd;
    float d0, d1;
    ld2(ds + (size_t)p * c, d0, d1);
    const float z0 = fmaf(xh0, g0, b0), z1 = fmaf(xh1, g1, b1);
    const float dz0 = z0 >= 0.f ? d0 : d0 * slope, dz1 = z1 >= 0.f ? d1 : d1 * slope;
    dg0 += dz0 * xh0; dg1 += dz1 * xh1; db0 += dz0; db1 += dz1;
    const float h0 = dz0 * g0, h1 = dz1 * g1;
    s1 += h0 + h1;
    s2 += h0 * xh0 + h1 * xh1;
    v0[p] = xh0; v1[p] = xh1; w0[p] = h0; w1[p] = h1;
  }
  for (int m = 1; m < lanes; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
  const float m1 = s1 * inv_n, m2 = s2 * inv_n;
  T* os = dx + (size_t)sample * HW * c + 2 * t;
#pragma unroll
  for (int p = 0; p < HW; ++p)
    st2(os + (size_t)p * c, rstd * (w0[p] - m1 - v0[p] * m2), rstd * (w1[p] - m1 - v1[p] * m2));
  float* pp = part + (size_t)sample * 2 * c;
  pp[2 * t] = dg0; pp[2 * t + 1] = dg1;
  pp[c + 2 * t] = db0; pp[c + 2 * t + 1] = db1;
}

// dgamma[ch] += sum_samples part[s][0][ch]; dbeta likewise.  32 channels x 32 sample lanes per workgroup: a lane adds the
// samples s = lane, lane + 32, ... in order, the 32 lane sums are folded in a fixed tree (deterministic).  (One thread per
// channel walking all 1,024 samples was a 330 us serial chain of dependent strided loads, three times per step.)
__global__ void __launch_bounds__(1024) gn_rois_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int n_samples, int c) {
  __shared__ float red[32][33];
  const int chl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + chl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ch < 2 * c) {
    int i = sl;
    for (; i + 96 < n_samples; i += 128) {          // four independent loads in flight
      s0 += part[(size_t)i * 2 * c + ch];
      s1 += part[(size_t)(i + 32) * 2 * c + ch];
      s2 += part[(size_t)(i + 64) * 2 * c + ch];
      s3 += part[(size_t)(i + 96) * 2 * c + ch];
    }
    for (; i < n_samples; i += 32) s0 += part[(size_t)i * 2 * c + ch];
  }
  red[sl][chl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int d = 16; d >= 1; d >>= 1) {
    if (sl < d) red[sl][chl] += red[sl + d][chl];
    __syncthreads();
  }
  if (sl == 0 && ch < 2 * c) {
    if (ch < c) dgamma[ch] += red[0][chl];
    else dbeta[ch - c] += red[0][chl];
  }
}

// out[img][e] = sum over the ROIs r < rois_per_image of x[img * rois_per_image + r][e]   (e over hw * c elements)
template <typename T>
__global__ void rois_sum_kernel(const T* __restrict__ x, T* __restrict__ out, int rois_per_image, long long elems) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int img = blockIdx.y;
  if (i >= elems) return;
  float s = 0.f;
  const T* p = x + (size_t)img * rois_per_image * elems + i;
  for (int r = 0; r < rois_per_image; ++r) s += to_f32(p[(size_t)r * elems]);
  out[(size_t)img * elems + i] = from_f32<T>(s);
}

struct PoolGradLevels {
  float* gx[OSD_MAX_ROI_LEVELS];
  int h[OSD_MAX_ROI_LEVELS];
  int w[OSD_MAX_ROI_LEVELS];
  float scale[OSD_MAX_ROI_LEVELS];
  int n_levels;
};

// Backward of roi_pool_levels_kernel (box_head.hip): the same level routing and sample geometry; every sample's
// gradient dy / count goes to its four taps with the bilinear weights (ROIAlign_cuda.cu:178-254), fp32 atomics: up to 784
// per ROI and channel, 0.8 GB of memory-side atomic traffic per step at 0.72 ms = the rate those run at (before the
// per-axis tap merge below: 0.30-0.52 ms).  Measured alternatives, both
// slower (tools/roi_bwd_bench.py): summing a ROI's footprint in LDS first (helps only ROIs narrower than ~14 map pixels; the
// 64 KB LDS image costs the direct path its occupancy: 0.99 ms) and a per-tile gather without atomics (one thread per
// channel walking the image's ROIs: a chain of dependent dy loads per contributing cell, 0.9-3.8 ms).
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_levels_bwd_kernel(PoolGradLevels lv, const float* __restrict__ boxes,
                                                                  const int32_t* __restrict__ counts, const T* __restrict__ dy,
                                                                  int c, int max_rois, int pool, int sampling, int dy_stride) {
  const int roi = blockIdx.x;
  const int img = roi / max_rois, ri = roi % max_rois;
  if (counts != nullptr && ri >= counts[img]) return;
  const float* bx = boxes + (size_t)roi * 4;
  const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  const int k_min = 3, k_max = 3 + lv.n_levels - 1;
  int l;
  {
    const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
    float f = floorf(4.f + log2f(sqrtf(area) / 224.f + 1e-6f));
    f = fminf(fmaxf(f, (float)k_min), (float)k_max);
    l = (int)f - k_min;
  }
  int h = lv.h[0], w = lv.w[0];
  float scale = lv.scale[0];
  float* gxl = lv.gx[0];
#pragma unroll
  for (int i = 1; i < OSD_MAX_ROI_LEVELS; ++i)
    if (l == i) { h = lv.h[i]; w = lv.w[i]; scale = lv.scale[i]; gxl = lv.gx[i]; }
  float* gx = gxl + (size_t)img * h * w * c;
  const float rsw = x1 * scale, rsh = y1 * scale, rew = x2 * scale, reh = y2 * scale;
  const float roi_w = fmaxf(rew - rsw, 1.f), roi_h = fmaxf(reh - rsh, 1.f);
  const float bin_h = roi_h / (float)pool, bin_w = roi_w / (float)pool;
  const int gh = sampling > 0 ? sampling : (int)ceilf(roi_h / pool);
  const int gw = sampling > 0 ? sampling : (int)ceilf(roi_w / pool);
  const float inv_count = 1.f / (float)(gh * gw);
  const int items = pool * pool * c;
  const T* dr = dy + (size_t)roi * pool * pool * dy_stride;
  // Bilinear weights are separable: sample (iy, ix) adds wy[iy][.] x wx[ix][.] x g, so a cell's gh x gw samples add
  // (sum over iy of wy) x (sum over ix of wx).  With the config's sampling ratio 2 the two samples of an axis are half a bin
  // apart — less than a pixel for every ROI up to 14 map pixels wide — and share taps: merging equal pixel indices per axis
  // first leaves 3 x 3 (or fewer) atomics per cell and channel instead of 16.
  const bool merge = gh <= 2 && gw <= 2;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int cell = it / c, ch = it - cell * c;
    const int py = cell / pool, px = cell % pool;
    const float g = to_f32(dr[(size_t)cell * dy_stride + ch]) * inv_count;
    if (merge) {
      int iyv[4], ixv[4];
      float wyv[4], wxv[4];
      int ny = 0, nx = 0;
      auto axis = [](float v0, int size, int* idx, float* wt, int& n) {     // one sample's two taps, merged into the list
        float v = v0;
        if (v < -1.0f || v > (float)size) return;                             // the sample lies outside: no contribution
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
        const float fr = v - lo;
        const int ids[2] = {lo, hi};
        const float ws[2] = {1.f - fr, fr};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bool found = false;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < n && idx[q] == ids[t]) { wt[q] += ws[t]; found = true; }
          if (!found) { idx[n] = ids[t]; wt[n] = ws[t]; ++n; }
        }
      };
      for (int iy = 0; iy < gh; ++iy) axis(rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh, h, iyv, wyv, ny);
      for (int ix = 0; ix < gw; ++ix) axis(rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw, w, ixv, wxv, nx);
      for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b)
          atomicAdd(gx + ((size_t)iyv[a] * w + ixv[b]) * c + ch, wyv[a] * wxv[b] * g);
      continue;
    }
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw;
        float yv = yy, xv = xx;
        if (yv < -1.0f || yv > (float)h || xv < -1.0f || xv > (float)w) continue;
        if (yv <= 0.f) yv = 0.f;
        if (xv <= 0.f) xv = 0.f;
        int yl = (int)yv, xl = (int)xv, yh, xh;
        if (yl >= h - 1) { yh = yl = h - 1; yv = (float)yl; } else { yh = yl + 1; }
        if (xl >= w - 1) { xh = xl = w - 1; xv = (float)xl; } else { xh = xl + 1; }
        const float ly = yv - yl, lx = xv - xl, hy = 1.f - ly, hx = 1.f - lx;
        atomicAdd(gx + ((size_t)yl * w + xl) * c + ch, hy * hx * g);
        atomicAdd(gx + ((size_t)yl * w + xh) * c + ch, hy * lx * g);
        atomicAdd(gx + ((size_t)yh * w + xl) * c + ch, ly * hx * g);
        atomicAdd(gx + ((size_t)yh * w + xh) * c + ch, ly * lx * g);
      }
    }
  }
}

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int osd_box_match_sample(const float* boxes, const int32_t* counts, const float* gt_boxes, const int32_t* gt_count,
                                    const int32_t* gt_labels, const float* keys, int n, int max_props, int max_gt,
                                    int batch_per_image, float positive_fraction, float iou_thresh, const float* reg_weights,
                                    float* s_boxes, int32_t* s_labels, float* s_targets, int32_t* s_index, int32_t* s_count,
                                    int32_t* all_labels, int32_t* all_matched, void* stream) {
  if (!boxes || !gt_boxes || !gt_count || !keys || !reg_weights || !s_boxes || !s_labels || !s_targets || !s_index || !s_count)
    return osd_fail(OSD_ERR_INVALID_ARG, "box_match_sample: null argument");
  if (n == 0) return OSD_OK;
  if (max_props <= 0 || max_props > kMaxProps) return osd_fail(OSD_ERR_UNSUPPORTED, "box_match_sample: 1..%d proposals per image", kMaxProps);
  if (max_gt <= 0 || batch_per_image <= 0 || batch_per_image > max_props)
    return osd_fail(OSD_ERR_INVALID_ARG, "box_match_sample: bad sizes");
  const int num_pos = (int)(batch_per_image * positive_fraction);      // int(self.batch_size_per_image * self.positive_fraction)
  hipLaunchKernelGGL(box_match_sample_kernel, dim3(n), dim3(1024), 0, OSD_STREAM(stream), boxes, counts, gt_boxes, gt_count,
                     gt_labels, keys, max_props, max_gt, batch_per_image, num_pos, iou_thresh, reg_weights[0], reg_weights[1],
                     reg_weights[2], reg_weights[3], s_boxes, s_labels, s_targets, s_index, s_count, all_labels, all_matched);
  return osd_check_launch("box_match_sample");
}

extern "C" int osd_box_loss(const void* pred, const int32_t* labels, const float* targets, const int32_t* s_count, int n,
                            int rois_per_image, int pred_stride, float w_cls, float w_box, float* losses, void* d_pred,
                            int grad_stride, int dtype, void* stream) {
  if (!pred || !labels || !targets || !s_count || !losses) return osd_fail(OSD_ERR_INVALID_ARG, "box_loss: null argument");
  if (pred_stride < 10 || (d_pred && grad_stride < 10)) return osd_fail(OSD_ERR_INVALID_ARG, "box_loss: 2 logits + 8 deltas per row");
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(box_loss_kernel<float>, dim3(1), dim3(1024), 0, OSD_STREAM(stream), (const float*)pred, labels, targets,
                       s_count, n, rois_per_image, pred_stride, w_cls, w_box, losses, (float*)d_pred, grad_stride);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL(box_loss_kernel<__bf16>, dim3(1), dim3(1024), 0, OSD_STREAM(stream), (const __bf16*)pred, labels, targets,
                       s_count, n, rois_per_image, pred_stride, w_cls, w_box, losses, (__bf16*)d_pred, grad_stride);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "box_loss: bad dtype");
  return osd_check_launch("box_loss");
}

extern "C" int osd_groupnorm_act_rois_bwd(const void* x, const void* addend, const float* gamma, const float* beta,
                                          const void* dy, void* dx, float* part_ws, float* dgamma, float* dbeta,
                                          int n_samples, int hw, int c, int groups, float eps, float slope,
                                          int rois_per_add, int add_stride, int add_offset, int dtype, void* stream) {
  if (!x || !gamma || !beta || !dy || !dx || !part_ws || !dgamma || !dbeta)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_act_rois_bwd: null argument");
  if (n_samples == 0) return OSD_OK;
  if (hw != 49) return osd_fail(OSD_ERR_UNSUPPORTED, "groupnorm_act_rois_bwd: 7x7 ROI maps only (hw = %d)", hw);
  const int cpg = groups > 0 ? c / groups : 0;
  if (groups <= 0 || c % groups || cpg % 2 || (cpg / 2) > 64 || ((cpg / 2) & (cpg / 2 - 1)) || c / 2 > 256 || c % 2)
    return osd_fail(OSD_ERR_UNSUPPORTED, "groupnorm_act_rois_bwd: c = %d, groups = %d not supported", c, groups);
  if (rois_per_add <= 0) rois_per_add = 1;
  hipStream_t st = OSD_STREAM(stream);
  if (dtype == OSD_F32)
    hipLaunchKernelGGL((gn_act_rois_bwd_kernel<float, 49>), dim3(n_samples), dim3(c / 2), 0, st, (const float*)x, (const float*)addend,
                       gamma, beta, (const float*)dy, (float*)dx, part_ws, c, groups, eps, slope, rois_per_add, add_stride, add_offset);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL((gn_act_rois_bwd_kernel<__bf16, 49>), dim3(n_samples), dim3(c / 2), 0, st, (const __bf16*)x,
                       (const __bf16*)addend, gamma, beta, (const __bf16*)dy, (__bf16*)dx, part_ws, c, groups, eps, slope,
                       rois_per_add, add_stride, add_offset);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_act_rois_bwd: bad dtype");
  int rc = osd_check_launch("groupnorm_act_rois_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_rois_param_reduce_kernel, dim3(cdiv(2 * c, 32)), dim3(1024), 0, st, (const float*)part_ws, dgamma, dbeta,
                     n_samples, c);
  return osd_check_launch("groupnorm_act_rois_bwd(reduce)");
}

extern "C" int osd_rois_sum(const void* x, void* out, int n, int rois_per_image, int64_t elems, int dtype, void* stream) {
  if (!x || !out) return osd_fail(OSD_ERR_INVALID_ARG, "rois_sum: null argument");
  if (n == 0 || elems == 0) return OSD_OK;
  dim3 grid((unsigned)((elems + 255) / 256), n);
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(rois_sum_kernel<float>, grid, dim3(256), 0, OSD_STREAM(stream), (const float*)x, (float*)out, rois_per_image, (long long)elems);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL(rois_sum_kernel<__bf16>, grid, dim3(256), 0, OSD_STREAM(stream), (const __bf16*)x, (__bf16*)out, rois_per_image, (long long)elems);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "rois_sum: bad dtype");
  return osd_check_launch("rois_sum");
}

extern "C" int osd_roi_pool_levels_bwd(int n_levels, float* const* gxs, const int32_t* hs, const int32_t* ws, const float* scales,
                                       const float* boxes, const int32_t* counts, const void* dy, int n, int c, int max_rois,
                                       int pool, int sampling_ratio, int dy_stride, int dtype, void* stream) {
  if (n_levels < 1 || n_levels > OSD_MAX_ROI_LEVELS || !gxs || !hs || !ws || !scales || !boxes || !dy)
    return osd_fail(OSD_ERR_INVALID_ARG, "roi_pool_levels_bwd: bad arguments");
  if (n == 0 || max_rois == 0) return OSD_OK;
  PoolGradLevels lv;
  lv.n_levels = n_levels;
  for (int i = 0; i < OSD_MAX_ROI_LEVELS; ++i) {
    const int j = i < n_levels ? i : 0;
    if (!gxs[j]) return osd_fail(OSD_ERR_INVALID_ARG, "roi_pool_levels_bwd: null level map");
    lv.gx[i] = gxs[j]; lv.h[i] = hs[j]; lv.w[i] = ws[j]; lv.scale[i] = scales[j];
  }
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(roi_pool_levels_bwd_kernel<float>, dim3(n * max_rois), dim3(256), 0, OSD_STREAM(stream), lv, boxes, counts,
                       (const float*)dy, c, max_rois, pool, sampling_ratio, dy_stride);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL(roi_pool_levels_bwd_kernel<__bf16>, dim3(n * max_rois), dim3(256), 0, OSD_STREAM(stream), lv, boxes, counts,
                       (const __bf16*)dy, c, max_rois, pool, sampling_ratio, dy_stride);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "roi_pool_levels_bwd: bad dtype");
  return osd_check_launch("roi_pool_levels_bwd");
}
