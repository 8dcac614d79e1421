// FCOS training loss on device (R12): target assignment with centre sampling + size ranges
// (modeling/rpn/fcos/loss.py:52-204), sigmoid focal loss in the CUDA-kernel form
// (csrc/cuda/SigmoidFocalLoss_cuda.cu:21-101), GIoU loss weighted by the centerness target
// (layers/iou_loss.py:10-49, loss.py:263-267) and BCE-with-logits centerness loss (loss.py:268-271).
// Two passes per FPN level over [N][H*W] locations (targets are recomputed, nothing per-location is stored):
//   pass A accumulates  sums = {num_pos, sum_w, sum_focal, sum_w*(1-giou), sum_bce}
//   pass B writes the gradients w.r.t. (logit, centerness) and w.r.t. s = scale*bbox_pred (pre-exp), using
//          cls = sum_focal/(num_pos+N), reg = sum_wg/sum_w, ctr = sum_bce/num_pos, total = cls + reg + ctr.
#include "osd_common.h"

namespace {

constexpr float kInf = 100000000.f;

struct LossLevel {
  int h, w, stride;
  float lo, hi;       // object size range of the level (loss.py:102-108)
  float radius_px;    // stride * POS_RADIUS
};

struct Target {
  int label;
  float l, t, r, b;
};

__device__ __forceinline__ Target assign_target(float x, float y, const float* __restrict__ gt, int ng, const LossLevel lv) {
  Target out;
  out.label = 0;
  out.l = out.t = out.r = out.b = 0.f;
  float best = kInf;
  // quirk of get_sample_region (loss.py:58-60): no sampling region at all when the first box's centre x is 0
  const bool no_region = (ng == 0) || ((gt[0] + gt[2]) * 0.5f == 0.f);
  for (int g = 0; g < ng; ++g) {
    const float x1 = gt[g * 4 + 0], y1 = gt[g * 4 + 1], x2 = gt[g * 4 + 2], y2 = gt[g * 4 + 3];
    const float l = x - x1, t = y - y1, r = x2 - x, b = y2 - y;
    if (g == 0) { out.l = l; out.t = t; out.r = r; out.b = b; }   // argmin over all-INF rows is index 0
    if (no_region) continue;
    const float cx = (x1 + x2) * 0.5f, cy = (y1 + y2) * 0.5f;
    const float xmin = cx - lv.radius_px, ymin = cy - lv.radius_px, xmax = cx + lv.radius_px, ymax = cy + lv.radius_px;
    const float c1 = xmin > x1 ? xmin : x1, c2 = ymin > y1 ? ymin : y1;
    const float c3 = xmax > x2 ? x2 : xmax, c4 = ymax > y2 ? y2 : ymax;
    const bool inside = fminf(fminf(x - c1, y - c2), fminf(c3 - x, c4 - y)) > 0.f;
    const float mx = fmaxf(fmaxf(l, t), fmaxf(r, b));
    const bool cared = (mx >= lv.lo) && (mx <= lv.hi);
    const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
    if (inside && cared && area < best) {
      best = area;
      out.label = 1;
      out.l = l; out.t = t; out.r = r; out.b = b;
    }
  }
  return out;
}

__device__ __forceinline__ float focal_value(float x, int label, float gamma, float alpha) {
  const float p = 1.f / (1.f + expf(-x));
  if (label == 1) return -alpha * powf(1.f - p, gamma) * logf(fmaxf(p, 1.17549435e-38f));
  const float ge = x >= 0.f ? 1.f : 0.f;
  return -(1.f - alpha) * powf(p, gamma) * (-1.f * x * ge - logf(1.f + expf(x - 2.f * x * ge)));
}

__device__ __forceinline__ float focal_grad(float x, int label, float gamma, float alpha) {
  const float p = 1.f / (1.f + expf(-x));
  if (label == 1) return -alpha * powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(fmaxf(p, 1.17549435e-38f))));
  const float ge = x >= 0.f ? 1.f : 0.f;
  return -(1.f - alpha) * powf(p, gamma) * ((-1.f * x * ge - logf(1.f + expf(x - 2.f * x * ge))) * (1.f - p) * gamma - p);
}

// GIoU loss 1 - giou and its gradient w.r.t. the predicted distances (layers/iou_loss.py:10-43)
__device__ __forceinline__ float giou_loss(const float pd[4], const Target& tg, float grad[4]) {
  const float pl = pd[0], pt = pd[1], pr = pd[2], pb = pd[3];
  const float ta = (tg.l + tg.r) * (tg.t + tg.b), pa = (pl + pr) * (pt + pb);
  const float wi = fminf(pl, tg.l) + fminf(pr, tg.r), gw = fmaxf(pl, tg.l) + fmaxf(pr, tg.r);
  const float hi = fminf(pb, tg.b) + fminf(pt, tg.t), gh = fmaxf(pb, tg.b) + fmaxf(pt, tg.t);
  const float ac = gw * gh + 1e-7f, ai = wi * hi, au = ta + pa - ai;
  const float iou = (ai + 1.f) / (au + 1.f);
  const float giou = iou - (ac - au) / ac;
  if (grad) {
    // d/dp of min(p,t) is 1 when p < t (0.5 on ties, as autograd), of max(p,t) 1 when p > t
    const float tt[4] = {tg.l, tg.t, tg.r, tg.b};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float p = pd[k], t = tt[k];
      const float dmin = p < t ? 1.f : (p == t ? 0.5f : 0.f), dmax = p > t ? 1.f : (p == t ? 0.5f : 0.f);
      const bool horiz = (k == 0 || k == 2);
      const float dai = horiz ? hi * dmin : wi * dmin;
      const float dac = horiz ? gh * dmax : gw * dmax;
      const float dpa = horiz ? (pt + pb) : (pl + pr);
      const float dau = dpa - dai;
      const float diou = (dai * (au + 1.f) - (ai + 1.f) * dau) / ((au + 1.f) * (au + 1.f));
      const float dgiou = diou + (dau * ac - au * dac) / (ac * ac);
      grad[k] = -dgiou;
    }
  }
  return 1.f - giou;
}

template <typename T, int PHASE>
__device__ __forceinline__ void fcos_loss_body(const T* __restrict__ cls_ctr, const T* __restrict__ reg,
                                                        const float* __restrict__ gt, const int* __restrict__ gt_count,
                                                        int max_gt, LossLevel lv, int n_images, float gamma, float alpha,
                                                        const float* __restrict__ scale_dev, float* __restrict__ sums,
                                                        T* __restrict__ d_cls_ctr, T* __restrict__ d_reg, int gstride,
                                                        float* __restrict__ d_scale_raw, int bx, int nbx) {
  const float scale = scale_dev ? *scale_dev : 1.f;
  const int img = blockIdx.y;
  const int hw = lv.h * lv.w;
  const int ng = min(gt_count[img], max_gt);
  const float* g = gt + (size_t)img * max_gt * 4;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float dsc = 0.f;
  float inv_cls = 0.f, inv_w = 0.f, inv_pos = 0.f;
  if (PHASE == 1) {
    const float npos = sums[0], sw = sums[1];
    inv_cls = 1.f / (npos + (float)n_images);
    inv_w = sw > 0.f ? 1.f / sw : (npos > 0.f ? 1.f / npos : 0.f);   // iou_loss.py:46-49: weighted mean, else plain mean
    inv_pos = npos > 0.f ? 1.f / npos : 0.f;
  }
  for (int i = bx * blockDim.x + threadIdx.x; i < hw; i += nbx * blockDim.x) {
    const int yy = i / lv.w, xx = i - yy * lv.w;
    const float x = (float)(xx * lv.stride + lv.stride / 2), y = (float)(yy * lv.stride + lv.stride / 2);
    const Target tg = assign_target(x, y, g, ng, lv);
    const size_t pix = (size_t)img * hw + i;
    const float logit = to_f32(cls_ctr[pix * 4 + 0]);
    if (PHASE == 0) {
      acc[2] += focal_value(logit, tg.label, gamma, alpha);
      if (tg.label) {
        const float pd[4] = {to_f32(reg[pix * 4 + 0]), to_f32(reg[pix * 4 + 1]), to_f32(reg[pix * 4 + 2]), to_f32(reg[pix * 4 + 3])};
        const float wgt = sqrtf((fminf(tg.l, tg.r) / fmaxf(tg.l, tg.r)) * (fminf(tg.t, tg.b) / fmaxf(tg.t, tg.b)));
        const float c = to_f32(cls_ctr[pix * 4 + 1]);
        acc[0] += 1.f;
        acc[1] += wgt;
        acc[3] += wgt * giou_loss(pd, tg, nullptr);
        acc[4] += fmaxf(c, 0.f) - c * wgt + log1pf(expf(-fabsf(c)));        // BCEWithLogits
      }
    } else {
      float dl = focal_grad(logit, tg.label, gamma, alpha) * inv_cls, dc = 0.f;
      float dr[4] = {0.f, 0.f, 0.f, 0.f};
      if (tg.label) {
        const float pd[4] = {to_f32(reg[pix * 4 + 0]), to_f32(reg[pix * 4 + 1]), to_f32(reg[pix * 4 + 2]), to_f32(reg[pix * 4 + 3])};
        const float wgt = sqrtf((fminf(tg.l, tg.r) / fmaxf(tg.l, tg.r)) * (fminf(tg.t, tg.b) / fmaxf(tg.t, tg.b)));
        const float c = to_f32(cls_ctr[pix * 4 + 1]);
        float gg[4];
        giou_loss(pd, tg, gg);
        const float wsel = sums[1] > 0.f ? wgt : 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float ds = gg[k] * wsel * inv_w * pd[k];      // chain through reg = exp(s): ds = dreg * reg
          dsc += ds * logf(pd[k]);                            // d scale = sum ds * x, x = log(reg)/scale (divided later)
          dr[k] = ds * scale;                                 // gradient w.r.t. the conv output x = bbox_pred
        }
        dc = (1.f / (1.f + expf(-c)) - wgt) * inv_pos;
      }
      T* o = d_cls_ctr + pix * gstride;
      o[0] = from_f32<T>(dl); o[1] = from_f32<T>(dc);
      T* q = d_reg + pix * gstride;
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = from_f32<T>(dr[k]);
    }
  }
  // block reduction of the accumulators
  __shared__ float red[6][256];
  if (PHASE == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = acc[k];
  } else {
    red[5][threadIdx.x] = dsc;
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      if (PHASE == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
      } else {
        red[5][threadIdx.x] += red[5][threadIdx.x + s];
      }
    }
    __syncthreads();
  }
  // the five sums of a workgroup go out as ONE atomic wave-instruction (lanes 0..4 on five consecutive words of one line): as five
  // instructions of thread 0 the ~550 workgroups of a launch queued 2,700 same-line operations at the memory-side atomic unit, and the
  // statistics phase took 45 us against 14 - 24 for the gradient phase that does more arithmetic (profiles/r6_bench_train_bf16_kernel_stats.csv)
  if (PHASE == 0) {
    if (threadIdx.x < 5) atomicAdd(sums + threadIdx.x, red[threadIdx.x][0]);
  } else if (threadIdx.x == 0) {
    atomicAdd(d_scale_raw, red[5][0]);
  }
}

template <typename T, int PHASE>
__global__ void __launch_bounds__(256) fcos_loss_kernel(const T* __restrict__ cls_ctr, const T* __restrict__ reg,
                                                        const float* __restrict__ gt, const int* __restrict__ gt_count,
                                                        int max_gt, LossLevel lv, int n_images, float gamma, float alpha,
                                                        const float* __restrict__ scale_dev, float* __restrict__ sums,
                                                        T* __restrict__ d_cls_ctr, T* __restrict__ d_reg, int gstride,
                                                        float* __restrict__ d_scale_raw) {
  fcos_loss_body<T, PHASE>(cls_ctr, reg, gt, gt_count, max_gt, lv, n_images, gamma, alpha, scale_dev, sums, d_cls_ctr, d_reg,
                           gstride, d_scale_raw, blockIdx.x, gridDim.x);
}

// all FPN levels in one launch: blockIdx.z = level (the per-level launches sit on the critical path between forward and
// backward, each a few microseconds of work behind a launch gap)
constexpr int kLossLevels = 6;
struct LossLevels {
  const void* cc[kLossLevels]; const void* rg[kLossLevels]; void* dcc[kLossLevels]; void* drg[kLossLevels];
  const float* scale_dev[kLossLevels]; float* d_scale_raw[kLossLevels];
  LossLevel lv[kLossLevels];
  int nbx[kLossLevels];
};

template <typename T, int PHASE>
__global__ void __launch_bounds__(256) fcos_loss_levels_kernel(LossLevels L, const float* __restrict__ gt,
                                                               const int* __restrict__ gt_count, int max_gt, int n_images,
                                                               float gamma, float alpha, float* __restrict__ sums, int gstride) {
  const T* cc = nullptr; const T* rg = nullptr; T* dcc = nullptr; T* drg = nullptr;
  const float* sd = nullptr; float* dsr = nullptr; LossLevel lv = L.lv[0]; int nbx = 0;
#pragma unroll
  for (int i = 0; i < kLossLevels; ++i)
    if (i == (int)blockIdx.z) {
      cc = (const T*)L.cc[i]; rg = (const T*)L.rg[i]; dcc = (T*)L.dcc[i]; drg = (T*)L.drg[i];
      sd = L.scale_dev[i]; dsr = L.d_scale_raw[i]; lv = L.lv[i]; nbx = L.nbx[i];
    }
  if ((int)blockIdx.x >= nbx) return;
  fcos_loss_body<T, PHASE>(cc, rg, gt, gt_count, max_gt, lv, n_images, gamma, alpha, sd, sums, dcc, drg, gstride, dsr, blockIdx.x, nbx);
}

// (raw, scales, gscales: optional — the gradient of the learnable per-level Scale, fcos.py:81: gscales[l] += raw[l] / scales[l], which was
// a torch division and a torch add on the main chain between the loss and the backward pass: two launches and their gaps in the one
// window of the step where nothing else runs.  Round 6.)
__global__ void fcos_loss_finalize_kernel(const float* __restrict__ sums, float* __restrict__ losses, int n_images,
                                          const float* __restrict__ raw, const float* __restrict__ scales, float* __restrict__ gscales,
                                          int n_levels) {
  if (raw != nullptr && (int)threadIdx.x < n_levels) gscales[threadIdx.x] += raw[threadIdx.x] / scales[threadIdx.x];
  if (threadIdx.x != 0) return;
  const float npos = sums[0], sw = sums[1];
  losses[0] = sums[2] / (npos + (float)n_images);
  losses[1] = npos > 0.f ? (sw > 0.f ? sums[3] / sw : sums[3] / npos) : 0.f;
  losses[2] = npos > 0.f ? sums[4] / npos : 0.f;
  losses[3] = npos;
}

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)

// phase 0: accumulate `sums[5]` (caller zeroes them once before the first level); phase 1: gradients.
extern "C" int osd_fcos_loss_level(int phase, const void* cls_ctr, const void* reg, const float* gt_boxes,
                                   const int32_t* gt_count, int max_gt, int n, int h, int w, int stride, float size_lo,
                                   float size_hi, float radius, float gamma, float alpha, const float* scale_dev,
                                   float* sums, void* d_cls_ctr, void* d_reg, int grad_stride, float* d_scale_raw,
                                   int dtype, void* stream) {
  if (!cls_ctr || !reg || !gt_boxes || !gt_count || !sums) return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss: null argument");
  if (phase == 1 && (!d_cls_ctr || !d_reg || !d_scale_raw || grad_stride < 4))
    return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss: bad gradient output");
  if (n == 0 || h * w == 0) return OSD_OK;
  LossLevel lv;
  lv.h = h; lv.w = w; lv.stride = stride; lv.lo = size_lo; lv.hi = size_hi; lv.radius_px = stride * radius;
  int bx = cdiv(h * w, 256);
  if (bx > 256) bx = 256;
  dim3 grid(bx, n);
#define OSD_LOSS_LAUNCH(TT, PH)                                                                                           \
  hipLaunchKernelGGL((fcos_loss_kernel<TT, PH>), grid, dim3(256), 0, OSD_STREAM(stream), (const TT*)cls_ctr, (const TT*)reg, \
                     gt_boxes, gt_count, max_gt, lv, n, gamma, alpha, scale_dev, sums, (TT*)d_cls_ctr, (TT*)d_reg, grad_stride, \
                     d_scale_raw)
  if (dtype == OSD_F32) { if (phase == 0) OSD_LOSS_LAUNCH(float, 0); else OSD_LOSS_LAUNCH(float, 1); }
  else if (dtype == OSD_BF16) { if (phase == 0) OSD_LOSS_LAUNCH(__bf16, 0); else OSD_LOSS_LAUNCH(__bf16, 1); }
  else return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss: bad dtype");
#undef OSD_LOSS_LAUNCH
  return osd_check_launch("fcos_loss_level");
}

// losses[4] = {loss_cls, loss_reg, loss_centerness, num_pos} from the accumulated sums
extern "C" int osd_fcos_loss_finalize(const float* sums, float* losses, int n, void* stream) {
  if (!sums || !losses) return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss_finalize: null argument");
  hipLaunchKernelGGL(fcos_loss_finalize_kernel, dim3(1), dim3(1), 0, OSD_STREAM(stream), sums, losses, n, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr, 0);
  return osd_check_launch("fcos_loss_finalize");
}

extern "C" int osd_fcos_loss_finalize_scales(const float* sums, float* losses, int n, const float* d_scale_raw, const float* scales,
                                             float* d_scales, int n_levels, void* stream) {
  if (!sums || !losses || !d_scale_raw || !scales || !d_scales || n_levels < 1 || n_levels > 64)
    return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss_finalize_scales: bad arguments");
  hipLaunchKernelGGL(fcos_loss_finalize_kernel, dim3(1), dim3(64), 0, OSD_STREAM(stream), sums, losses, n, d_scale_raw, scales, d_scales,
                     n_levels);
  return osd_check_launch("fcos_loss_finalize");
}

// every FPN level in one launch per phase (phase 0: sums; phase 1: gradients).  Host arrays of n_levels entries.
extern "C" int osd_fcos_loss_levels(int phase, int n_levels, const void* const* cls_ctrs, const void* const* regs,
                                    const float* gt_boxes, const int32_t* gt_count, int max_gt, int n, const int32_t* hs,
                                    const int32_t* ws, const int32_t* strides, const float* size_lo, const float* size_hi,
                                    float radius, float gamma, float alpha, const float* const* scale_devs, float* sums,
                                    void* const* d_cls_ctrs, void* const* d_regs, int grad_stride,
                                    float* const* d_scale_raws, int dtype, void* stream) {
  if (!cls_ctrs || !regs || !gt_boxes || !gt_count || !sums || !hs || !ws || !strides || !size_lo || !size_hi ||
      n_levels < 1 || n_levels > kLossLevels)
    return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss_levels: bad arguments");
  if (phase == 1 && (!d_cls_ctrs || !d_regs || !d_scale_raws || !scale_devs || grad_stride < 4))
    return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss_levels: bad gradient output");
  if (n == 0) return OSD_OK;
  LossLevels L;
  int bmax = 1;
  for (int i = 0; i < kLossLevels; ++i) {
    const int j = i < n_levels ? i : 0;
    if (!cls_ctrs[j] || !regs[j]) return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss_levels: null level %d", j);
    L.cc[i] = cls_ctrs[j]; L.rg[i] = regs[j];
    L.dcc[i] = phase == 1 ? d_cls_ctrs[j] : nullptr; L.drg[i] = phase == 1 ? d_regs[j] : nullptr;
    L.scale_dev[i] = scale_devs ? scale_devs[j] : nullptr; L.d_scale_raw[i] = phase == 1 ? d_scale_raws[j] : nullptr;
    L.lv[i].h = hs[j]; L.lv[i].w = ws[j]; L.lv[i].stride = strides[j]; L.lv[i].lo = size_lo[j]; L.lv[i].hi = size_hi[j];
    L.lv[i].radius_px = strides[j] * radius;
    int bx = cdiv(hs[j] * ws[j], 256);
    if (bx > 256) bx = 256;
    L.nbx[i] = i < n_levels ? bx : 0;
    if (i < n_levels && bx > bmax) bmax = bx;
  }
  dim3 grid(bmax, n, n_levels);
#define OSD_LOSSL_LAUNCH(TT, PH)                                                                                       \
  hipLaunchKernelGGL((fcos_loss_levels_kernel<TT, PH>), grid, dim3(256), 0, OSD_STREAM(stream), L, gt_boxes, gt_count, max_gt, n, \
                     gamma, alpha, sums, grad_stride)
  if (dtype == OSD_F32) { if (phase == 0) OSD_LOSSL_LAUNCH(float, 0); else OSD_LOSSL_LAUNCH(float, 1); }
  else if (dtype == OSD_BF16) { if (phase == 0) OSD_LOSSL_LAUNCH(__bf16, 0); else OSD_LOSSL_LAUNCH(__bf16, 1); }
  else return osd_fail(OSD_ERR_INVALID_ARG, "fcos_loss_levels: bad dtype");
#undef OSD_LOSSL_LAUNCH
  return osd_check_launch("fcos_loss_levels");
}
