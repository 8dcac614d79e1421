// PASCAL VOC detection matching on the device (evaluation half): for every detection the ground-truth box it
// is matched to and whether it counts as a true positive, a false positive or is ignored — the per-image body of
// calc_detection_voc_prec_rec (maskrcnn_benchmark/data/datasets/evaluation/voc/voc_eval.py:84-137).  The precision / recall
// curves and the AP integrals over the whole dataset are a few hundred scalars and stay on the host (oneshotdet_amd/evaluation.py).
//
// One workgroup per image.  The reference walks a class's detections in descending score order and lets the FIRST one matched
// to a ground-truth box claim it; equivalently, box g belongs to the matched detection with the highest (score, index) key, which
// is an atomic max — no ordering pass.  IoU follows the reference's float32 operation order ("+1" on x2, y2 of both boxes,
// voc_eval.py:111-114, then boxlist_iou's own "+1" areas, structures/boxlist_ops.py:221-256) with contraction off, so a
// comparison against the threshold falls the same way.
#include "osd_common.h"

namespace {

constexpr int kVocMaxGt = 512;

__device__ __forceinline__ unsigned voc_order_bits(float s) {      // order-preserving float -> uint
  const unsigned u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256) voc_match_kernel(const float* __restrict__ det_boxes, const float* __restrict__ det_scores,
                                                        const int32_t* __restrict__ det_labels, const int32_t* __restrict__ det_count,
                                                        const float* __restrict__ gt_boxes, const int32_t* __restrict__ gt_labels,
                                                        const uint8_t* __restrict__ gt_difficult, const int32_t* __restrict__ gt_count,
                                                        int max_det, int max_gt, float iou_thresh, int8_t* __restrict__ match,
                                                        int32_t* __restrict__ matched_gt) {
#pragma clang fp contract(off)
  __shared__ float gb[kVocMaxGt][4];
  __shared__ float garea[kVocMaxGt];
  __shared__ int glab[kVocMaxGt];
  __shared__ unsigned long long owner[kVocMaxGt];
  const int img = blockIdx.x, t = threadIdx.x;
  const int nd = min(det_count[img], max_det), ng = min(gt_count[img], max_gt);
  for (int g = t; g < ng; g += 256) {
    const float* b = gt_boxes + ((size_t)img * max_gt + g) * 4;
    const float x1 = b[0], y1 = b[1], x2 = b[2] + 1.f, y2 = b[3] + 1.f;      // voc_eval.py:113-114
    gb[g][0] = x1; gb[g][1] = y1; gb[g][2] = x2; gb[g][3] = y2;
    garea[g] = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);                           // bounding_box.py:226-231 (TO_REMOVE = 1)
    glab[g] = gt_labels[(size_t)img * max_gt + g];
    owner[g] = 0ull;
  }
  __syncthreads();
  const float* db = det_boxes + (size_t)img * max_det * 4;
  const float* ds = det_scores + (size_t)img * max_det;
  const int32_t* dl = det_labels + (size_t)img * max_det;
  int32_t* mg = matched_gt + (size_t)img * max_det;
  // pass 1: every detection's ground-truth box (first maximum of the IoU over the boxes of its class; -1 below the threshold)
  for (int d = t; d < nd; d += 256) {
    const float x1 = db[d * 4 + 0], y1 = db[d * 4 + 1], x2 = db[d * 4 + 2] + 1.f, y2 = db[d * 4 + 3] + 1.f;    // :111-112
    const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
    const int l = dl[d];
    int best = -1;
    float best_iou = -1.f;
    for (int g = 0; g < ng; ++g) {
      if (glab[g] != l) continue;
      const float ltx = fmaxf(x1, gb[g][0]), lty = fmaxf(y1, gb[g][1]);
      const float rbx = fminf(x2, gb[g][2]), rby = fminf(y2, gb[g][3]);
      const float w = fmaxf(rbx - ltx + 1.f, 0.f), h = fmaxf(rby - lty + 1.f, 0.f);
      const float inter = w * h;
      const float iou = inter / (area + garea[g] - inter);
      if (iou > best_iou) { best_iou = iou; best = g; }                     // numpy argmax: the first maximum
    }
    if (best >= 0 && best_iou < iou_thresh) best = -1;                       // :120
    mg[d] = best;
    if (best >= 0)
      atomicMax(&owner[best], ((unsigned long long)voc_order_bits(ds[d]) << 32) | (unsigned)d);
  }
  __syncthreads();
  // pass 2: 1 = the first (highest-scoring) detection of a non-difficult box, 0 = unmatched or a later one, -1 = difficult box
  for (int d = t; d < max_det; d += 256) {
    int8_t m = 0;
    if (d < nd) {
      const int g = mg[d];
      if (g >= 0) {
        if (gt_difficult[(size_t)img * max_gt + g]) m = -1;
        else m = owner[g] == (((unsigned long long)voc_order_bits(ds[d]) << 32) | (unsigned)d) ? 1 : 0;
      }
    }
    match[(size_t)img * max_det + d] = m;
  }
}


// ---- COCO-style matching (the reference's COCO configs evaluate through pycocotools' COCOeval, coco_eval.py:385-408; the
// package is not in the image: this follows its published algorithm, cocoeval.py evaluateImg + maskApi.c bbIou — see
// oracle/coco_eval_ref.py, "parity unpinned") ----
// One workgroup per (image, category) pair; thread (a, t) = (area range, IoU threshold) walks the pair's detections in descending
// score order (the host sorted and cut them at the largest maxDets) and, for each, the ground-truth boxes with the non-ignored
// ones first: the box with the highest IoU >= t that is still free (a crowd box is never used up), preferring non-ignored boxes.
// Boxes are [x, y, w, h] doubles; IoU without '+1', against a crowd box the union is the detection's area.
constexpr int kCocoMaxGt = 512, kCocoMaxT = 10, kCocoMaxA = 4;
struct CocoParams {
  double iou_thr[kCocoMaxT];
  double area_lo[kCocoMaxA], area_hi[kCocoMaxA];
  int n_thr, n_area;
};

__global__ void __launch_bounds__(64) coco_match_kernel(const double* __restrict__ det_boxes, const int32_t* __restrict__ det_count,
                                                        const double* __restrict__ gt_boxes, const double* __restrict__ gt_area,
                                                        const uint8_t* __restrict__ gt_crowd, const int32_t* __restrict__ gt_count,
                                                        int max_det, int max_gt, CocoParams cp, int32_t* __restrict__ dt_match,
                                                        uint8_t* __restrict__ dt_ignore, uint8_t* __restrict__ gt_ignore) {
  __shared__ double gb[kCocoMaxGt][4];
  __shared__ double ga[kCocoMaxGt];
  __shared__ uint8_t gcrowd[kCocoMaxGt];
  __shared__ int16_t order[kCocoMaxA][kCocoMaxGt];      // per area range: ground-truth indices, non-ignored first (stable)
  __shared__ uint8_t gig[kCocoMaxA][kCocoMaxGt];        // ... and the ignore flag of the box at that sorted position
  __shared__ uint8_t taken[kCocoMaxA * kCocoMaxT][kCocoMaxGt];
  const int pr = blockIdx.x, t = threadIdx.x;
  const int nd = min(det_count[pr], max_det), ng = min(gt_count[pr], max_gt);
  for (int g = t; g < ng; g += 64) {
    const double* b = gt_boxes + ((size_t)pr * max_gt + g) * 4;
    gb[g][0] = b[0]; gb[g][1] = b[1]; gb[g][2] = b[2]; gb[g][3] = b[3];
    ga[g] = gt_area[(size_t)pr * max_gt + g];
    gcrowd[g] = gt_crowd[(size_t)pr * max_gt + g];
  }
  __syncthreads();
  if (t < cp.n_area) {                                  // stable partition of the boxes by their ignore flag for this range
    int n = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int g = 0; g < ng; ++g) {
        const bool ig = gcrowd[g] || ga[g] < cp.area_lo[t] || ga[g] > cp.area_hi[t];
        if ((int)ig == pass) { order[t][n] = (int16_t)g; gig[t][n] = (uint8_t)ig; ++n; }
      }
    for (int n2 = 0; n2 < ng; ++n2) gt_ignore[((size_t)pr * cp.n_area + t) * max_gt + n2] = gig[t][n2];
    for (int n2 = ng; n2 < max_gt; ++n2) gt_ignore[((size_t)pr * cp.n_area + t) * max_gt + n2] = 0;
  }
  for (int i = t; i < cp.n_area * cp.n_thr * kCocoMaxGt; i += 64) (&taken[0][0])[i] = 0;
  __syncthreads();
  if (t >= cp.n_area * cp.n_thr) return;
  const int a = t / cp.n_thr, ti = t - a * cp.n_thr;
  const double thr = cp.iou_thr[ti];
  int32_t* dm = dt_match + (((size_t)pr * cp.n_area + a) * cp.n_thr + ti) * max_det;
  uint8_t* di = dt_ignore + (((size_t)pr * cp.n_area + a) * cp.n_thr + ti) * max_det;
  for (int d = 0; d < max_det; ++d) {
    int32_t match = 0;
    uint8_t ign = 0;
    if (d < nd) {
      const double* b = det_boxes + ((size_t)pr * max_det + d) * 4;
      const double dx = b[0], dy = b[1], dw = b[2], dh = b[3];
      const double darea = dw * dh;
      double iou = fmin(thr, 1.0 - 1e-10);
      int m = -1;
      for (int n = 0; n < ng; ++n) {
        const int g = order[a][n];
        if (taken[t][n] && !gcrowd[g]) continue;
        if (m > -1 && !gig[a][m] && gig[a][n]) break;
        double o = 0.0;
        const double w = fmin(dw + dx, gb[g][2] + gb[g][0]) - fmax(dx, gb[g][0]);
        if (w > 0) {
          const double h = fmin(dh + dy, gb[g][3] + gb[g][1]) - fmax(dy, gb[g][1]);
          if (h > 0) {
            const double inter = w * h;
            o = inter / (gcrowd[g] ? darea : darea + gb[g][2] * gb[g][3] - inter);
          }
        }
        if (o < iou) continue;
        iou = o;
        m = n;
      }
      if (m > -1) {
        ign = gig[a][m];
        match = order[a][m] + 1;                          // the matched box, as its index in the INPUT order + 1
        taken[t][m] = 1;
      } else {
        ign = (darea < cp.area_lo[a] || darea > cp.area_hi[a]) ? 1 : 0;
      }
    }
    dm[d] = match;
    di[d] = ign;
  }
}

// ---- precision / recall curves and average precision, all classes in one launch each (voc_eval.py:139-216) ----
// Input: the dataset's match flags sorted ONCE by (class ascending, score descending) — class c owns [class_begin[c],
// class_begin[c + 1]).  One workgroup per class walks its range in 256-element chunks with a carry.
constexpr int kApThreads = 256;

// inclusive scan of one value per thread over the workgroup (Hillis-Steele in LDS); OP(a, b) associative
template <typename V, typename OP>
__device__ __forceinline__ V ap_block_scan(V v, V* sh, OP op, bool reverse) {
  const int t = threadIdx.x;
  const int pos = reverse ? kApThreads - 1 - t : t;
  sh[pos] = v;
  __syncthreads();
  for (int off = 1; off < kApThreads; off <<= 1) {
    V o = sh[pos];
    if (pos >= off) o = op(sh[pos - off], o);
    __syncthreads();
    sh[pos] = o;
    __syncthreads();
  }
  const V r = sh[pos];
  __syncthreads();
  return r;
}

// prec[i] = tp_i / (tp_i + fp_i) (NaN while both are 0: only ignored detections so far), rec[i] = tp_i / n_pos (NaN when the
// class has no countable ground truth: the host then reports no recall curve), with tp_i / fp_i = number of flags 1 / 0 among
// the class's first i + 1 detections.  Integer counts, IEEE double division: the values numpy computes.
__global__ void __launch_bounds__(kApThreads) voc_curves_kernel(const int8_t* __restrict__ flags, const int32_t* __restrict__ class_begin,
                                                                const int32_t* __restrict__ n_pos, double* __restrict__ prec,
                                                                double* __restrict__ rec) {
  __shared__ int2 sh[kApThreads];
  const int c = blockIdx.x, t = threadIdx.x;
  const int b = class_begin[c], e = class_begin[c + 1];
  const double np = (double)n_pos[c];
  int2 carry = make_int2(0, 0);
  auto add2 = [](int2 a, int2 b2) { return make_int2(a.x + b2.x, a.y + b2.y); };
  for (int base = b; base < e; base += kApThreads) {
    const int i = base + t;
    const int f = i < e ? (int)flags[i] : -1;
    int2 v = ap_block_scan(make_int2(f == 1, f == 0), sh, add2, false);
    v = add2(v, carry);
    if (i < e) {
      prec[i] = (double)v.x / (double)(v.x + v.y);
      rec[i] = n_pos[c] > 0 ? (double)v.x / np : __longlong_as_double(0x7ff8000000000000ll);
    }
    __shared__ int2 last;
    if (t == kApThreads - 1) last = v;
    __syncthreads();
    carry = last;
    __syncthreads();
  }
}

// AP of every class from its curves.  With P_i = max over j >= i of prec_j (NaN read as 0) — the monotone precision envelope,
// a suffix maximum — both metrics are sums over the points where recall steps:
//   area metric:      sum over i with rec_i != rec_{i-1} (rec_{-1} = 0) of (rec_i - rec_{i-1}) * P_i
//   VOC 2007 metric:  (1 / 11) sum over t in {0, 0.1, .., 1} of P_{first i with rec_i >= t} (0 when recall never reaches t)
// has_rec[c] == 0 (class without countable ground truth) or has_prec[c] == 0 (class id never seen): NaN.
__global__ void __launch_bounds__(kApThreads) voc_ap_kernel(const double* __restrict__ prec, const double* __restrict__ rec,
                                                            const int32_t* __restrict__ class_begin, const uint8_t* __restrict__ has_prec,
                                                            const uint8_t* __restrict__ has_rec, int use_07, double* __restrict__ ap) {
  __shared__ double sh[kApThreads];
  __shared__ double p11[11];
  __shared__ double carry_sh;
  const int c = blockIdx.x, t = threadIdx.x;
  if (!has_prec[c] || !has_rec[c]) {
    if (t == 0) ap[c] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  const int b = class_begin[c], e = class_begin[c + 1];
  if (t < 11) p11[t] = 0.0;
  if (t == 0) carry_sh = 0.0;
  __syncthreads();
  double area = 0.0;
  auto dmax = [](double a, double b2) { return a > b2 ? a : b2; };
  const int nchunk = (e - b + kApThreads - 1) / kApThreads;
  for (int ch = nchunk - 1; ch >= 0; --ch) {            // from the class's last detections backwards: the suffix maximum
    const int i = b + ch * kApThreads + t;
    double p = 0.0;
    if (i < e) { p = prec[i]; if (!(p == p)) p = 0.0; }
    double env = ap_block_scan(p, sh, dmax, true);
    env = dmax(env, carry_sh);
    __syncthreads();
    if (i < e) {
      const double r = rec[i], rp = i > b ? rec[i - 1] : 0.0;
      if (r != rp) area += (r - rp) * env;
      if (use_07) {
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const double thr = (double)k * 0.1;          // numpy.arange(0., 1.1, 0.1)[k]
          if (r >= thr && (i == b || !(rec[i - 1] >= thr))) p11[k] = env;
        }
      }
    }
    if (t == 0) carry_sh = env;                         // thread 0 holds the chunk's first element = the max of everything after
    __syncthreads();
  }
  if (use_07) {
    if (t == 0) {
      double a = 0.0;
      for (int k = 0; k < 11; ++k) a += p11[k] / 11.0;  // voc_eval.py:185-192 adds p / 11 in this order
      ap[c] = a;
    }
    return;
  }
  sh[t] = area;
  __syncthreads();
  for (int off = kApThreads / 2; off > 0; off >>= 1) {  // fixed-order tree: the same bits on every run
    if (t < off) sh[t] += sh[t + off];
    __syncthreads();
  }
  if (t == 0) ap[c] = sh[0];
}

}  // namespace

extern "C" int osd_voc_match(const float* det_boxes, const float* det_scores, const int32_t* det_labels, const int32_t* det_count,
                             const float* gt_boxes, const int32_t* gt_labels, const uint8_t* gt_difficult, const int32_t* gt_count,
                             int n, int max_det, int max_gt, float iou_thresh, int8_t* match, int32_t* matched_gt, void* stream) {
  if (n < 0 || max_det < 0 || max_gt < 0) return osd_fail(OSD_ERR_INVALID_ARG, "voc_match: negative size");
  if (n == 0 || max_det == 0) return OSD_OK;
  if (!det_boxes || !det_scores || !det_labels || !det_count || !gt_count || !match || !matched_gt ||
      (max_gt > 0 && (!gt_boxes || !gt_labels || !gt_difficult)))
    return osd_fail(OSD_ERR_INVALID_ARG, "voc_match: null argument");
  if (max_gt > kVocMaxGt) return osd_fail(OSD_ERR_UNSUPPORTED, "voc_match: at most %d ground-truth boxes per image (got %d)", kVocMaxGt, max_gt);
  hipLaunchKernelGGL(voc_match_kernel, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), det_boxes, det_scores, det_labels,
                     det_count, gt_boxes, gt_labels, gt_difficult, gt_count, max_det, max_gt, iou_thresh, match, matched_gt);
  return osd_check_launch("voc_match");
}

extern "C" int osd_voc_curves(const int8_t* flags_sorted, const int32_t* class_begin, const int32_t* n_pos, int n_classes,
                              double* prec, double* rec, void* stream) {
  if (n_classes < 0) return osd_fail(OSD_ERR_INVALID_ARG, "voc_curves: negative class count");
  if (n_classes == 0) return OSD_OK;
  if (!class_begin || !n_pos || !prec || !rec) return osd_fail(OSD_ERR_INVALID_ARG, "voc_curves: null argument");
  hipLaunchKernelGGL(voc_curves_kernel, dim3(n_classes), dim3(kApThreads), 0, reinterpret_cast<hipStream_t>(stream), flags_sorted,
                     class_begin, n_pos, prec, rec);
  return osd_check_launch("voc_curves");
}

extern "C" int osd_voc_ap(const double* prec, const double* rec, const int32_t* class_begin, const uint8_t* has_prec,
                          const uint8_t* has_rec, int n_classes, int use_07_metric, double* ap, void* stream) {
  if (n_classes < 0) return osd_fail(OSD_ERR_INVALID_ARG, "voc_ap: negative class count");
  if (n_classes == 0) return OSD_OK;
  if (!class_begin || !has_prec || !has_rec || !ap) return osd_fail(OSD_ERR_INVALID_ARG, "voc_ap: null argument");
  hipLaunchKernelGGL(voc_ap_kernel, dim3(n_classes), dim3(kApThreads), 0, reinterpret_cast<hipStream_t>(stream), prec, rec, class_begin,
                     has_prec, has_rec, use_07_metric, ap);
  return osd_check_launch("voc_ap");
}

extern "C" int osd_coco_match(const double* det_boxes_xywh, const int32_t* det_count, const double* gt_boxes_xywh, const double* gt_area,
                              const uint8_t* gt_crowd, const int32_t* gt_count, int n_pairs, int max_det, int max_gt,
                              const double* iou_thrs, int n_thrs, const double* area_ranges, int n_areas, int32_t* dt_match,
                              uint8_t* dt_ignore, uint8_t* gt_ignore, void* stream) {
  if (n_pairs < 0 || max_det < 0 || max_gt < 0) return osd_fail(OSD_ERR_INVALID_ARG, "coco_match: negative size");
  if (n_thrs < 1 || n_thrs > kCocoMaxT || n_areas < 1 || n_areas > kCocoMaxA || !iou_thrs || !area_ranges)
    return osd_fail(OSD_ERR_INVALID_ARG, "coco_match: 1..%d IoU thresholds and 1..%d area ranges (host arrays)", kCocoMaxT, kCocoMaxA);
  if (n_pairs == 0) return OSD_OK;
  if (!det_count || !gt_count || !dt_match || !dt_ignore || !gt_ignore || (max_det > 0 && !det_boxes_xywh) ||
      (max_gt > 0 && (!gt_boxes_xywh || !gt_area || !gt_crowd)))
    return osd_fail(OSD_ERR_INVALID_ARG, "coco_match: null argument");
  if (max_gt > kCocoMaxGt) return osd_fail(OSD_ERR_UNSUPPORTED, "coco_match: at most %d ground-truth boxes per (image, category) (got %d)", kCocoMaxGt, max_gt);
  CocoParams cp;
  cp.n_thr = n_thrs; cp.n_area = n_areas;
  for (int i = 0; i < kCocoMaxT; ++i) cp.iou_thr[i] = i < n_thrs ? iou_thrs[i] : 2.0;
  for (int i = 0; i < kCocoMaxA; ++i) { cp.area_lo[i] = i < n_areas ? area_ranges[2 * i] : 0.0; cp.area_hi[i] = i < n_areas ? area_ranges[2 * i + 1] : 0.0; }
  hipLaunchKernelGGL(coco_match_kernel, dim3(n_pairs), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), det_boxes_xywh, det_count,
                     gt_boxes_xywh, gt_area, gt_crowd, gt_count, max_det, max_gt, cp, dt_match, dt_ignore, gt_ignore);
  return osd_check_launch("coco_match");
}
