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
