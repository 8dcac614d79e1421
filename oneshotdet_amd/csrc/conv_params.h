// Kernel-side parameter block shared by the two implicit-GEMM convolution implementations.
#pragma once
#include <hip/hip_runtime.h>

// one (input, output) pair of a grouped launch: same conv geometry (channels, taps, stride, pad), its own tensors, batch and
// spatial size, and its own weights / bias pointer (the FPN levels of a tower conv repeat one pointer; the two towers, or
// the target and the query backbone, bring their own)
constexpr int kConvMaxSeg = 12;

// GroupNorm-backward statistics gathered by the epilogue of the data-gradient conv that PRODUCES dt, the gradient w.r.t. the
// GroupNorm + ReLU output (conv_igemm_sp.hip only; u == nullptr: off for this segment).  With z = a u + b, dz = z > 0 ? dt : 0 and
// xhat = xa u + xb, per output pixel and channel, the epilogue adds (fp32 atomics) the sums the backward apply kernel
// (backward.hip: gnl_bwd_apply_kernel) reads — so that the separate pass over (u, dt) of gnl_bwd_stats_kernel is not needed
constexpr int kGnSlabs = 64;   // partial sums per (level, image): spreads the atomics; = OSD_GN_SPLITS of the C ABI
struct ConvGnb {
  const void* u;            // the GroupNorm's input at this conv's OUTPUT pixels: [M][out_stride], the conv's dtype
  const float* ab;          // [4][n][Cout]: a, b, xa, xb of this level (osd_groupnorm_relu_fwd_levels' ab block)
  const float* gamma;       // [Cout]
  float* ws;                // [n][kGnSlabs][groups][2]: sum dz * gamma, sum dz * gamma * xhat
  float* pw;                // [n][kGnSlabs][2][Cout]: sum dz * xhat (d gamma), sum dz (d beta)
};

struct ConvSeg {
  const void* x;
  void* y;
  const void* res;
  const void* mask;
  const float* act_scale_dev;
  const void* w;
  const float* bias;
  int H, W, Ho, Wo, M, sN, sH;
  int tile_begin;     // first pixel tile of this segment (filled by the launcher: depends on the tile height)
  ConvGnb gn;
};

struct ConvKParams {
  const void* x;
  const void* w;
  const float* bias;
  const void* res;
  const void* mask;   // optional: y = (mask[m][c] > 0) ? y : 0   (ReLU backward fused into a dgrad epilogue)
  void* y;
  int H, W, Cin, sN, sH, sW;
  int Ho, Wo, Cout, HoWo;
  int R, S, sh, sw, ph, pw;
  int w_rows, Ktot, out_stride;
  int res_mode, res_h, res_w, res_stride;
  int act;
  float act_scale;
  const float* act_scale_dev;   // optional device scalar overriding act_scale (the learnable Scale of fcos.py:81)
  int relu_in;
  int M, tilesM, tilesN, KT;
  // second pixel source of a 1x1 conv (x2 != nullptr): input channels [cin1, Cin) come from x2, read at output pixel
  // (ho, wo) -> x2[n][ho * st2][wo * st2] (dense NHWC, Cin - cin1 channels): conv3 + downsample of a bottleneck's first
  // block as ONE GEMM over the concatenated K (conv_dma kernels only, single problem)
  const void* x2;
  const void* w2;     // optional: the second part's own packed weights [w_rows][Cin - cin1] (w then holds [w_rows][cin1], Ktot =
                      // cin1); nullptr: w holds both parts side by side (Ktot = Cin)
  int cin1, x2_sN, x2_sH, x2_sW, st2;
  int gn_n, gn_groups;   // ConvSeg.gn: images per level and GroupNorm groups (0: no segment asks for the statistics)
  int n_seg;          // 0 = single problem (fields above); > 0: seg[] overrides x/y/res/mask/act_scale_dev/H/W/Ho/Wo/M/sN/sH
  ConvSeg seg[kConvMaxSeg];
};

// conv_igemm_dma.hip
int osd_conv_dma_dispatch(int dtype, int tile, int variant, const ConvKParams& p, hipStream_t s);
// conv_igemm_dma.hip: the 64 x 64 tile with a ring of 5 / 8 stages (algos 58 / 59: cold weight streams of the latency-sized launches)
int osd_conv_dma_deep(int dtype, int nst, const ConvKParams& p, hipStream_t s);
// conv_igemm_sp.hip: bf16, 3x3 / stride 1 / pad 1, pixel rows fetched once per filter row, software-pipelined operand
// fragments and a mid-stage barrier (tile id 6, variant 1:
// any map width — the padded-image form where every width is 64 / 128 / 256, else the consecutive-rows form; variant 2 =
// general_width: the consecutive-rows form on every width, for tests and A/B timing; variant 3 = half_tile: 128-pixel tiles)
int osd_conv_sp_launch(const ConvKParams& p, hipStream_t s, bool general_width = false, bool half_tile = false, bool small_tile = false);
// conv_px.hip: bf16, plain 1x1 / stride 1 conv with cin 64 / 128 / 256: every wave keeps its 32 pixels' K values in registers and
// the workgroups stream only weight chunks (algo ids 49 / 50: eight waves of 16 pixels / four of 32; the expanding bottleneck convs
// and conv1's data gradient)
int osd_conv_px_launch(const ConvKParams& p, hipStream_t s, bool wide_waves);
// conv_pred.hip (round 6, algo 51): bf16 3x3 / stride 1 / pad 1 convs with <= 4 output channels (the FCOS prediction convs' forward): an
// 8 x 32 output patch per workgroup, the 10 x 34 input patch staged once per 64-channel slab and read by all nine taps
int osd_conv_pred_launch(const ConvKParams& p, hipStream_t s);
