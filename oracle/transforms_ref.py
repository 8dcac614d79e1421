"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's input transforms
(transforms only), numpy, bit-exact.

Reference (paths relative to /root/reference/maskrcnn_benchmark/):
  data/transforms/transforms.py:27-63   Resize.get_size / __call__      (min / max size, aspect ratio kept)
  data/transforms/transforms.py:66-75   RandomHorizontalFlip
  data/transforms/transforms.py:78-92   ToTensor, Normalize (to_bgr255: image[[2, 1, 0]] * 255, then (x - mean) / std)
  data/transforms/build.py:5-52         Compose order Resize -> flip -> ToTensor -> Normalize; target and support sizes
  structures/bounding_box.py:91-128     BoxList.resize (boxes follow the image)
  structures/bounding_box.py:130-166    BoxList.transpose(FLIP_LEFT_RIGHT) (TO_REMOVE = 1)
  structures/image_list.py:52-70        zero padding of the batch to a multiple of SIZE_DIVISIBILITY

Third-party arithmetic NOT under /root/reference: `F.resize` is torchvision==0.2.1 (INSTALL.md:5), which calls
`PIL.Image.resize((w, h), BILINEAR)`; the resampler is Pillow's `ImagingResample` (src/libImaging/Resample.c; Pillow is
unpinned in the reference, 12.2.0 in this image).  Its published algorithm for 8-bit images, restated in
`pil_bilinear_resize`:
  * per axis, out pixel xx: scale = in / out; filterscale = max(scale, 1); support = filterscale (bilinear support 1);
    center = (xx + 0.5) * scale; taps xmin = int(center - support + 0.5) clamped to >= 0 .. xmax = int(center + support +
    0.5) clamped to <= in; weight w(x) = max(0, 1 - |(x + xmin - center + 0.5) / filterscale|), normalised by their
    sequential double sum (`precompute_coeffs`);
  * weights -> 22-bit fixed point, kk = int(+-0.5 + w * 2^22) (`normalize_coeffs_8bpc`);
  * out = clip8((2^21 + sum pixel * kk) >> 22); the HORIZONTAL pass runs first and its uint8 result feeds the vertical
    pass (`ImagingResampleInner`); an axis whose size does not change is skipped.
PINNED: against Pillow itself on random images (tests/test_oracle_golden.py::test_pil_resize_restatement, bit-exact) and
against fixtures recorded through the reference's own transforms.py / build.py / BoxList in tests/golden/transforms.npz
(tests/golden/make_golden.py::gen_transforms; torchvision's four thin functional wrappers over PIL / torch are provided by
the harness as written in torchvision 0.2.1, everything else is the reference's code executing unmodified).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
PIXEL_MEAN = (102.9801, 115.9465, 122.7717)      # config/defaults.py:64
PIXEL_STD = (1.0, 1.0, 1.0)                       # config/defaults.py:66


def get_size(image_size_wh, min_size, max_size):
    """transforms.py:35-57 (`random.choice(self.min_size)` with the config's single-element tuple). -> (oh, ow)."""
    w, h = image_size_wh
    size = min_size
    if max_size is not None:
        min_original_size = float(min((w, h)))
        max_original_size = float(max((w, h)))
        if max_original_size / min_original_size * size > max_size:
            size = int(round(max_size * min_original_size / max_original_size))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        ow = size
        oh = int(size * h / w)
    else:
        oh = size
        ow = int(size * w / h)
    return (oh, ow)


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter -> (kk [out, ksize] int64,
    bounds [out, 2] = (first tap, tap count))."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            v = w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _resample_axis(img, out_size, axis):
    kk, b = resample_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.zeros((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        xmin, xmax = b[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * kk[xx, x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def pil_bilinear_resize(img_hwc_u8, oh, ow):
    """PIL.Image.resize((ow, oh), BILINEAR) on an RGB uint8 image [H, W, 3]."""
    h, w = img_hwc_u8.shape[:2]
    t = img_hwc_u8
    if ow != w:
        t = _resample_axis(t, ow, 1)
    if oh != h:
        t = _resample_axis(t, oh, 0)
    return t


def to_tensor_normalize(img_hwc_u8, mean=PIXEL_MEAN, std=PIXEL_STD, to_bgr255=True):
    """ToTensor (uint8 HWC -> float32 CHW / 255) then Normalize (transforms.py:82-92), float32 operation by operation."""
    t = img_hwc_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    if to_bgr255:
        t = t[[2, 1, 0]] * np.float32(255.0)
    out = np.empty_like(t)
    for c in range(3):
        out[c] = (t[c] - np.float32(mean[c])) / np.float32(std[c])
    return out


def transform_image(img_hwc_u8, min_size, max_size, flip=False, mean=PIXEL_MEAN, std=PIXEL_STD, to_bgr255=True):
    """The Compose of build.py:39-46 on one image -> float32 [3, oh, ow]."""
    h, w = img_hwc_u8.shape[:2]
    oh, ow = get_size((w, h), min_size, max_size)
    t = pil_bilinear_resize(img_hwc_u8, oh, ow)
    if flip:
        t = t[:, ::-1]
    return to_tensor_normalize(np.ascontiguousarray(t), mean, std, to_bgr255)


def transform_boxes(boxes_xyxy, size_wh, new_size_wh, flip=False):
    """BoxList.resize (bounding_box.py:91-128) then transpose(FLIP_LEFT_RIGHT) (:130-166), float32 like torch."""
    b = np.asarray(boxes_xyxy, np.float32).reshape(-1, 4)
    ratios = tuple(float(s) / float(so) for s, so in zip(new_size_wh, size_wh))
    if ratios[0] == ratios[1]:
        b = b * np.float32(ratios[0])
    else:
        rw, rh = np.float32(ratios[0]), np.float32(ratios[1])
        b = np.stack([b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh], 1)
    if flip:
        width = np.float32(new_size_wh[0])
        one = np.float32(1)
        b = np.stack([width - b[:, 2] - one, b[:, 1], width - b[:, 0] - one, b[:, 3]], 1)
    return b.astype(np.float32)


def batch_images(tensors_chw, size_divisible=32):
    """to_image_list on a list (image_list.py:52-70): zero padding bottom / right -> ([N,3,Hp,Wp], [(h, w)])."""
    mh = max(t.shape[1] for t in tensors_chw)
    mw = max(t.shape[2] for t in tensors_chw)
    if size_divisible > 0:
        mh = int(np.ceil(mh / size_divisible) * size_divisible)
        mw = int(np.ceil(mw / size_divisible) * size_divisible)
    out = np.zeros((len(tensors_chw), 3, mh, mw), np.float32)
    for i, t in enumerate(tensors_chw):
        out[i, :, :t.shape[1], :t.shape[2]] = t
    return out, [t.shape[1:] for t in tensors_chw]
