"""Input transforms with the reference's own interface, executed by ONE fused HIP kernel chain per image.

Reference: data/transforms/transforms.py (Compose :9, Resize :27-63, RandomHorizontalFlip :66-75, ToTensor :78-80,
Normalize :82-92) built by data/transforms/build.py:5-52, then BatchCollator / to_image_list
(data/collate_batch.py:15-20, structures/image_list.py:30-73).  There each step materialises an image (PIL resize on the
host, flip, float conversion, two elementwise passes, a padding copy).  Here the transform objects keep the reference's
names, constructor arguments and call signature `t(image, target) -> (image, target)`, but an image travelling through a
Compose is a `DeviceImage` — the uint8 source on the device plus the pending geometry — and the pixels are produced once,
by `osd_image_transform` (csrc/transforms.hip): bit-exact PIL bilinear resampling, flip, /255, BGR255 - mean, written either
as the float CHW tensor the reference returns (Normalize) or directly into the padded batch (`collate`), optionally already
in the stem conv's NHWC4 input format.  Boxes follow with BoxList.resize / transpose semantics (bounding_box.py:91-166).

There is no CPU path: a host uint8 array is uploaded, everything else happens on the device.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from . import _lib, ops

PIXEL_MEAN = (102.9801, 115.9465, 122.7717)     # config/defaults.py:64 (BGR)
PIXEL_STD = (1.0, 1.0, 1.0)                      # config/defaults.py:66
FLIP_LEFT_RIGHT = 0                              # PIL.Image.FLIP_LEFT_RIGHT, the constant BoxList.transpose is given


class DeviceImage(object):
    """An RGB uint8 image [H, W, 3] on the device with the geometry the transforms have requested so far.
    `.size` is (width, height) of the CURRENT logical image, like PIL.Image.size."""

    def __init__(self, src, out_hw=None, flip=False):
        if isinstance(src, np.ndarray):
            src = torch.from_numpy(np.ascontiguousarray(src))
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
            raise TypeError("DeviceImage needs a uint8 [H, W, 3] RGB array")
        if not torch.cuda.is_available():
            raise _lib.OsdError("oneshotdet_amd.transforms needs an MI355X: there is no CPU path")
        self.src = src.cuda().contiguous()
        self.out_hw = tuple(out_hw) if out_hw is not None else (int(src.shape[0]), int(src.shape[1]))
        self.flip = bool(flip)

    @property
    def size(self):
        return (self.out_hw[1], self.out_hw[0])

    def resized(self, hw):
        return DeviceImage(self.src, hw, self.flip)

    def flipped(self):
        return DeviceImage(self.src, self.out_hw, not self.flip)

    def crop(self, box):
        """PIL.Image.crop((left, upper, right, lower)) — how the dataset cuts a support patch out of its image
        (data/datasets/coco.py:350: `img.crop((x, y, x + w, y + h))` on COCO's float boxes): the coordinates are rounded
        (`int(round(v))`, as Image.crop does), the part outside the image is black.  Only before any Resize / flip."""
        if self.flip or self.out_hw != (int(self.src.shape[0]), int(self.src.shape[1])):
            raise ValueError("DeviceImage.crop applies to the untransformed image")
        x0, y0, x1, y1 = [int(round(float(v))) for v in box]
        if x1 <= x0 or y1 <= y0:
            raise ValueError("empty crop %r" % (box,))
        h, w = int(self.src.shape[0]), int(self.src.shape[1])
        cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
        if (cx0, cy0, cx1, cy1) == (x0, y0, x1, y1):
            return DeviceImage(self.src[y0:y1, x0:x1])
        out = torch.zeros((y1 - y0, x1 - x0, 3), device=self.src.device, dtype=torch.uint8)
        if cx1 > cx0 and cy1 > cy0:
            out[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0] = self.src[cy0:cy1, cx0:cx1]
        return DeviceImage(out)


def _resize_boxes(target, new_size_wh):
    """BoxList.resize (structures/bounding_box.py:91-128): boxes scale with the image; one ratio when both agree."""
    if target is None:
        return None
    from .modules import BoxList
    rw = float(new_size_wh[0]) / float(target.size[0])
    rh = float(new_size_wh[1]) / float(target.size[1])
    b = target.bbox
    if rw == rh:
        nb = b * rw
    else:
        nb = torch.stack([b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh], 1)
    out = BoxList(nb, tuple(new_size_wh), mode=target.mode)
    for k in target.fields():
        out.add_field(k, target.get_field(k))
    return out


def _flip_boxes(target):
    """BoxList.transpose(FLIP_LEFT_RIGHT) (bounding_box.py:130-166): x' = width - x - 1 (TO_REMOVE = 1), ends swapped."""
    if target is None:
        return None
    from .modules import BoxList
    w = target.size[0]
    b = target.bbox
    nb = torch.stack([w - b[:, 2] - 1, b[:, 1], w - b[:, 0] - 1, b[:, 3]], 1)
    out = BoxList(nb, target.size, mode=target.mode)
    for k in target.fields():
        out.add_field(k, target.get_field(k))
    return out


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target=None):
        if not isinstance(image, DeviceImage):
            image = DeviceImage(np.asarray(image))        # a PIL image or an [H, W, 3] uint8 array
        for t in self.transforms:
            image, target = t(image, target)
        return image, target

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join("\n    {0}".format(t) for t in self.transforms) + "\n)"


class Resize(object):
    def __init__(self, min_size, max_size):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size

    def get_size(self, image_size):
        """transforms.py:35-57, (w, h) -> (oh, ow): short side to min_size unless the long side would pass max_size."""
        w, h = image_size
        size = random.choice(self.min_size)
        if self.max_size is not None:
            lo, hi = float(min(w, h)), float(max(w, h))
            if hi / lo * size > self.max_size:
                size = int(round(self.max_size * lo / hi))
        if (w <= h and w == size) or (h <= w and h == size):
            return (h, w)
        if w < h:
            return (int(size * h / w), size)
        return (size, int(size * w / h))

    def __call__(self, image, target=None):
        image = image.resized(self.get_size(image.size))
        return image, _resize_boxes(target, image.size)

    def __repr__(self):
        return "Resize(min_size=%s, max_size=%s)" % (self.min_size, self.max_size)


class RandomHorizontalFlip(object):
    def __init__(self, prob=0.5, rng=None):
        # rng: the source of randomness (anything with .random()); default = Python's global `random`, like the reference
        self.prob, self.rng = prob, rng

    def __call__(self, image, target=None):
        if (self.rng if self.rng is not None else random).random() < self.prob:
            return image.flipped(), _flip_boxes(target)
        return image, target


class ToTensor(object):
    """The float conversion happens inside the fused kernel; in the Compose this step only marks its place."""

    def __call__(self, image, target=None):
        return image, target


def _launch(image, mean, std, to_bgr255, dst, layout, batch_index, dst_h, dst_w, pad_t, pad_l):
    h, w = int(image.src.shape[0]), int(image.src.shape[1])
    oh, ow = image.out_hw
    need = int(_lib.load().osd_image_transform_workspace_bytes(h, w, oh, ow))
    ws = torch.empty((need // 8 + 1,), device=image.src.device, dtype=torch.int64)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    dt = ops.OSD_F32 if dst.dtype == torch.float32 else ops.OSD_BF16
    _lib.call("osd_image_transform", ops._ptr(image.src), h, w, oh, ow, int(image.flip), int(bool(to_bgr255)), m3, s3, ops._ptr(dst),
              layout, dt, batch_index, dst_h, dst_w, pad_t, pad_l, ops._ptr(ws), ops._stream())


def _launch_batch(images, mean, std, to_bgr255, dst, layout, dst_h, dst_w, pad_t, pad_l):
    """All images of a batch in one launch chain (osd_image_transform_batch): image i -> batch slot i of dst."""
    n = len(images)
    lib = _lib.load()
    need = sum(int(lib.osd_image_transform_workspace_bytes(int(im.src.shape[0]), int(im.src.shape[1]), im.out_hw[0], im.out_hw[1]))
               for im in images)
    ws = torch.empty((need // 8 + 1,), device=dst.device, dtype=torch.int64)
    srcs = (C.c_void_p * n)(*[im.src.data_ptr() for im in images])
    arr = lambda vals: (C.c_int32 * n)(*[int(v) for v in vals])      # noqa: E731
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    dt = ops.OSD_F32 if dst.dtype == torch.float32 else ops.OSD_BF16
    _lib.call("osd_image_transform_batch", n, srcs, arr(im.src.shape[0] for im in images), arr(im.src.shape[1] for im in images),
              arr(im.out_hw[0] for im in images), arr(im.out_hw[1] for im in images), arr(im.flip for im in images),
              int(bool(to_bgr255)), m3, s3, ops._ptr(dst), layout, dt, 0, dst_h, dst_w, pad_t, pad_l, ops._ptr(ws), ops._stream())


class Normalize(object):
    def __init__(self, mean, std, to_bgr255=True):
        self.mean, self.std, self.to_bgr255 = tuple(mean), tuple(std), to_bgr255

    def __call__(self, image, target=None):
        """-> float32 [3, oh, ow] on the device: exactly the tensor the reference's Compose returns."""
        oh, ow = image.out_hw
        out = torch.empty((1, 3, oh, ow), device=image.src.device, dtype=torch.float32)
        _launch(image, self.mean, self.std, self.to_bgr255, out, 0, 0, oh, ow, 0, 0)
        return out[0], target


def build_transforms(min_size=800, max_size=1200, supp_min_size=200, supp_max_size=400, is_train=True, mean=PIXEL_MEAN,
                     std=PIXEL_STD, to_bgr255=True):
    """data/transforms/build.py:5-52 with the config of record's sizes (yaml :39-47) as defaults -> [transform,
    transform_supp] (target images, support / query images)."""
    flip_prob = 0.5 if is_train else 0
    norm = Normalize(mean=mean, std=std, to_bgr255=to_bgr255)
    return [Compose([Resize(min_size, max_size), RandomHorizontalFlip(flip_prob), ToTensor(), norm]),
            Compose([Resize(supp_min_size, supp_max_size), RandomHorizontalFlip(flip_prob), ToTensor(), norm])]


def collate(images, size_divisible=32, mean=PIXEL_MEAN, std=PIXEL_STD, to_bgr255=True, stem_dtype=None):
    """Normalize + BatchCollator / to_image_list (collate_batch.py:15-20, image_list.py:52-70) in one pass per image: the
    DeviceImages (as they come out of Resize / RandomHorizontalFlip) are written straight into the zero-padded batch, the
    whole batch in three launches.
    -> layers.ImageList of float32 [N, 3, Hp, Wp] with every image's true (h, w); with stem_dtype (torch.bfloat16 /
    float32) instead a `PackedImages`: the stem conv's NHWC4 input (what osd_pack_image would produce from that batch)."""
    from .layers import ImageList
    mh = max(im.out_hw[0] for im in images)
    mw = max(im.out_hw[1] for im in images)
    if size_divisible > 0:
        mh = int(math.ceil(mh / size_divisible) * size_divisible)
        mw = int(math.ceil(mw / size_divisible) * size_divisible)
    n = len(images)
    dev = images[0].src.device
    sizes = [im.out_hw for im in images]
    if stem_dtype is None:
        batch = torch.empty((n, 3, mh, mw), device=dev, dtype=torch.float32)
        _launch_batch(images, mean, std, to_bgr255, batch, 0, mh, mw, 0, 0)
        return ImageList(batch, sizes)
    ho, wo = ops.conv_out(mh, 7, 2, 3), ops.conv_out(mw, 7, 2, 3)
    hp, wp = max(2 * (ho - 1) + 7, mh + 3), max(2 * (wo - 1) + 8, mw + 3)
    wp += wp & 1
    packed = torch.empty((n, hp, wp, 4), device=dev, dtype=stem_dtype)
    _launch_batch(images, mean, std, to_bgr255, packed, 1, hp, wp, 3, 3)
    return ops.PackedImages(packed, (mh, mw), sizes)
