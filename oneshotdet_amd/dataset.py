"""Few-shot detection dataset items: which (image, category) pairs an epoch holds, the target boxes of an item
and its support crops — the host-side half of the reference's `data/datasets/coco.py:56-547` (COCODataset), on an in-memory
COCO-format annotation set instead of pycocotools + files.

What the reference does, restated (file:line = data/datasets/coco.py):
  * categories keep their json order; contiguous ids are 1-based positions in it; the contiguous ids in `exclude` (the
    training or the test exclusion list, :104-131) drop out;
  * per remaining category, the catalog is the sorted list of images that have at least one non-crowd annotation of it with
    every box side > 1 (:150-170, has_valid_annotation :32-46);
  * the epoch is one item per (category, catalog image), categories in json order, shuffled ONCE with Python's `random`
    seeded 6666 (:65, :176-196);
  * an item's target (:470-492): the image's non-crowd annotations of the item's category, xywh -> xyxy with the "- 1" width
    convention (structures/bounding_box.py:75-87), clipped to the image, empty boxes removed, labels all 1;
  * its supports (get_random_item_from_cat :292-355): the category's catalog, shuffled with the SAME random stream, walked
    until `shot` images other than the item's own are found whose largest object of the category has area >
    INPUT.SUPP_AREA_THRESHOLD; the crop is PIL's `crop((x, y, x + w, y + h))` of that object (coordinates rounded half to even,
    black outside the image); with support augmentation each crop is followed by its horizontal flip (:343-349, NUM_SUPP_AUG 1).
The random stream makes items order-dependent exactly as in the reference: the fixture (tests/golden/dataset.npz, recorded
through the reference class) reads items 0 .. len - 1 once after construction, and so does the test.
Images come from `load_image(image_info) -> uint8 HxWx3` (RGB); `transforms` / `supp_transforms` are applied to the query image
(with its target) and to every support crop, e.g. oneshotdet_amd.transforms.build_transforms' pair.
"""
import random

import numpy as np
import torch

from .modules import BoxList


class CocoIndex(object):
    """The lookups the dataset needs from a COCO-format dict {"images", "annotations", "categories"}."""

    def __init__(self, coco):
        self.images = {im["id"]: im for im in coco["images"]}
        self.category_ids = [c["id"] for c in coco["categories"]]          # json order
        self.by_image = {}
        for a in coco["annotations"]:
            self.by_image.setdefault(a["image_id"], []).append(a)

    def objects(self, image_id, category_id=None, crowd=None):
        """annotations of an image in file order, optionally of one category / one crowd flag"""
        return [a for a in self.by_image.get(image_id, [])
                if (category_id is None or a["category_id"] == category_id) and (crowd is None or a["iscrowd"] == crowd)]

    def images_with(self, category_id):
        return sorted({a["image_id"] for anns in self.by_image.values() for a in anns if a["category_id"] == category_id})


def _usable(objects):
    """has_valid_annotation (:32-46) for detection annotations: something there, and not only boxes with a side <= 1"""
    return len(objects) > 0 and not all(any(side <= 1 for side in o["bbox"][2:]) for o in objects)


def crop_like_pil(image, box_xywh):
    """PIL.Image.crop((x, y, x + w, y + h)) on an HxWx3 array: corners rounded half to even, zeros outside the image."""
    x, y, w, h = box_xywh
    x0, y0, x1, y1 = (int(round(v)) for v in (x, y, x + w, y + h))
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), image.shape[2]), dtype=image.dtype)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, image.shape[1]), min(y1, image.shape[0])
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = image[sy0:sy1, sx0:sx1]
    return out


class FewShotCocoDataset(object):
    """The reference's COCODataset (data/datasets/coco.py:56-547) on an in-memory annotation index.

    Random stream: epoch order and support choices are drawn from ONE private `random.Random(seed)` in the reference's order
    of draws.  The reference seeds and consumes Python's GLOBAL `random`, which its transforms (RandomHorizontalFlip of the
    query and support pipelines, data/transforms/transforms.py:66-75) also draw from between items; the support sequence
    therefore equals the reference's only when no random transform runs between items — how tests/golden/dataset.npz was
    recorded (flip probability 0).  To reproduce the interleaving with is_train transforms, pass the SAME generator as `rng`
    here and to transforms.RandomHorizontalFlip(rng=...); `rng=random` (the module itself, after `random.seed(6666)`) is the
    reference's arrangement exactly."""

    def __init__(self, coco, load_image, is_train=True, shot=1, exclude_contiguous=(), supp_area_threshold=80 * 80,
                 supp_aug=False, transforms=None, supp_transforms=None, selected_category=-1, seed=6666, rng=None):
        self.index = coco if isinstance(coco, CocoIndex) else CocoIndex(coco)
        self.load_image, self.is_train, self.shot = load_image, is_train, int(shot)
        self.supp_area_threshold, self.supp_aug = supp_area_threshold, bool(supp_aug)
        self.transforms, self.supp_transforms = transforms, supp_transforms
        self.rng = rng if rng is not None else random.Random(seed)
        self.categories = [c for pos, c in enumerate(self.index.category_ids) if pos + 1 not in set(exclude_contiguous)]
        self.contiguous = {c: pos + 1 for pos, c in enumerate(self.index.category_ids) if c in self.categories}
        self.catalog = {c: [i for i in self.index.images_with(c) if _usable(self.index.objects(i, c, crowd=0))]
                        for c in self.categories}
        pairs = [(i, c) for c in self.categories if selected_category in (-1, c) for i in self.catalog[c]]
        order = list(range(len(pairs)))
        self.rng.shuffle(order)
        self.ids = [pairs[k][0] for k in order]
        self.chosen_cats = [pairs[k][1] for k in order]

    def __len__(self):
        return len(self.ids)

    def get_img_info(self, idx):
        return self.index.images[self.ids[idx]], self.chosen_cats[idx]

    def target(self, idx):
        """BoxList (xyxy, labels 1) of item idx in the ORIGINAL image's coordinates"""
        info, cat = self.get_img_info(idx)
        w, h = info["width"], info["height"]
        b = torch.as_tensor([o["bbox"] for o in self.index.objects(info["id"], cat, crowd=0)], dtype=torch.float32).reshape(-1, 4)
        x1, y1 = b[:, 0], b[:, 1]
        x2, y2 = x1 + (b[:, 2] - 1).clamp(min=0), y1 + (b[:, 3] - 1).clamp(min=0)
        box = torch.stack([x1.clamp(0, w - 1), y1.clamp(0, h - 1), x2.clamp(0, w - 1), y2.clamp(0, h - 1)], dim=1)
        box = box[(box[:, 3] > box[:, 1]) & (box[:, 2] > box[:, 0])]
        t = BoxList(box, (w, h), mode="xyxy")
        t.add_field("labels", torch.ones(len(box), dtype=torch.int64))
        return t

    def supports(self, category, exclude_image):
        """`shot` support crops of `category` (uint8 arrays; each followed by its flip under support augmentation)"""
        choices = list(self.catalog[category])
        self.rng.shuffle(choices)
        crops = []
        for image_id in choices:
            if image_id == exclude_image:
                continue
            objs = self.index.objects(image_id, category, crowd=0)
            best = objs[0]
            for o in objs:                       # the FIRST of the largest (strict >)
                if o["area"] > best["area"]:
                    best = o
            if best["area"] > self.supp_area_threshold:
                crops.append(crop_like_pil(self.load_image(self.index.images[image_id]), best["bbox"]))
                if len(crops) == self.shot:
                    break
        if len(crops) < self.shot:
            # (the reference indexes past the end of its list here: IndexError, :341)
            raise IndexError("category %d has only %d support candidates above the area threshold, %d shots wanted"
                             % (category, len(crops), self.shot))
        if self.supp_aug:
            crops = [v for c in crops for v in (c, np.ascontiguousarray(c[:, ::-1]))]
        return crops

    def __getitem__(self, idx):
        info, cat = self.get_img_info(idx)
        img = self.load_image(info)
        target = self.target(idx)
        supp = self.supports(cat, info["id"])
        if self.transforms is not None:
            img, target = self.transforms(img, target)
        if self.supp_transforms is not None:
            supp = [self.supp_transforms(s, target)[0] for s in supp]
        return {"img": img, "img_supp": supp, "img_neg_supp": supp, "target": target, "idx": idx, "target_id": cat}
