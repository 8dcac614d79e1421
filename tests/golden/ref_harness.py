"""Import the real reference (RyanXLi/OneshotDet, /root/reference) on CPU, in THIS container only.

Test infrastructure for *generating* golden fixtures (tests/golden/make_golden.py) and for validating the
oracle restatement (oracle/hotpath_ref.py).  Never imported by the product, never needed on the GPU box:
`/root/reference` does not exist there and nothing under tests marked gpu touches this file.

The reference cannot be imported as shipped in this image: four oracle-only shims are needed.
  1. `maskrcnn_benchmark._C`: the reference's csrc/cpu sources are compiled with torch.utils.cpp_extension from a
     SCRATCH COPY under /tmp (never inside this repo).  Two lines need `x.type()` -> `x.scalar_type()` for torch 2.10
     (csrc/cpu/nms_cpu.cpp:71, csrc/cpu/ROIAlign_cpu.cpp:242); the sed below applies exactly that to the scratch copy.
  2. `yacs.config.CfgNode`: tiny attribute-dict stand-in (the reference only uses attribute access, merge_from_file,
     merge_from_list, freeze/defrost/clone).
  3. `cv2`, `pycocotools(.mask)`: empty stub modules (imported at module scope by code that is never executed here).
  4. `torch._six`: removed from torch; `PY3 = True`.
  5. `torchvision.transforms.functional` (torchvision is absent): the four thin wrappers the reference's
     data/transforms/transforms.py calls — resize / hflip / to_tensor / normalize — written as torchvision==0.2.1
     (INSTALL.md:5) has them: `img.resize(size[::-1], BILINEAR)`, `img.transpose(FLIP_LEFT_RIGHT)`, uint8 HWC -> float CHW
     `.div(255)`, per-channel `t.sub_(m).div_(s)`.  The arithmetic they forward to is PIL's and torch's own; the
     reference's Resize.get_size, Compose order, Normalize (BGR255) and BoxList code run unmodified.
  6. (dataset fixture only: install_coco_shims) `pycocotools.coco.COCO` and `torchvision.datasets.coco.CocoDetection`, both
     absent: the index queries the reference's data/datasets/coco.py makes (getCatIds, getImgIds(catIds), getAnnIds(imgIds,
     catIds, iscrowd), loadAnns / loadImgs / loadCats, .imgs) with pycocotools 2.0's filter order, and torchvision 0.2.1's
     CocoDetection (`ids = list(coco.imgs.keys())`, `__getitem__` -> (PIL RGB image, loadAnns(getAnnIds(imgIds=id)))).  The
     category choice, catalog, shuffles (Python's `random`, seed 6666), annotation filter, BoxList conversion / clipping,
     support selection, crop and augmentation — data/datasets/coco.py:56-547 — run unmodified on top of them.
"""
import os
import re
import shutil
import sys
import types

REFERENCE_ROOT = "/root/reference"
SCRATCH = "/tmp/osd_ref_scratch"
CONFIG_OF_RECORD = "configs/fcos/2019_10_25_vanilla_siamse_backbone.yaml"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "maskrcnn_benchmark"))


class _CfgNode(dict):
    """Minimal yacs.config.CfgNode stand-in (shim 2)."""

    def __init__(self, init=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        if init:
            for k, v in init.items():
                self[k] = _CfgNode(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError("frozen cfg")
        self[k] = v

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, _CfgNode):
                v.freeze()

    def defrost(self):
        self.__dict__["_frozen"] = False
        for v in self.values():
            if isinstance(v, _CfgNode):
                v.defrost()

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        import copy
        n = _CfgNode()
        for k, v in self.items():
            n[k] = copy.deepcopy(v, memo)
        return n

    @staticmethod
    def _coerce(new, old):
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        return new

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self:
                    self[k] = _CfgNode()
                self[k]._merge(v)
            else:
                if isinstance(v, str):  # yacs literal_evals yaml strings such as "(0.125, 0.0625)"
                    import ast
                    try:
                        v = ast.literal_eval(v)
                    except Exception:
                        pass
                self[k] = self._coerce(v, self.get(k)) if k in self else v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for key, val in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if isinstance(val, str):
                import ast
                try:
                    val = ast.literal_eval(val)
                except Exception:
                    pass
            node[parts[-1]] = self._coerce(val, node.get(parts[-1]))


def _install_shims():
    if "yacs" not in sys.modules:
        yacs = types.ModuleType("yacs")
        yacs_config = types.ModuleType("yacs.config")
        yacs_config.CfgNode = _CfgNode
        yacs.config = yacs_config
        sys.modules["yacs"] = yacs
        sys.modules["yacs.config"] = yacs_config
    for name in ("cv2", "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    if "torch._six" not in sys.modules:
        import torch
        six = types.ModuleType("torch._six")
        six.PY3 = True
        six.string_classes = (str,)
        sys.modules["torch._six"] = six
        torch._six = six
    try:
        import torchvision  # noqa: F401
    except Exception:
        tv = types.ModuleType("torchvision")
        for sub in ("transforms", "datasets", "models", "ops"):
            m = types.ModuleType("torchvision." + sub)
            setattr(tv, sub, m)
            sys.modules["torchvision." + sub] = m
        tvf = types.ModuleType("torchvision.transforms.functional")
        sys.modules["torchvision.transforms.functional"] = tvf
        tv.transforms.functional = tvf

        def _resize(img, size, interpolation=2):                     # torchvision 0.2.1 functional.py: resize
            from PIL import Image
            assert not isinstance(size, int) and len(size) == 2 and interpolation == Image.BILINEAR
            return img.resize(size[::-1], interpolation)

        def _hflip(img):                                             # functional.py: hflip
            from PIL import Image
            return img.transpose(Image.FLIP_LEFT_RIGHT)

        def _to_tensor(pic):                                         # functional.py: to_tensor (PIL RGB branch)
            import numpy as np
            import torch
            assert pic.mode == "RGB"
            img = torch.from_numpy(np.frombuffer(pic.tobytes(), dtype=np.uint8).copy())
            img = img.view(pic.size[1], pic.size[0], 3)
            img = img.transpose(0, 1).transpose(0, 2).contiguous()
            return img.float().div(255)

        def _normalize(tensor, mean, std):                           # functional.py: normalize
            for t, m, s in zip(tensor, mean, std):
                t.sub_(m).div_(s)
            return tensor
        tvf.resize, tvf.hflip, tvf.to_tensor, tvf.normalize = _resize, _hflip, _to_tensor, _normalize
        tvc = types.ModuleType("torchvision.datasets.coco")

        class _CocoDetection(object):
            pass
        tvc.CocoDetection = _CocoDetection
        tv.datasets.coco = tvc
        tv.datasets.CocoDetection = _CocoDetection
        sys.modules["torchvision.datasets.coco"] = tvc
        sys.modules["torchvision"] = tv


def _build_C():
    """Shim 1: compile the reference's CPU extension from a scratch copy (needs 2 one-token patches)."""
    from torch.utils.cpp_extension import load
    src = os.path.join(SCRATCH, "maskrcnn_benchmark", "csrc")
    for rel, pat in (("cpu/nms_cpu.cpp", r"AT_DISPATCH_FLOATING_TYPES\(dets\.type\(\)"),
                     ("cpu/ROIAlign_cpu.cpp", r"AT_DISPATCH_FLOATING_TYPES\(input\.type\(\)")):
        p = os.path.join(src, rel)
        s = open(p).read()
        s2 = re.sub(pat, lambda m: m.group(0).replace(".type()", ".scalar_type()"), s)
        if s2 != s:
            open(p, "w").write(s2)
    sources = [os.path.join(src, "vision.cpp"), os.path.join(src, "cpu", "nms_cpu.cpp"),
               os.path.join(src, "cpu", "ROIAlign_cpu.cpp")]
    os.makedirs(os.path.join(SCRATCH, "build"), exist_ok=True)
    return load(name="osd_ref_C", sources=sources, extra_include_paths=[src],
                build_directory=os.path.join(SCRATCH, "build"), verbose=False,
                extra_cflags=["-O2", "-w"])


_loaded = {}


def load_reference():
    """Returns the imported `maskrcnn_benchmark` package (CPU), building the scratch copy on first use."""
    if "pkg" in _loaded:
        return _loaded["pkg"]
    if not reference_available():
        raise RuntimeError("reference not present at %s (it only exists in the build container)" % REFERENCE_ROOT)
    if not os.path.isdir(os.path.join(SCRATCH, "maskrcnn_benchmark")):
        os.makedirs(SCRATCH, exist_ok=True)
        shutil.copytree(os.path.join(REFERENCE_ROOT, "maskrcnn_benchmark"),
                        os.path.join(SCRATCH, "maskrcnn_benchmark"))
    _install_shims()
    C = _build_C()
    sys.modules["maskrcnn_benchmark._C"] = C
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)
    import maskrcnn_benchmark
    maskrcnn_benchmark._C = C
    _loaded["pkg"] = maskrcnn_benchmark
    return maskrcnn_benchmark


def build_reference_model(extra_opts=()):
    """build_detection_model(cfg) with the config of record on CPU (SURVEY.md §8c 'What runs')."""
    load_reference()
    from maskrcnn_benchmark.config import cfg as global_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = global_cfg.clone()
    cfg.defrost()
    cfg.merge_from_file(os.path.join(REFERENCE_ROOT, CONFIG_OF_RECORD))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHT", ""] + list(extra_opts))
    cfg.freeze()
    model = build_detection_model(cfg)
    return model, cfg


class _CocoIndex(object):
    """pycocotools.coco.COCO, the part data/datasets/coco.py uses (shim 6)."""

    def __init__(self, annotation_file):
        import json
        self.dataset = json.load(open(annotation_file))
        self.anns = {a["id"]: a for a in self.dataset["annotations"]}
        self.imgs = {i["id"]: i for i in self.dataset["images"]}
        self.cats = {c["id"]: c for c in self.dataset["categories"]}
        self.imgToAnns, self.catToImgs = {}, {}
        for a in self.dataset["annotations"]:
            self.imgToAnns.setdefault(a["image_id"], []).append(a)
            self.catToImgs.setdefault(a["category_id"], []).append(a["image_id"])

    @staticmethod
    def _lst(v):
        return list(v) if isinstance(v, (list, tuple)) else [v]

    def getCatIds(self):
        return [c["id"] for c in self.dataset["categories"]]

    def getImgIds(self, imgIds=[], catIds=[]):
        imgIds, catIds = self._lst(imgIds), self._lst(catIds)
        if len(imgIds) == len(catIds) == 0:
            return list(self.imgs.keys())
        ids = set(imgIds)
        for i, c in enumerate(catIds):
            if i == 0 and len(ids) == 0:
                ids = set(self.catToImgs.get(c, []))
            else:
                ids &= set(self.catToImgs.get(c, []))
        return list(ids)

    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
        imgIds, catIds = self._lst(imgIds), self._lst(catIds)
        if len(imgIds) == len(catIds) == len(areaRng) == 0:
            anns = self.dataset["annotations"]
        else:
            anns = [a for i in imgIds for a in self.imgToAnns.get(i, [])] if len(imgIds) else self.dataset["annotations"]
            anns = anns if len(catIds) == 0 else [a for a in anns if a["category_id"] in catIds]
        if iscrowd is not None:
            return [a["id"] for a in anns if a["iscrowd"] == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids=[]):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadImgs(self, ids=[]):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]

    def loadCats(self, ids=[]):
        return [self.cats[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.cats[ids]]


def install_coco_shims():
    """Shim 6 (after load_reference): give the stub torchvision CocoDetection torchvision 0.2.1's behaviour on top of _CocoIndex."""
    base = sys.modules["torchvision.datasets.coco"].CocoDetection

    def __init__(self, root, annFile, transform=None, target_transform=None):
        self.root = root
        self.coco = _CocoIndex(annFile)
        self.ids = list(self.coco.imgs.keys())
        self.transform, self.target_transform = transform, target_transform

    def __getitem__(self, index):
        from PIL import Image
        img_id = self.ids[index]
        target = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id))
        img = Image.open(os.path.join(self.root, self.coco.loadImgs(img_id)[0]["file_name"])).convert("RGB")
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return img, target

    def __len__(self):
        return len(self.ids)
    base.__init__, base.__getitem__, base.__len__ = __init__, __getitem__, __len__
    sys.modules["pycocotools.coco"].COCO = _CocoIndex
