"""One-shot detection end to end on an MI355X with this build, the way a user of RyanXLi/OneshotDet would call it:

    python examples/detect.py [--checkpoint model_0040000.pth | --c2 R-50.pkl] [--dtype bf16|f32] [--second-stage]

  * weights: a reference `.pth` (utils/checkpoint.py format), a Detectron ResNet `.pkl` for the backbones, or — there is no
    network here — the deterministic synthetic weights the tests use;
  * inputs: lists of CHW BGR-minus-mean images of different sizes for targets and queries; `to_image_list` pads them to
    a common /32 size and keeps the true sizes (data/collate_batch.py + structures/image_list.py of the reference);
  * output: one BoxList per target image (boxes, scores, labels = the query's class id), like `GeneralizedRCNN.forward`.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import checkpoint, layers, modules, spec, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--c2", default="")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--first-stage-only", action="store_true")
    args = ap.parse_args()
    shapes = spec.hot_path_shapes() if args.first_stage_only else spec.full_model_shapes()
    defaults = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes).items()}
    if args.checkpoint:
        sd, extras = checkpoint.load_checkpoint(args.checkpoint, defaults=defaults)
        print("loaded", args.checkpoint, {k: v for k, v in extras.items() if not isinstance(v, dict)})
    elif args.c2:
        sd = checkpoint.load_c2_resnet(args.c2, defaults)
        print("backbones initialised from", args.c2)
    else:
        sd = defaults
        print("synthetic weights (no checkpoint given)")
    det = modules.OneShotDetector(sd, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    # two targets and two queries of different sizes
    targets = [torch.from_numpy(synth.make_images("ex.t%d" % i, 1, h, w)[0]) for i, (h, w) in enumerate([(480, 640), (512, 384)])]
    queries = [torch.from_numpy(synth.make_images("ex.q%d" % i, 1, h, w)[0]) for i, (h, w) in enumerate([(127, 127), (96, 160)])]
    images = layers.to_image_list(targets, size_divisible=spec.SIZE_DIVISIBILITY)
    images_supp = layers.to_image_list(queries, size_divisible=spec.SIZE_DIVISIBILITY)
    print("padded batch", tuple(images.tensors.shape), "true sizes", images.image_sizes)
    results = det(images, images_supp, target_ids=[17, 3])
    torch.cuda.synchronize()
    for i, bl in enumerate(results):
        s = bl.get_field("scores")
        print("image %d: %s, top score %.3f, label %s" % (i, bl, float(s[0]) if len(bl) else float("nan"),
                                                          int(bl.get_field("labels")[0]) if bl.has_field("labels") and len(bl) else "-"))


if __name__ == "__main__":
    main()
