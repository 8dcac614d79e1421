"""Training end to end on MI355X with this build, the way `tools/train_net.py` + `engine/trainer.py:do_train` of
RyanXLi/OneshotDet drive it (both stages, SGD with the reference's parameter groups, periodic checkpoints, resume):

    python examples/train.py [--iters 20] [--batch 2] [--dtype bf16|f32] [--first-stage-only] [--out /tmp/osd_run]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train.py ...      # one rank per GPU

  * data: `dataset.FewShotCocoDataset` (= the reference's COCODataset: one item per (category, image), seeded epoch order,
    targets of the item's category, support crops of OTHER images' largest objects above the area threshold) over a synthetic
    COCO-format annotation set — there are no datasets here, so the "photos" are random uint8 images of different sizes;
  * input pipeline: `transforms.build_transforms` (Resize / flip / ToTensor / Normalize of the reference, executed by the
    fused device kernels), boxes follow their image; `transforms.collate(..., stem_dtype)` = BatchCollator straight into the
    stem conv's input format;
  * step: `TrainEngine.train_step` = forward + FCOS loss + second-stage loss + backward + (N > 1: bucketed RCCL all-reduce
    behind the backward pass) + SGD + weight repack;
  * checkpoints: `checkpoint.save_training_checkpoint` writes the reference's `{"model", "optimizer", "iteration"}` format
    (reference parameter names, momentum included); `--resume` continues from `last_checkpoint`.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import checkpoint, dataset, spec, synth, train, transforms as T  # noqa: E402


def synthetic_coco(rng, n_images=24, n_categories=3):
    """A COCO-format annotation set over synthetic uint8 "photos" of different sizes (1 - 3 objects each): what the reference
    reads from instances_*.json, for `dataset.FewShotCocoDataset` = its COCODataset (category catalog, seeded epoch order,
    per-item category, targets, support crops by area threshold)."""
    images, anns, pix = [], [], {}
    for k in range(n_images):
        h, w = int(rng.randint(300, 480)), int(rng.randint(400, 640))
        images.append({"id": k + 1, "file_name": "synthetic_%d" % (k + 1), "height": h, "width": w})
        pix[k + 1] = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        for _ in range(int(rng.randint(1, 4))):
            x0, y0 = float(rng.uniform(0, w * 0.6)), float(rng.uniform(0, h * 0.6))
            bw, bh = float(rng.uniform(40, w * 0.35)), float(rng.uniform(40, h * 0.35))
            anns.append({"id": len(anns) + 1, "image_id": k + 1, "category_id": int(rng.randint(1, n_categories + 1)),
                         "bbox": [x0, y0, bw, bh], "area": bw * bh, "iscrowd": 0})
    return {"images": images, "annotations": anns, "categories": [{"id": c + 1, "name": "c%d" % c} for c in range(n_categories)]}, pix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--first-stage-only", action="store_true")
    ap.add_argument("--out", default="/tmp/osd_example_run")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--checkpoint-period", type=int, default=10)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    shapes = spec.hot_path_shapes() if args.first_stage_only else spec.full_model_shapes()

    def make_engine(sd):
        return train.TrainEngine(sd, dtype=dtype, lr=0.0005, second_stage=not args.first_stage_only)
    start = 0
    last = os.path.join(args.out, "last_checkpoint")
    if args.resume and os.path.exists(last):
        eng, start = checkpoint.resume_training(open(last).read().strip(), make_engine)
        print("resumed at iteration", start)
    else:
        eng = make_engine(synth.make_state_dict(shapes))      # or checkpoint.load_checkpoint / load_c2_resnet, see detect.py
    eng.defer_join = True
    if world > 1:
        # RCCL AFTER the engine's streams have their hardware queues (profiles/r3_live_exchange.md: the other order cost 14 %
        # of the step), then the bucketed gradient exchange on the engine's update stream
        import torch.distributed as dist
        eng.warm_streams()
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
        eng.attach_exchange(dist.group.WORLD)
    tf_img, tf_supp = T.build_transforms(min_size=480, max_size=800, supp_min_size=128, supp_max_size=192, is_train=True)
    coco, pix = synthetic_coco(np.random.RandomState(1234))                # the same annotation set on every rank
    ds = dataset.FewShotCocoDataset(coco, lambda info: pix[info["id"]], is_train=True, shot=1, supp_area_threshold=40 * 40)
    os.makedirs(args.out, exist_ok=True)
    for it in range(start, args.iters):
        imgs, supps, targets = [], [], []
        for b in range(args.batch):
            item = ds[((it * world + rank) * args.batch + b) % len(ds)]       # DistributedSampler's split: every rank its own items
            dimg, tgt = T.DeviceImage(item["img"]), item["target"]
            supp = T.DeviceImage(item["img_supp"][0])                          # the support crop (coco.py:341)
            # the Compose's Resize and flip steps (Normalize is fused into the collation below)
            for t in tf_img.transforms[:2]:
                dimg, tgt = t(dimg, tgt)
            for t in tf_supp.transforms[:2]:
                supp, _ = t(supp, None)
            imgs.append(dimg); supps.append(supp); targets.append(tgt)
        images = T.collate(imgs, spec.SIZE_DIVISIBILITY, stem_dtype=dtype)
        queries = T.collate(supps, spec.SIZE_DIVISIBILITY, stem_dtype=dtype)
        g = max(len(t) for t in targets)
        gt = torch.zeros(args.batch, g, 4)
        for i, t in enumerate(targets):
            gt[i, :len(t)] = t.bbox
        cnt = torch.tensor([len(t) for t in targets], dtype=torch.int32)
        losses = eng.train_step(images, queries, gt.cuda(), cnt.cuda())
        if rank == 0 and (it % 5 == 0 or it + 1 == args.iters):
            l = losses.float().cpu()
            extra = "" if eng.box_losses is None else "  loss_classifier %.4f  loss_box_reg %.4f" % tuple(eng.box_losses[:2].float().cpu())
            print("iter %4d  loss_cls %.4f  loss_reg %.4f  loss_centerness %.4f  (%d positives)%s"
                  % (it, l[0], l[1], l[2], int(l[3]), extra))
        if rank == 0 and ((it + 1) % args.checkpoint_period == 0 or it + 1 == args.iters):
            path = os.path.join(args.out, "model_%07d.pth" % (it + 1))
            checkpoint.save_training_checkpoint(path, eng, it + 1)
            print("saved", path)
    eng.join()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
