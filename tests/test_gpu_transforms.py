"""GPU (-m gpu): the fused input transforms (csrc/transforms.hip behind oneshotdet_amd/transforms.py, the reference's
data/transforms interface) against fixtures recorded through the reference's own transforms (tests/golden/transforms.npz)
and against the oracle (oracle/transforms_ref.py).  Integer resampling and operation-by-operation float32: BIT-EXACT."""
import hashlib

import numpy as np
import pytest
import torch

import golden_utils as gu
from oneshotdet_amd import spec, synth
from golden.make_golden_cases import TRANSFORM_CASES, TRANSFORM_SIZES, transform_source
from oracle import transforms_ref as otr

pytestmark = pytest.mark.gpu


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).digest()


def _compose(pipe, flip):
    from oneshotdet_amd import transforms as T
    mn, mx = TRANSFORM_SIZES[pipe]
    return T.Compose([T.Resize(mn, mx), T.RandomHorizontalFlip(1.0 if flip else 0.0), T.ToTensor(),
                      T.Normalize(T.PIXEL_MEAN, T.PIXEL_STD, to_bgr255=True)])


@pytest.mark.parametrize("case", TRANSFORM_CASES, ids=[c[0] for c in TRANSFORM_CASES])
def test_compose_matches_reference_fixture_bit_for_bit(case):
    """Compose(Resize, RandomHorizontalFlip, ToTensor, Normalize) as build.py:39-46 assembles it: the float CHW tensor and
    the boxes equal what the reference's Compose produced on the same uint8 image (sha256 of all values + samples)."""
    from oneshotdet_amd.modules import BoxList
    name, hw, pipe, flip = case
    f = gu.load("transforms.npz")
    src = transform_source(name, hw)
    tgt = BoxList(torch.from_numpy(f[name + ".boxes_in"]), (hw[1], hw[0]), mode="xyxy")
    tgt.add_field("labels", torch.ones(len(tgt), dtype=torch.int64))
    img, out_t = _compose(pipe, flip)(src, tgt)
    got = img.cpu().numpy()
    assert got.shape == tuple(f[name + ".shape"]) and got.dtype == np.float32
    flat = got.reshape(-1)
    assert np.array_equal(flat[gu.sample_indices(flat.size, "transform." + name)], f[name + ".samples"])
    assert _sha(got) == f[name + ".sha256"].tobytes()
    if name + ".full" in f.files:
        assert np.array_equal(got, f[name + ".full"])
    assert np.array_equal(out_t.bbox.numpy(), f[name + ".boxes_out"])
    assert out_t.size == (got.shape[2], got.shape[1]) and out_t.has_field("labels")


def test_collate_writes_the_padded_batch_and_the_stem_input():
    """transforms.collate = Normalize + BatchCollator / to_image_list (collate_batch.py:15-20, image_list.py:52-70): the
    zero-padded float batch equals the reference's (sha256); with stem_dtype it equals osd_pack_image of that batch bit
    for bit, so the float batch never has to exist."""
    from oneshotdet_amd import ops, transforms as T
    f = gu.load("transforms.npz")
    names = ("landscape", "portrait_maxsize", "landscape_flip")
    cases = {c[0]: c for c in TRANSFORM_CASES}
    imgs = []
    for n in names:
        _, hw, pipe, flip = cases[n]
        im = T.DeviceImage(transform_source(n, hw))
        im, _ = T.Resize(*TRANSFORM_SIZES[pipe])(im, None)
        if flip:
            im = im.flipped()
        imgs.append(im)
    batch = T.collate(imgs, 32)
    assert tuple(batch.tensors.shape) == tuple(f["batch.shape"])
    assert [tuple(s) for s in batch.image_sizes] == [tuple(s) for s in f["batch.sizes"]]
    assert _sha(batch.tensors.cpu().numpy()) == f["batch.sha256"].tobytes()
    for dt in (torch.float32, torch.bfloat16):
        packed = T.collate(imgs, 32, stem_dtype=dt)
        ref, _ = ops.stem_input(batch.tensors, dt)
        assert packed.tensor.shape == ref.shape and torch.equal(packed.tensor, ref)
        assert packed.shape == tuple(batch.tensors.shape) and packed.image_sizes == batch.image_sizes


def test_random_sizes_against_the_oracle():
    """Up- and down-scaling at odd sizes, one axis unchanged, extreme ratios, without BGR255: product == oracle."""
    from oneshotdet_amd import transforms as T
    rng = np.random.RandomState(3)
    for (h, w, mn, mx, flip, bgr) in [(37, 53, 80, 200, False, True), (120, 90, 40, 60, True, True), (64, 100, 64, 400, True, False),
                                      (9, 300, 20, 64, False, True), (301, 17, 33, 3000, True, True), (50, 70, 50, 70, False, False)]:
        src = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        comp = T.Compose([T.Resize(mn, mx), T.RandomHorizontalFlip(1.0 if flip else 0.0), T.ToTensor(),
                          T.Normalize((1.5, 100.25, 7.0), (2.0, 0.5, 3.0), to_bgr255=bgr)])
        got, _ = comp(src, None)
        ref = otr.transform_image(src, mn, mx, flip, (1.5, 100.25, 7.0), (2.0, 0.5, 3.0), bgr)
        assert got.shape == ref.shape and np.array_equal(got.cpu().numpy(), ref), (h, w, mn, mx)


def test_batched_chain_equals_the_per_image_one_for_mixed_batches():
    """collate runs the whole batch through osd_image_transform_batch (three launches, a kernarg table of <= 16 images per
    chain): 19 images of different sizes, flips, up- / down-scaling and unchanged axes land in their slots exactly as the
    per-image transform (pinned to the reference above) writes them, zero padding included, in both layouts."""
    from oneshotdet_amd import ops, transforms as T
    rng = np.random.RandomState(11)
    imgs, singles = [], []
    norm = T.Normalize(T.PIXEL_MEAN, T.PIXEL_STD, to_bgr255=True)
    for i in range(19):
        h, w = int(rng.randint(20, 140)), int(rng.randint(20, 140))
        src = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        im = T.DeviceImage(src)
        if i % 5 != 4:                                   # every fifth image keeps its size (both passes skipped)
            im, _ = T.Resize(int(rng.randint(30, 90)), 160)(im, None)
        if i % 2:
            im = im.flipped()
        imgs.append(im)
        singles.append(norm(im, None)[0])
    batch = T.collate(imgs, 32)
    for i, t in enumerate(singles):
        hh, ww = t.shape[1:]
        assert torch.equal(batch.tensors[i, :, :hh, :ww], t), i
        assert not batch.tensors[i, :, hh:].any() and not batch.tensors[i, :, :, ww:].any()
    packed = T.collate(imgs, 32, stem_dtype=torch.bfloat16)
    ref, _ = ops.stem_input(batch.tensors, torch.bfloat16)
    assert torch.equal(packed.tensor, ref)


def test_crop_matches_pil_crop():
    """DeviceImage.crop = PIL.Image.crop as the dataset uses it to cut support patches (coco.py:350): float COCO boxes are
    rounded, regions outside the image come back black; the cropped patch then runs through the support transform exactly
    like a PIL patch would (compared through the oracle's resize of PIL's crop)."""
    from PIL import Image
    from oneshotdet_amd import transforms as T
    rng = np.random.RandomState(5)
    src = rng.randint(0, 256, (90, 130, 3)).astype(np.uint8)
    pil = Image.fromarray(src)
    for box in [(10.4, 5.5, 70.6, 60.49), (0, 0, 130, 90), (-7.2, 20.0, 40.0, 95.5), (100.5, 70.5, 140.2, 99.0), (3, 4, 5, 6)]:
        ref = np.asarray(pil.crop(box))
        got = T.DeviceImage(src).crop(box)
        assert got.size == (ref.shape[1], ref.shape[0]) and np.array_equal(got.src.cpu().numpy(), ref), box
        comp = T.Compose([T.Resize(40, 80), T.RandomHorizontalFlip(0.0), T.ToTensor(), T.Normalize(T.PIXEL_MEAN, T.PIXEL_STD)])
        out, _ = comp(got, None)
        assert np.array_equal(out.cpu().numpy(), otr.transform_image(np.ascontiguousarray(ref), 40, 80, False, T.PIXEL_MEAN, T.PIXEL_STD, True)), box
    with pytest.raises(ValueError):
        T.DeviceImage(src).crop((10, 10, 10, 20))


def test_engine_accepts_the_packed_stem_input():
    """HotPathEngine.detect on transforms.collate(..., stem_dtype) gives the same proposals as on the float batch."""
    from oneshotdet_amd import model, spec, synth, transforms as T
    eng = model.HotPathEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
    tiny, supp = TRANSFORM_SIZES["tiny"], (32, 64)
    tg = [T.Resize(*tiny)(T.DeviceImage(transform_source("e%d" % i, hw)), None)[0] for i, hw in enumerate([(90, 120), (100, 75)])]
    qs = [T.Resize(*supp)(T.DeviceImage(transform_source("q%d" % i, hw)), None)[0] for i, hw in enumerate([(40, 40), (30, 50)])]
    a = eng.detect(T.collate(tg, 32), T.collate(qs, 32))
    b = eng.detect(T.collate(tg, 32, stem_dtype=torch.bfloat16), T.collate(qs, 32, stem_dtype=torch.bfloat16))
    for x, y in zip(a["proposals"], b["proposals"]):
        assert torch.equal(x, y)
    assert int(a["proposals"][2].min()) > 0


def test_train_engine_accepts_collated_batches():
    """TrainEngine.forward_backward on what the input pipeline produces — transforms.collate(...) as layers.ImageList or,
    with stem_dtype, as ops.PackedImages (different true sizes inside one padded batch) — gives exactly the losses and
    gradients of the same step on the float tensors with image_sizes passed by hand."""
    from oneshotdet_amd import train, transforms as T
    rng = np.random.RandomState(4)
    tg = [T.Resize(96, 160)(T.DeviceImage(rng.randint(0, 256, (60 + 14 * i, 100, 3)).astype(np.uint8)), None)[0] for i in range(2)]
    qs = [T.Resize(48, 64)(T.DeviceImage(rng.randint(0, 256, (40, 40 + 6 * i, 3)).astype(np.uint8)), None)[0] for i in range(2)]
    gtb = torch.tensor([[[10., 12., 60., 70.]], [[30., 20., 90., 80.]]]).cuda()
    cnt = torch.tensor([1, 1], dtype=torch.int32).cuda()
    sd = synth.make_state_dict(spec.hot_path_shapes())
    outs = []
    for mode in ("float", "imagelist", "packed"):
        eng = train.TrainEngine(sd, dtype=torch.float32)
        ti, qi = T.collate(tg, 32), T.collate(qs, 32)
        if mode == "float":
            losses = eng.forward_backward(ti.tensors, qi.tensors, gtb, cnt, image_sizes=ti.image_sizes)
        elif mode == "imagelist":
            losses = eng.forward_backward(ti, qi, gtb, cnt)
        else:
            losses = eng.forward_backward(T.collate(tg, 32, stem_dtype=torch.float32), T.collate(qs, 32, stem_dtype=torch.float32), gtb, cnt)
        torch.cuda.synchronize()
        outs.append((losses.clone().cpu(), eng.flat_g.clone().cpu()))
    # the hand-fed float run pools the queries as full-size images: only the two collated forms see the true query sizes
    torch.testing.assert_close(outs[2][0], outs[1][0], rtol=1e-6, atol=0)
    assert (outs[2][1] - outs[1][1]).abs().max() <= 1e-5 * outs[1][1].abs().max()
    assert torch.isfinite(outs[0][0]).all()


def test_no_cpu_path():
    from oneshotdet_amd import _lib, transforms as T
    with pytest.raises(TypeError):
        T.DeviceImage(np.zeros((4, 4, 3), np.float32))
    assert _lib.load().osd_image_transform_workspace_bytes(0, 4, 4, 4) == 0
