"""GPU-side sample preparation.
CPU: the numpy oracle (oracle/data_ref.py) and the host logic (PFM header, crop draws) against tests/golden/data.npz —
samples the reference's own CustomDataset / RandomCrop / ToTensor / readPFM produced (oracle/make_golden.py gen_data).
GPU: `sdhip_prepare_sample` (through data.SamplePreparer) bit-exact against the same golden tensors."""
import os

import numpy as np
import pytest
import torch

from oracle import data_ref as DR

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data.npz")
CASES = ("roses_linear", "roses_crop", "roses_sigmoid_be", "garden_tanh", "city_linear", "city_flip", "city_flip_whole",
         "roses_slice", "garden_double", "city_slice_double_down", "kitti_band")


def _case(gold, name):
    g = lambda k: gold["%s.%s" % (name, k)]
    ds = str(g("dataset"))
    depth = g("depth_file").tobytes() if ds in ("roses", "garden") else g("depth_u16")
    return dict(left=g("left_u8"), right=g("right_u8"), seg=g("seg_u8"), depth=depth, dataset=ds, n_labels=int(g("n_labels")),
                max_d=float(g("max_d")), activation=str(g("activation")), normalize=g("normalize"), crop=tuple(int(v) for v in g("crop")),
                seed=int(g("seed")), flip=bool(int(g("flip"))), opt_slice=bool(int(g("opt_slice"))), opt_double=bool(int(g("opt_double"))),
                opt_down=bool(int(g("opt_down"))))


def _draws(c):
    """(crop, row_roll, double_left, flip): RandomCrop's random decisions from the fixture's seed, drawn in the reference's
    order (util/utilTorchDataLoader.py:435-476) by the product's host-side helpers."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import data as PD
    torch.manual_seed(c["seed"])
    H, W = c["left"].shape[:2]
    crop = PD.draw_crop(H, W, list(c["crop"]), c["dataset"], is_down=c["opt_down"])
    roll = PD.draw_slice_and_switch(H, crop[2], c["opt_slice"])
    dbl = PD.draw_double_left(c["opt_double"])
    flip = PD.draw_flip(c["dataset"], c["flip"])
    return crop, roll, dbl, flip


def _crop(c):
    return _draws(c)[0]


@pytest.mark.parametrize("name", CASES)
def test_oracle_and_crop_draws_match_reference_loader(name):
    gold = np.load(GOLD)
    c = _case(gold, name)
    crop, roll, dbl, flip = _draws(c)
    lut = gold["cityscapes.id2trainId"]
    out = DR.prepare_sample(c["left"], c["right"], c["seg"], c["depth"], c["dataset"], c["n_labels"], c["max_d"], c["activation"],
                            c["normalize"], crop, id2train=lut)
    out = DR.slice_and_switch(*out, roll)
    assert flip == name.startswith("city_flip") and dbl == ("double" in name) and (roll > 0) == ("slice" in name)
    if dbl:
        out = DR.double_left(*out)
    if flip:                                   # RandomCrop(flipHorizontal=True): the draw after the crop offsets says "flip"
        out = DR.flip_sample(*out)
    for k, v in zip(("left", "right", "seg", "disp"), out):
        np.testing.assert_array_equal(v, gold["%s.%s" % (name, k)], err_msg=k)


def test_pfm_header_rules():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import parse_pfm_header
    assert parse_pfm_header(b"Pf\n40 24\n-1.000000\n" + b"\0" * 8) == (False, 40, 24, 1.0, True, 19)
    assert parse_pfm_header(b"PF\n7 3\n2.5\n")[:5] == (True, 7, 3, 2.5, False)
    with pytest.raises(Exception, match="Not a PFM"):
        parse_pfm_header(b"P5\n7 3\n1\n")
    with pytest.raises(Exception, match="Malformed"):
        parse_pfm_header(b"Pf\n7  3\n1\n")
    gold = np.load(GOLD)
    buf = gold["roses_sigmoid_be.depth_file"].tobytes()
    color, w, h, scale, little, off = parse_pfm_header(buf)
    assert (color, w, h, little) == (False, 40, 24, False)
    np.testing.assert_array_equal(DR.read_pfm(buf)[0], np.flipud(np.frombuffer(buf[off:], ">f4").reshape(h, w)))


def test_cityscapes_table_equals_reference():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import cityscapes_lut
    ref = np.load(GOLD)["cityscapes.id2trainId"]
    want = np.where((ref == 255) | (ref < 0), 19, ref).astype(np.uint8)
    np.testing.assert_array_equal(cityscapes_lut(19), want)


def test_crop_draw_consumes_generator_like_reference():
    """Whole-image mode draws nothing; the plain path draws multinomial + 2 randint, in that order."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import draw_crop
    torch.manual_seed(5)
    assert draw_crop(24, 40, [0, 0]) == (0, 0, 24, 40)
    a = torch.rand(1)
    torch.manual_seed(5)
    assert torch.equal(torch.rand(1), a)
    torch.manual_seed(5)
    torch.multinomial(torch.tensor([0.2, 0.8]), 1)
    t, l = int(torch.randint(0, 24 - 16 + 1, (1,))), int(torch.randint(0, 40 - 24 + 1, (1,)))
    torch.manual_seed(5)
    assert draw_crop(24, 40, (16, 24)) == (t, l, 16, 24)
    assert draw_crop(24, 40, (16, 24), is_down=True) == (8, 8, 16, 24)


def test_kitti_lower_band_and_slice_rules():
    """kitti: with probability 0.8 the crop comes from the lowest new_h + 100 rows (util/utilTorchDataLoader.py:442-445); the
    sliceandSwitch cut is computed from the UNCROPPED row count and is a no-op when it falls beyond the crop."""
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import data as PD
    tops = []
    for seed in range(40):
        torch.manual_seed(seed)
        hit = bool(torch.multinomial(torch.tensor([0.2, 0.8]), 1).item())
        torch.manual_seed(seed)
        top = PD.draw_crop(400, 600, (50, 100), "kitti")[0]
        assert (top >= 250) if hit else (0 <= top <= 350)
        tops.append((hit, top))
        torch.manual_seed(seed)
        assert 0 <= PD.draw_crop(400, 600, (50, 100), "roses")[0] <= 350
    assert any(h for h, _ in tops) and any(not h for h, _ in tops)
    for seed in range(10):
        torch.manual_seed(seed)
        div = float(torch.randint(2, 6, (1,)))
        torch.manual_seed(seed)
        assert PD.draw_slice_and_switch(400, 50, True) == (int(400 / div) if int(400 / div) < 50 else 0) == 0
        torch.manual_seed(seed)
        assert PD.draw_slice_and_switch(40, 32, True) == int(40 / div)
    assert PD.draw_slice_and_switch(40, 32, False) == 0 and PD.draw_double_left(False) is False


def _gpu_prepare(c, crop, dtype, slot=1, B=3, flip=False, roll=0, dbl=False):
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import SamplePreparer
    sp = SamplePreparer(c["dataset"], c["n_labels"], c["max_d"], c["activation"], c["normalize"], dtype=dtype, device="cuda:0")
    batch = sp.alloc_batch(B, crop[2], crop[3])
    for t in batch:
        t.fill_(-7.0)
    sp.prepare_into(batch, slot, c["left"], c["right"], c["seg"], c["depth"], crop, row_roll=roll)
    if dbl:
        sp.double_left_slot(batch, slot)
    if flip:
        sp.flip_slot(batch, slot)
    torch.cuda.synchronize()
    sp.release()
    return batch


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_prepare_matches_reference_loader(name):
    gold = np.load(GOLD)
    c = _case(gold, name)
    crop, roll, dbl, flip = _draws(c)
    batch = _gpu_prepare(c, crop, torch.float32, flip=flip, roll=roll, dbl=dbl)
    for k, t in zip(("left", "right", "seg", "disp"), batch):
        np.testing.assert_array_equal(t[1].cpu().numpy(), gold["%s.%s" % (name, k)], err_msg=k)     # bit-exact (NaN-free outputs)
        assert float(t[0].min()) == -7.0 and float(t[2].max()) == -7.0                               # other slots untouched
    b16 = _gpu_prepare(c, crop, torch.bfloat16, flip=flip, roll=roll, dbl=dbl)
    for k, t in zip(("left", "right"), b16[:2]):
        want = torch.from_numpy(gold["%s.%s" % (name, k)]).to(torch.bfloat16)
        assert torch.equal(t[1].cpu(), want), k
    np.testing.assert_array_equal(b16[3][1].cpu().numpy(), gold[name + ".disp"])


@pytest.mark.gpu
def test_hip_prepare_full_size_matches_oracle():
    """A full-resolution roses-like sample (1024x2048 source, 512x1024 crop, RGBA inputs): bit-exact against the oracle."""
    rng = np.random.default_rng(3)
    H, W = 1024, 2048
    left = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)[:, :, :3]
    right = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    seg = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    depth = rng.uniform(-1, 30, (H, W)).astype(np.float32)
    pfm = b"Pf\n%d %d\n-1.0\n" % (W, H) + np.flipud(depth).tobytes()
    c = dict(left=left, right=right, seg=seg, depth=pfm, dataset="roses", n_labels=2, max_d=192.0, activation="linear",
             normalize=np.array([[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]], dtype=np.float32))
    crop = (300, 517, 512, 1024)
    batch = _gpu_prepare(c, crop, torch.float32, slot=0, B=1)
    want = DR.prepare_sample(left, right, seg, pfm, "roses", 2, 192.0, "linear", c["normalize"], crop)
    for k, t, w in zip(("left", "right", "seg", "disp"), batch, want):
        np.testing.assert_array_equal(t[0].cpu().numpy(), w, err_msg=k)
    assert batch[2].sum().item() == 512 * 1024      # one-hot: one class per pixel


@pytest.mark.gpu
def test_hip_prepare_rejects_bad_arguments():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import SdhipError
    from pmt_learning_for_semantic_segmentation_and_disparity_amd.data import SamplePreparer
    sp = SamplePreparer("roses", 2, 192, device="cuda:0")
    img = np.zeros((8, 8, 3), dtype=np.uint8)
    pfm = b"Pf\n8 8\n-1.0\n" + np.zeros((8, 8), dtype=np.float32).tobytes()
    batch = sp.alloc_batch(1, 8, 8)
    with pytest.raises(SdhipError):
        sp.prepare_into(batch, 0, img, img, img, pfm, crop=(4, 4, 8, 8))            # crop leaves the image
    with pytest.raises(SdhipError):
        sp.prepare_into(batch, 0, img, img, img[:, :, 0], pfm)                       # roses seg needs 3 channels
    with pytest.raises(SdhipError):
        sp.prepare_into(batch, 0, img, img, img, pfm[:-4])                           # truncated payload
    with pytest.raises(SdhipError):
        sp.prepare_into(batch, 0, img, img, img, b"Pf\n4 4\n-1.0\n" + b"\0" * 64)     # size mismatch
    with pytest.raises(SdhipError):
        SamplePreparer("roses", 2, 192, output_activation="relu", device="cuda:0")
