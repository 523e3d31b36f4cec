"""ORACLE (test infrastructure, never shipped or timed as the product): numpy restatement of the reference's sample
preparation — readPFM (util/utilIOPfm.py:66-101) and the post-decode part of CustomDataset.__getitem__ + RandomCrop's
crop + ToTensor (util/utilTorchDataLoader.py:173-211,247-249,453-457,608-630).  Pinned by tests/golden/data.npz, which
oracle/make_golden.py gen_data produced by running the reference's own classes on the same synthetic files."""
import re

import numpy as np


def read_pfm(buf):
    """readPFM on an in-memory file: (data flipped to top-down, scale)."""
    buf = bytes(buf)
    lines, pos = [], 0
    for _ in range(3):
        end = buf.index(b"\n", pos)
        lines.append(buf[pos:end + 1])
        pos = end + 1
    header = lines[0].rstrip().decode("ascii")
    if header not in ("PF", "Pf"):
        raise Exception("Not a PFM file.")
    m = re.match(r"^(\d+)\s(\d+)\s$", lines[1].decode("ascii"))
    if not m:
        raise Exception("Malformed PFM header.")
    width, height = map(int, m.groups())
    scale = float(lines[2].decode("ascii").rstrip())
    endian = "<" if scale < 0 else ">"
    data = np.frombuffer(buf[pos:], dtype=endian + "f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(np.reshape(data, shape)), abs(scale)


def prepare_sample(left, right, seg, depth, dataset, n_labels, max_d, activation, normalize, crop, id2train=None, f=640, b=0.03):
    """-> left, right (3,h,w) f32, seg (C,h,w) f32, disp (1,h,w) f32.  `depth`: PFM file bytes (roses/garden) or a
    uint16 map (kitti/cityscapes); crop = (top, left, h, w)."""
    if dataset in ("roses", "garden"):
        d = read_pfm(depth)[0]
        with np.errstate(invalid="ignore", divide="ignore"):
            disp = np.where(d > 0, f * b * 1 / d, 0)                              # :173-181
    else:
        disp = depth.astype(np.float32) / 256.0                                    # :183-185
    disp = np.array(disp, dtype=np.float32)
    if activation != "linear":
        disp[disp > max_d] = max_d                                                 # :188-189
    if activation == "sigmoid":
        disp = disp / max_d                                                        # :191-192
    if activation == "tanh":
        disp = np.where(disp != 0, 2 * disp / float(max_d) - 1, -1)                # :196-197
    if dataset in ("kitti", "cityscapes"):
        seg_image = np.zeros(seg.shape[:2] + (n_labels + 1,), dtype=np.float32)    # ImgId2trainId, utilCityscape.py:173-186
        for i in np.unique(seg):
            t = id2train[int(i)]
            seg_image[:, :, n_labels if t == 255 else t] += (seg == i)
    else:
        seg_image = np.zeros(seg.shape[:2] + (n_labels,), dtype=np.float32)        # :199-211
        for j in range(n_labels):
            if dataset == "roses":
                seg_image[:, :, j] = ((seg > 128)[:, :, 2] == j)
            else:
                seg_image[:, :, j] = (seg == j + 1)
    top, lft, h, w = crop
    cut = lambda a: a[top:top + h, lft:lft + w]
    norm = lambda a: ((cut(a) / 255.0 - normalize[0]) / normalize[1]).astype(np.float32)   # :247-248
    chw = lambda a: np.ascontiguousarray(a.transpose(2, 0, 1))
    return chw(norm(left)), chw(norm(right)), chw(cut(seg_image)), chw(cut(disp)[:, :, None].astype(np.float32))


def flip_sample(left, right, seg, disp):
    """RandomCrop's horizontal flip (util/utilTorchDataLoader.py:476-499) on CHW float arrays as prepare_sample returns them:
    -> left, right, seg, disp flipped.  The scatter `a[r, target] = a[r, c]` is a fancy-index assignment evaluated in
    row-major order: for contested targets the largest source column wins, untouched targets keep their content."""
    C, H, W = seg.shape
    d = disp[0].copy()
    sg = seg.transpose(1, 2, 0).copy()
    new_left, new_right = right[:, :, ::-1].copy(), left[:, :, ::-1].copy()
    cols = np.arange(W)[None, :].repeat(H, 0)
    tgt = (cols - d).astype(np.int64)          # float64 difference truncated towards zero
    tgt[tgt < 0] = 0
    d_new, s_new = d.copy(), sg.copy()
    for r in range(H):
        for c in range(W):                     # increasing c: later (larger) columns overwrite
            d_new[r, tgt[r, c]] = d[r, c]
            s_new[r, tgt[r, c], :] = sg[r, c, :]
    d_new[:, -10:] = 0
    s_new[:, -20:, :] = 0
    mask = (d_new == 0).astype(np.float32)
    s_new[:, :, -1] = mask
    s_new[:, :, :-1] *= (1 - mask[:, :, None])
    return new_left, new_right, np.ascontiguousarray(s_new[:, ::-1, :].transpose(2, 0, 1)), np.ascontiguousarray(d_new[None, :, ::-1])


def slice_and_switch(left, right, seg, disp, roll):
    """`sliceandSwitch` of RandomCrop (util/utilTorchDataLoader.py:455-467) on CHW outputs: rows [roll:] first, then [:roll]."""
    if roll <= 0:
        return left, right, seg, disp
    sw = lambda a: np.ascontiguousarray(np.concatenate((a[:, roll:], a[:, :roll]), axis=1))
    return sw(left), sw(right), sw(seg), sw(disp)


def double_left(left, right, seg, disp):
    """`augment_DoubleLeftImg` (util/utilTorchDataLoader.py:469-474) on CHW outputs."""
    fl = np.ascontiguousarray(left[:, :, ::-1])
    return fl, fl.copy(), np.ascontiguousarray(seg[:, :, ::-1]), (np.zeros_like(disp) + 0.0001).astype(disp.dtype)
