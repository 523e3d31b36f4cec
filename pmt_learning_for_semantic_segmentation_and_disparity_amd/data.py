"""GPU-side sample preparation.

Counterpart of `CustomDataset.__getitem__` + `RandomCrop` + `ToTensor` (util/utilTorchDataLoader.py:133-274,
348-463,608-630) and `readPFM` (util/utilIOPfm.py:66-101).  The host keeps what only a host can do — open files,
unpack the PNG containers (PIL), read the PFM header, draw the crop offsets from torch's CPU generator in the
reference's order — and hands the raw bytes to `sdhip_prepare_sample`, which writes one slot of the batch tensors
directly in the layout the network consumes (channels-last, optionally bf16).

Augmentations of RandomCrop that are plain numpy / torch-RNG logic upstream run on the device, bit-exact against samples
the reference's own loader produced: crop (random, `is_down`, kitti's lower band), the cityscapes flip-with-disparity-shift
(:476-499), `sliceandSwitch` (:455-467) and `augment_DoubleLeftImg` (:469-474); the draw_* helpers consume torch's CPU
generator in the reference's order.  Not reproduced, because the libraries that define their arithmetic are not in this
image and their results can therefore not be pinned: the `cv2.resize` scale augmentation (:411-433, cv2), `cropPerson`
(:528+, skimage.measure.label) and the colour jitter (:232-244,276-300: torchvision.transforms.functional on PIL images +
PIL GaussianBlur).
"""
import ctypes
import re

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

SEG_THRESHOLD, SEG_ID_PLUS_ONE, SEG_LUT = 0, 1, 2
DEPTH_PFM, DEPTH_U16 = 0, 1
ACTIVATIONS = {"linear": 0, "sigmoid": 1, "tanh": 2}

# id -> trainId of the public Cityscapes label table (the `id2label[i].trainId` that util/utilCityscape.py:173-186 looks
# up); ids 0..33, everything else is "ignore" (255).  tests/test_data.py checks it against the reference's table.
_CITYSCAPES_TRAIN_ID = {7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13,
                        27: 14, 28: 15, 31: 16, 32: 17, 33: 18}


def cityscapes_lut(n_labels=19):
    """uint8[256]: id -> one-hot channel; ids without a trainId go to the extra channel `n_labels` (utilCityscape.py:183-185)."""
    lut = np.full(256, n_labels, dtype=np.uint8)
    for i, t in _CITYSCAPES_TRAIN_ID.items():
        lut[i] = t
    return lut


def parse_pfm_header(buf):
    """(color, width, height, scale, little_endian, payload_offset) of a PFM file held in `buf` (bytes-like);
    same acceptance rules and errors as readPFM (util/utilIOPfm.py:66-96)."""
    mv = bytes(buf[:256])
    lines, pos = [], 0
    for _ in range(3):
        end = mv.find(b"\n", pos)
        if end < 0:
            raise Exception("Malformed PFM header.")
        lines.append(mv[pos:end + 1])
        pos = end + 1
    tag = lines[0].rstrip().decode("ascii")
    if tag == "PF":
        color = True
    elif tag == "Pf":
        color = False
    else:
        raise Exception("Not a PFM file.")
    m = re.match(r"^(\d+)\s(\d+)\s$", lines[1].decode("ascii"))
    if not m:
        raise Exception("Malformed PFM header.")
    width, height = map(int, m.groups())
    scale = float(lines[2].decode("ascii").rstrip())
    little = scale < 0
    return color, width, height, abs(scale), little, pos


def draw_crop(h, w, output_size, dataset_name="roses", is_down=False):
    """Crop offsets as RandomCrop.__call__ draws them on its plain path (no resize, no focusPerson):
    util/utilTorchDataLoader.py:435-452.  Consumes torch's CPU generator in the reference's order — one
    `multinomial([0.2, 0.8])` (evaluated for every dataset, only kitti uses it), then `randint` for the top and the left
    offset — so the same seed yields the same crops.  Returns (top, left, new_h, new_w); output_size[0] == 0 keeps the
    whole image and draws nothing (:394-409)."""
    new_h, new_w = (output_size, output_size) if isinstance(output_size, int) else output_size
    if new_h == 0:
        return 0, 0, h, w
    if is_down:
        return h - new_h, (w - new_w) // 2, new_h, new_w
    if torch.multinomial(torch.tensor([0.2, 0.8]), 1).item() and dataset_name == "kitti":
        y_start = max(h - new_h - 100, 0)
    else:
        y_start = 0
    top = int(torch.randint(y_start, h - new_h + 1, (1,)))
    left = int(torch.randint(0, w - new_w + 1, (1,)))
    return top, left, new_h, new_w


def draw_slice_and_switch(full_rows, out_rows, enabled=True):
    """`sliceandSwitch` (util/utilTorchDataLoader.py:455-467): one `randint(2, 6)` after the crop offsets; the cut row is
    int(ROWS OF THE UNCROPPED IMAGE / divisor) — `w, h, c = images[0].shape` is read before the crop loop — and it is applied
    to the cropped maps: a cut at or beyond their last row leaves them unchanged.  Returns the row_roll of prepare_into."""
    if not enabled:
        return 0
    divisor = float(torch.randint(2, 6, (1,)))
    cut = int(full_rows / divisor)
    return cut if cut < out_rows else 0


def draw_double_left(enabled=True):
    """`augment_DoubleLeftImg and multinomial([0.9, 0.1])` (:469): drawn only when the option is on, after sliceandSwitch."""
    return bool(enabled and torch.multinomial(torch.tensor([0.9, 0.1]), 1).item())


def draw_flip(dataset_name, flip_horizontal=True):
    """The flip decision as RandomCrop draws it (util/utilTorchDataLoader.py:476): `flipHorizontal and multinomial([.5,.5])
    and datasetName == 'cityscapes'` — the draw happens whenever the option is on (short-circuit order), after the crop
    offsets of draw_crop; only cityscapes samples are actually flipped."""
    if not flip_horizontal:
        return False
    hit = bool(torch.multinomial(torch.tensor([0.5, 0.5]), 1).item())
    return hit and dataset_name == "cityscapes"


class SamplePreparer:
    """Fills batch tensors sample by sample on the GPU.

    dataset_name       'roses' | 'garden' (PFM depth, f = 640, b = 0.03 — CustomDataset.__init__ :57-58) or
                       'kitti' | 'cityscapes' (16-bit disparity PNG / 256, id -> trainId one-hot with an ignore channel)
    n_labels, max_d, output_activation, normalize   as CustomDataset's arguments; normalize = [[mean x3], [std x3]]
    """

    def __init__(self, dataset_name, n_labels, max_d, output_activation="linear", normalize=((0., 0., 0.), (1., 1., 1.)),
                 dtype=torch.float32, device="cuda", f=640, b=0.03):
        if not torch.cuda.is_available():
            raise _lib.SdhipError("SamplePreparer needs a GPU; there is no CPU path")
        if output_activation not in ACTIVATIONS:
            raise _lib.SdhipError("unknown output activation %r" % (output_activation,))
        if dataset_name not in ("roses", "garden", "kitti", "cityscapes"):
            raise _lib.SdhipError("unknown dataset %r" % (dataset_name,))
        self.dataset_name, self.n_labels, self.max_d = dataset_name, int(n_labels), float(max_d)
        self.activation = ACTIVATIONS[output_activation]
        self.dtype, self.device = dtype, torch.device(device)
        self.fb = float(np.float32(f * b * 1))
        self.mean = (ctypes.c_float * 3)(*[float(v) for v in normalize[0]])
        self.std = (ctypes.c_float * 3)(*[float(v) for v in normalize[1]])
        self.lut = None
        if dataset_name in ("kitti", "cityscapes"):
            self.seg_mode, self.n_seg, self.depth_mode = SEG_LUT, self.n_labels + 1, DEPTH_U16
            self.lut = torch.from_numpy(cityscapes_lut(self.n_labels)).to(self.device)
        elif dataset_name == "roses":
            self.seg_mode, self.n_seg, self.depth_mode = SEG_THRESHOLD, self.n_labels, DEPTH_PFM
        else:
            self.seg_mode, self.n_seg, self.depth_mode = SEG_ID_PLUS_ONE, self.n_labels, DEPTH_PFM
        self._keep = []

    def alloc_batch(self, B, h, w):
        """left, right (B,3,h,w) of `dtype`; seg (B,n_seg,h,w) f32; disp (B,1,h,w) f32 — logical NCHW, channels-last memory."""
        dev = self.device
        mk = lambda c, dt: torch.empty((B, h, w, c), dtype=dt, device=dev).permute(0, 3, 1, 2)
        return mk(3, self.dtype), mk(3, self.dtype), mk(self.n_seg, torch.float32), mk(1, torch.float32)

    def _dev(self, a):
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:          # PIL hands out read-only views; torch wants to own a writable buffer
            a = a.copy()
        t = torch.from_numpy(a).to(self.device, non_blocking=True)
        self._keep.append(t)
        return t

    def prepare_into(self, batch, b, left, right, seg, depth, crop=None, row_roll=0):
        """left/right: uint8 (H,W,>=3) arrays; seg: uint8 (H,W[,C]); depth: the raw bytes of a .pfm file (roses/garden)
        or a uint16 (H,W) array (kitti/cityscapes); crop = (top, left, h, w) or None for the whole image; row_roll: the
        sliceandSwitch cut (draw_slice_and_switch).  Writes slot `b` of batch = (left, right, seg, disp) as returned by
        alloc_batch.  Asynchronous."""
        bl, br, bs, bd = batch
        H, W = left.shape[:2]
        top, lft, oh, ow = crop if crop is not None else (0, 0, H, W)
        if tuple(bl.shape[2:]) != (oh, ow):
            raise _lib.SdhipError("batch tensors are %s, the crop is %dx%d" % (tuple(bl.shape[2:]), oh, ow))
        if right.shape != left.shape or left.dtype != np.uint8 or right.dtype != np.uint8 or left.ndim != 3 or left.shape[2] < 3:
            raise _lib.SdhipError("left/right must be uint8 (H,W,>=3) arrays of one shape")
        if seg.dtype != np.uint8 or seg.shape[:2] != (H, W):
            raise _lib.SdhipError("seg must be a uint8 (H,W[,C]) array matching the images")
        seg3 = seg if seg.ndim == 3 else seg[:, :, None]
        seg_channel = 2 if self.dataset_name == "roses" else 0       # `seg_binary[:,:,2]` (:206)
        if seg_channel >= seg3.shape[2]:
            raise _lib.SdhipError("roses segmentation images need >= 3 channels")
        big = 0
        if self.depth_mode == DEPTH_PFM:
            raw = np.frombuffer(depth, dtype=np.uint8)
            color, w_, h_, _, little, off = parse_pfm_header(raw)
            if color or (h_, w_) != (H, W):
                raise _lib.SdhipError("PFM is %s %dx%d, the images are %dx%d" % ("colour" if color else "grey", h_, w_, H, W))
            if raw.size - off < H * W * 4:
                raise _lib.SdhipError("PFM payload is truncated")
            payload = raw[off:off + H * W * 4]
            # a copy re-bases the payload at an aligned address (the header length is arbitrary)
            dd, pitch, big = self._dev(payload.copy()), W * 4, 0 if little else 1
        else:
            if depth.dtype != np.uint16 or depth.shape != (H, W):
                raise _lib.SdhipError("disparity must be a uint16 (H,W) array")
            dd, pitch = self._dev(depth), W * 2
        dl, dr, ds = self._dev(left), self._dev(right), self._dev(seg3)
        cs = left.shape[2]
        sl = lambda t: t[b]
        ol, orr, os_, od = sl(bl), sl(br), sl(bs), sl(bd)
        ld_img, ld_seg = bl.stride(3), bs.stride(3)
        if bl.stride(1) != 1 or bs.stride(1) != 1 or not od.is_contiguous():
            raise _lib.SdhipError("batch tensors must be channels-last (use alloc_batch)")
        call("sdhip_prepare_sample", ptr(dl), ptr(dr), W * cs, cs, ptr(ds), W * seg3.shape[2], seg3.shape[2], seg_channel,
             self.seg_mode, 128, ptr(self.lut), ptr(dd), pitch, self.depth_mode, big, H, W, top, lft, oh, ow, int(row_roll), self.fb, self.max_d,
             self.activation, ctypes.cast(self.mean, ctypes.c_void_p), ctypes.cast(self.std, ctypes.c_void_p), ptr(ol), ptr(orr),
             ld_img, ptr(os_), ld_seg, self.n_seg, ptr(od), _lib.dtype_code(bl), stream_ptr())

    def flip_slot(self, batch, b):
        """Horizontal flip of slot `b` in place (RandomCrop(flipHorizontal=True), util/utilTorchDataLoader.py:476-499): call it
        after prepare_into when data.draw_flip said so.  Asynchronous."""
        bl, br, bs, bd = batch
        H, W = bl.shape[2], bl.shape[3]
        nbytes = _lib._lib.sdhip_flip_sample_workspace_bytes(H, W, self.n_seg)
        ws = getattr(self, "_flip_ws", None)
        if ws is None or ws.numel() < nbytes:
            ws = self._flip_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        call("sdhip_flip_sample", ptr(bl[b]), ptr(br[b]), bl.stride(3), ptr(bs[b]), bs.stride(3), self.n_seg, ptr(bd[b]), H, W,
             ptr(ws), ws.numel(), _lib.dtype_code(bl), stream_ptr())

    def double_left_slot(self, batch, b):
        """augment_DoubleLeftImg on slot `b`, in place (call it after prepare_into when draw_double_left said so)."""
        bl, br, bs, bd = batch
        call("sdhip_double_left_sample", ptr(bl[b]), ptr(br[b]), bl.stride(3), ptr(bs[b]), bs.stride(3), self.n_seg, ptr(bd[b]),
             bl.shape[2], bl.shape[3], _lib.dtype_code(bl), stream_ptr())

    def release(self):
        """Drop the staged uint8 inputs (call after the batch has been consumed or the stream synchronised)."""
        self._keep = []


def read_sample_files(left_path, right_path, seg_path, depth_path):
    """Host side of the loader: PNG containers through PIL (the reference uses skimage.io / PIL, :150-154), the PFM file
    as raw bytes, a 16-bit disparity PNG as a uint16 array."""
    from PIL import Image
    left = np.asarray(Image.open(left_path))[:, :, :3]
    right = np.asarray(Image.open(right_path))[:, :, :3]
    seg = np.asarray(Image.open(seg_path))
    if depth_path.lower().endswith(".pfm"):
        with open(depth_path, "rb") as f:
            depth = f.read()
    else:
        depth = np.asarray(Image.open(depth_path)).astype(np.uint16)
    return left, right, seg, depth
