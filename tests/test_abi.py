"""The C-ABI library loads and exports every symbol include/sdhip.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sdhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "pmt_learning_for_semantic_segmentation_and_disparity_amd", "libsdhip.so"))
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libsdhip.so does not export %s" % n


def test_ctypes_signatures_cover_the_header():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import _lib
    bound = set(_lib.SIGNATURES) | {"sdhip_abi_version", "sdhip_last_error", "sdhip_conv_packed_elems", "sdhip_lovasz_workspace_bytes", "sdhip_diag_reload", "sdhip_abort_capture", "sdhip_flip_sample_workspace_bytes", "sdhip_graph_node_counts", "sdhip_softargmin_bwd_workspace_floats"}
    assert set(_declared()) == bound, set(_declared()) ^ bound


def test_abi_version_and_error_channel():
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import _lib
    assert _lib.abi_version() >= 1
    # argument validation happens before any GPU work: a bad call returns an error code and a message, never aborts
    rc = _lib._lib.sdhip_corr_fwd(None, None, None, 1, 1, 1, 1, 1, 1, 17, 1, 17, 0, None)
    assert rc < 0 and b"null" in _lib._lib.sdhip_last_error()


def test_product_has_no_cpu_path():
    import pytest
    import torch
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import nn as N, SdhipError
    with pytest.raises(SdhipError):
        N.SpatialCorrelationSampler(1, (1, 17))(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4))


def test_phase_pack_rows_tile_the_eight_sub_kernels():
    """Host logic of the sub-pixel phases of a 3x3x3 stride-2 (transposed) convolution (ops._phase_rows: the descriptor rows the
    step's batched pack launch replays): the 27 single-tap rows read every tap of the parameter exactly once, in the order of
    ops._PHASE_INDEX, and their destinations tile the eight packed sub-kernels [depth tap][tap][Mpad][64] without gaps."""
    import torch
    from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops, _lib
    w = torch.zeros(64, 32, 3, 3, 3)                       # ConvTranspose3d(64, 32) / the adjoint view of Conv3d(32 -> 64, stride 2)
    rows, spans = ops._phase_rows(w, _lib.BF16)
    assert [r[0] for r in rows] == ops._PHASE_INDEX and sorted(r[0] for r in rows) == list(range(27))
    blk = _lib.packed_elems(32, 64, 1, _lib.BF16)          # one tap: [Mpad = 32][64]
    assert blk == 32 * 64
    assert sorted(r[1] for r in rows) == [i * blk for i in range(27)]
    assert all(r[2:] == (32, 64, 1, 27, 32 * 27, 0) for r in rows)
    assert [n for _, n in spans] == [blk * (1 + pd) * (1 + ph) * (1 + pw) for (pd, ph, pw) in ops._PHASES]
    assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(7))
