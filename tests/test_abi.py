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
