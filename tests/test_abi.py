"""The C-ABI library: builds, loads, and exports every symbol include/dega_hip.h declares; argument checking and the
no-GPU behaviour (fail loudly with ERROR_LIBRARY_INIT -- there is no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from __graft_entry__ import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dca():
    mod = load_package()
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mod


def test_every_declared_symbol_is_exported(dca):
    with open(os.path.join(ROOT, "include", "dega_hip.h")) as f:
        header = f.read()
    declared = sorted(set(re.findall(r"\b(dega_hip_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 18
    lib = C.CDLL(dca.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == dca.exported_symbols()  # the Python binding covers exactly the header


def test_version_and_worst_case(dca):
    L = dca.library()
    assert b"gfx950" in L.dega_hip_version()
    for T in (0, 1, 96, 86400):
        n = L.dega_hip_worst_case_bytes(T)
        assert n % 4 == 0 and n * 8 >= T * 65 + 32


def test_no_gpu_means_library_init_error(dca):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert dca.library().dega_hip_device_count() == 0
    with pytest.raises(dca.DegaError) as e:
        dca.Context(0)
    assert e.value.code == dca.ERROR_LIBRARY_INIT


def test_null_context_is_rejected(dca):
    L = dca.library()
    assert L.dega_hip_encode_dev(None, None, 1, 1, 1, 1, 32, None, 4, None, None, None) == dca.ERROR_INVALID_VALUE
    assert L.dega_hip_profile(None, 1) == dca.ERROR_INVALID_VALUE


def test_channel_partition_of_a_group(dca):
    """The per-GPU batch split of the multi-device entry points (no GPU needed): contiguous ranges in channel order that
    cover the batch exactly, balanced, whole 512-channel units for large batches; shards of an empty or tiny batch may
    be empty.  Channels are independent units (DCLib/src/diff.c:11, bac.c:150), so this partition is the whole story."""
    import ctypes as C
    L = dca.library()
    for Cn in (0, 1, 7, 511, 512, 513, 4096, 65536, 65537, 1048576, 8388608, 1000003):
        for G in (1, 2, 3, 4, 8):
            cuts = (C.c_size_t * (G + 1))()
            assert L.dega_hip_split_channels(Cn, G, cuts) == 0
            cuts = list(cuts)
            assert cuts[0] == 0 and cuts[-1] == Cn and all(a <= b for a, b in zip(cuts, cuts[1:])), (Cn, G, cuts)
            sizes = [b - a for a, b in zip(cuts, cuts[1:])]
            if Cn >= G * 1024:
                assert all(c % 512 == 0 for c in cuts[1:-1]), (Cn, G, cuts)
                assert max(sizes) - min(sizes) <= 1024, (Cn, G, sizes)
            else:
                assert max(sizes) - min(sizes) <= 1, (Cn, G, sizes)
    assert L.dega_hip_split_channels(10, 0, None) == dca.ERROR_INVALID_VALUE


def test_group_without_gpu_fails_loudly(dca):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(dca.DegaError) as e:
        dca.Group()
    assert e.value.code == dca.ERROR_LIBRARY_INIT
    assert dca.library().dega_hip_group_size(None) == 0
