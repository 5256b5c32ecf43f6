"""CPU: the C-ABI library loads and exports every symbol include/pysparse_hip.h declares,
and the product fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pysparse_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psp_[a-z0-9_]+)\s*\(", src)) - {"psp_host_apply_fn"})


def test_header_symbols_are_exported():
    from pysparse_amd import _capi
    L = _capi.lib()
    names = declared_symbols()
    assert len(names) > 50
    for name in names:
        assert hasattr(L, name), name
    assert sorted(_capi.SYMBOLS) == names
    assert b"gfx950" in L.psp_version()


def test_library_was_built_from_the_sources_on_disk():
    """psp_build_id() is the hash the build stamped into the binary; __graft_entry__.source_hash() hashes the HIP
    sources, headers and flags on disk.  Equal <=> the prebuilt .so that travels to the GPU box is these sources."""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    from pysparse_amd import _capi
    assert _capi.lib().psp_build_id().decode() == G.source_hash()


@pytest.mark.gpu
def test_library_on_the_gpu_box_was_built_from_the_sources_on_disk():
    """the same check where it matters: under `-m gpu` on the GPU box, where the .so is a prebuilt file that travelled
    with the snapshot -- the library the GPU tests load must answer the hash of the sources lying next to it."""
    test_library_was_built_from_the_sources_on_disk()


def test_no_cpu_fallback_without_gpu():
    from pysparse_amd import _capi, device
    if device.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    with pytest.raises(_capi.PspError, match="no HIP device"):
        device.DeviceCSR.poisson(4, 4)
    with pytest.raises(_capi.PspError, match="no HIP device"):
        device.DeviceCSR.from_arrays((1, 1), np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32),
                                     np.array([1.0]))
