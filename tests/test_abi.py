"""CPU: libvtaco_hip.so loads and exports every symbol include/vtaco_hip.h declares, and
the ctypes signature table covers exactly that set (no compute calls: no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    text = open(os.path.join(ROOT, "include", "vtaco_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    so = os.path.join(ROOT, "vtaco_amd", "libvtaco_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vtaco_amd", "csrc"), "-j4"])
    return so


def test_header_symbols_exported_and_bound(built):
    from vtaco_amd import _lib
    lib = _lib.load()
    names = declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in vtaco_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.vt_abi_version() == 1


def test_host_only_queries(built):
    from vtaco_amd import _lib
    lib = _lib.load()
    assert lib.vt_decoder_blob_bytes(32, 32, 5) > 60000
    assert lib.vt_decoder_blob_bytes(256, 128, 5) == 0          # unsupported shape -> 0, ops raises
    assert lib.vt_mc_workspace_bytes(1, 4, 4) == 0
    assert lib.vt_mc_workspace_bytes(128, 128, 128) > 127 ** 3 * 18


def test_no_cpu_fallback():
    import torch
    from vtaco_amd._lib import VtError
    from vtaco_amd.conv_onet.models import decoder_dict
    dec = decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32)
    with pytest.raises(VtError):
        dec(torch.zeros(1, 4, 3), {"grid": torch.zeros(1, 32, 4, 4, 4)})
