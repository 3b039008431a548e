"""CPU-side checks of the drop-in boundary: libposecnn_hip.so loads without a GPU, exports every
symbol include/posecnn_hip.h declares, and rejects bad arguments with the reference's
InvalidArgument conditions before touching the device (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "posecnn_hip.h")


@pytest.fixture(scope="module")
def L():
    from posecnn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcnn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    from posecnn_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), "libposecnn_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "posecnn_amd/_lib.py has no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version_and_status_strings(L):
    assert L.pcnn_abi_version() == 2
    assert L.pcnn_status_string(0) == b"ok"
    assert b"invalid" in L.pcnn_status_string(-1)


def test_argument_validation_happens_on_the_host(L):
    from posecnn_amd import _lib
    n = ctypes.c_size_t(0)
    # hough: skip_pixels >= 1, 2 <= num_classes <= 64
    assert L.pcnn_hough_voting_workspace_bytes(1, 480, 640, 22, -1.0, 0, 0, ctypes.byref(n)) == _lib.PCNN_EINVAL
    assert b"skip_pixels" in L.pcnn_last_error_string()
    assert L.pcnn_hough_voting_workspace_bytes(1, 480, 640, 1, -1.0, 10, 0, ctypes.byref(n)) == _lib.PCNN_EINVAL
    assert L.pcnn_hough_voting_workspace_bytes(16, 480, 640, 22, -1.0, 10, 0, ctypes.byref(n)) == 0
    small = n.value
    assert 0 < small < 64 << 20
    assert L.pcnn_hough_voting_workspace_bytes(16, 480, 640, 22, 5.0, 10, 0, ctypes.byref(n)) == 0
    assert n.value > small  # threshold_vote > 0 keeps the Hough space
    # hard_label: threshold > 0 (hard_label_op.cc:150-155)
    assert L.pcnn_hard_label_fwd(None, None, 10, 22, 0.0, None, None) == _lib.PCNN_EINVAL
    assert b"threshold > 0" in L.pcnn_last_error_string()
    # average distance: margin >= 0 (average_distance_loss_op.cc:262-267)
    assert L.pcnn_average_distance_fwd(None, None, None, None, None, 1, 22, 10, -1.0, None, None, None, None, 0, None) == _lib.PCNN_EINVAL
    assert b"margin >= 0" in L.pcnn_last_error_string()
    # roi_pool: >= 6 ROI columns
    assert L.pcnn_roi_pool_fwd(None, None, 1, 30, 40, 512, 3, 5, 7, 7, 0.0625, 0, None, None, None) == _lib.PCNN_EINVAL
    # backproject: kernel_size / threshold >= 0 (backprojecting_op.cc:303-320), 48 meta values
    assert L.pcnn_backproject_fwd(None, None, None, None, None, 1, 8, 8, 4, 3, 48, 4, -1, 0.1, None, None, None, None) == _lib.PCNN_EINVAL
    assert L.pcnn_backproject_fwd(None, None, None, None, None, 1, 8, 8, 4, 3, 12, 4, 1, 0.1, None, None, None, None) == _lib.PCNN_EINVAL
    # NULL pointers are reported, not dereferenced
    assert L.pcnn_hard_label_fwd(None, None, 10, 22, 0.5, None, None) == _lib.PCNN_ENULL


def test_ops_refuse_cpu_tensors():
    import torch
    from posecnn_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.hard_label(torch.zeros(1, 2, 2, 3), torch.zeros(1, 2, 2, dtype=torch.int32), 0.5)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.roi_pool(torch.zeros(1, 4, 4, 8), torch.zeros(1, 7), 7, 7, 0.5, 0)


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "posecnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "pcnn_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f


def test_plain_c_consumer_of_the_header_links_and_runs(L):
    """tests/capi_consumer.c includes include/posecnn_hip.h, links libposecnn_hip.so and calls the
    host-side entry points with the header's prototypes (built by __graft_entry__.build())."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "capi_consumer")
    src = os.path.join(ROOT, "tests", "capi_consumer.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(HEADER)):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "capi_consumer ok" in r.stdout


def test_library_is_loaded_after_torchs_hip_runtime():
    """Round 6: `build()` followed by `smoke()` in ONE process failed on the GPU box with "no ROCm-capable device is detected" — the
    ctypes load of libposecnn_hip.so pulled in the system's libamdhip64 before torch had mapped its own copy, and the process had two
    HIP runtimes. `_lib.lib()` now imports torch first; this holds it to that in a fresh interpreter."""
    import subprocess
    import sys
    code = ("import sys; assert 'torch' not in sys.modules; from posecnn_amd import _lib; _lib.lib(); "
            "assert 'torch' in sys.modules; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
