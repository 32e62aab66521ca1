"""The C-ABI library loads and exports every symbol include/sph3d.h declares; host-side argument
validation answers without touching a GPU (no compute launches here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sph3d.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph3d_[a-z0-9_]+)\s*\(", text)))


def test_header_cites_reference():
    text = open(HEADER).read()
    for cite in ["tf_nnquery_gpu.cu", "tf_buildkernel_gpu.cu", "tf_conv3d_gpu.cu", "tf_pool3d_gpu.cu",
                 "tf_unpool3d_gpu.cu", "tf_sample_gpu.cu"]:
        assert cite in text


def test_library_exports_every_declared_symbol():
    from sph3d_gcn_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libsph3d.so first (__graft_entry__.build())"
    l = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(l, n), "libsph3d.so does not export %s" % n
    # and the Python binding table covers exactly the header
    assert set(_lib.SIGNATURES) >= set(names)


def test_host_side_validation_without_gpu():
    from sph3d_gcn_amd import _lib
    l = _lib.lib()
    assert l.sph3d_abi_version() == 2 == _lib.ABI_VERSION
    # n must be > 2 and even (tf_buildkernel.cpp:43): rejected before any launch
    rc = l.sph3d_spherical_kernel(1, 4, 4, 4, 3, 2, 2, 0.1, None, None, None, None, None, None, None)
    assert rc == -1 and b"n_" in l.sph3d_last_error()
    rc = l.sph3d_build_sphere_neighbor(1, 4, 4, 4, -0.5, None, None, None, None, None, None)
    assert rc == -1 and b"radius>0" in l.sph3d_last_error()
    rc = l.sph3d_build_sphere_neighbor(1, 4, 4, 0, 0.5, None, None, None, None, None, None)
    assert rc == -1 and b"nn_sample>0" in l.sph3d_last_error()
    rc = l.sph3d_farthest_point_sample(1, 4, 0, None, None, None, 0, None)
    assert rc == -1
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_search_entry_points_with_caller_workspace():
    """the `_ws` twins of the neighbour-search entry points (no library allocation: SURVEY 8b) validate like the convenience
    ones, size their workspace on the host, and the scratch-release hooks answer without a device buffer to free"""
    from sph3d_gcn_amd import _lib
    l = _lib.lib()
    for name in ("sph3d_build_sphere_neighbor_workspace", "sph3d_build_sphere_neighbor_ws", "sph3d_build_sphere_neighbor_fixed_ws",
                 "sph3d_build_sphere_graph_ws", "sph3d_release_stream_scratch", "sph3d_release_all_scratch"):
        assert name in _declared()
    rc = l.sph3d_build_sphere_neighbor_ws(1, 4, 4, 4, -0.5, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"radius>0" in l.sph3d_last_error()
    rc = l.sph3d_build_sphere_graph_ws(1, 4, 4, 4, 0.1, 3, 2, 2, 1, None, None, None, None, None, 1, None, 0, None, 0, None)
    assert rc == -1 and b"n_" in l.sph3d_last_error()
    # shapes the cell grid never takes need no workspace; the bench's level 0 does: 16 B per point + 4 B per query + cell starts
    assert l.sph3d_build_sphere_neighbor_workspace(2, 512, 512) == 0
    need = l.sph3d_build_sphere_neighbor_workspace(16, 8192, 8192)
    assert need >= 16 * 8192 * (16 + 4) and need % 4 == 0
    assert l.sph3d_build_sphere_neighbor_workspace(16, 8192, 2048) < need


def test_round6_entry_points_validate_on_the_host():
    """the entry points added in round 6 answer their shape questions and reject bad requests before any launch"""
    import ctypes
    import torch
    from sph3d_gcn_amd import _lib, _tgraph
    l = _lib.lib()
    # training-mode one-kernel layer: C <= 128, C * r <= 256, Cout a power of two <= 256; statistics rows per launch
    assert l.sph3d_separable_conv3d_train_supported(8192, 33, 128, 2, 64, 128) == 1
    assert l.sph3d_separable_conv3d_train_supported(8192, 33, 256, 2, 64, 128) == 0
    assert l.sph3d_separable_conv3d_train_supported(8192, 33, 64, 2, 64, 96) == 0
    assert l.sph3d_separable_conv3d_train_blocks(128) == 512 and l.sph3d_separable_conv3d_train_blocks(256) == 256
    assert l.sph3d_separable_conv3d_train_blocks(96) == 0
    rc = l.sph3d_separable_conv3d_train(1, 64, 64, 33, 256, 2, 16, 128, None, None, None, None, None, None, None, None, None, None, None)
    assert rc == -4 and b"not covered" in l.sph3d_last_error()               # SPH3D_EUNSUPPORTED
    # product mode: query (-1) leaves it alone; the default is the split-bf16 form unless the environment says otherwise
    cur = l.sph3d_pointwise_gemm_mode(-1)
    assert cur in (0, 1) and l.sph3d_pointwise_gemm_mode(-1) == cur
    # packed transposed-graph entries need an un-weighted graph with K <= 255 (checked after the workspace)
    ws = ctypes.create_string_buffer(l.sph3d_graph_transpose_workspace(1, 8, 8, 300, 1))
    rc = l.sph3d_graph_transpose_finish(1, 8, 8, 300, 1, None, None, None, None, None, None, None, None, ctypes.cast(ws, ctypes.c_void_p),
                                        len(ws), None)
    assert rc == -1 and b"packed entries" in l.sph3d_last_error()
    # decoding of packed entries (tests / tools): row in the low 24 bits, 1 / count from the byte above
    key = torch.tensor([5 | (4 << 24), 70000 | (64 << 24), 0], dtype=torch.int32)
    k, s = _tgraph.entries((None, key, None, None))
    assert k.tolist() == [5, 70000, 0] and s.tolist() == [0.25, 1.0 / 64, 0.0]
    k2, s2 = _tgraph.entries((None, key, torch.ones(3), None))
    assert k2 is key and s2.tolist() == [1.0, 1.0, 1.0]


def test_library_default_bins_are_the_reference_builds():
    """tf_buildkernel's library-wide default is the device-library atan2f (= the reference build, bit for bit); the tests run in
    "shared" mode through conftest's fixture"""
    from sph3d_gcn_amd import tf_buildkernel
    assert tf_buildkernel.DEFAULT_ATAN2 == "ocml"
    assert tf_buildkernel._atan2 == "shared"          # this test runs under the fixture
    import subprocess, sys
    out = subprocess.run([sys.executable, "-c", "from sph3d_gcn_amd import tf_buildkernel as t; print(t._atan2)"], cwd=ROOT,
                         capture_output=True, text=True)
    assert out.stdout.strip() == "ocml", out.stderr[-400:]


def test_ops_refuse_cpu_tensors():
    import torch
    from sph3d_gcn_amd import tf_sample, tf_nnquery, _lib
    with pytest.raises(_lib.Sph3dError):
        tf_sample.farthest_point_sample(4, torch.zeros(1, 8, 3))
    with pytest.raises(_lib.Sph3dError):
        tf_nnquery.build_sphere_neighbor(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3), 0.1, None, 4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sph3d_gcn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
