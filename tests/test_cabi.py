"""The C-ABI library loads and exports every symbol include/satrender.h declares (no compute calls, CPU only)."""
import ctypes
import os
import re

import pytest

from satnerf_amd import _lib, packing

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "satrender.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def handle():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib.lib()


def test_header_and_binding_agree():
    names = declared_symbols()
    assert len(names) >= 14
    assert sorted(_lib.SIGNATURES) == names


def test_every_declared_symbol_is_exported(handle):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert getattr(raw, name) is not None, name


def test_version_and_stream_geometry(handle):
    assert handle.sr_version() == 100
    for tau in (4, 16):
        m = packing.forward_maps(256, tau)
        assert handle.sr_fwd_stream_elems(256, tau) == m["idx"].size == m["scale"].size
    for tau in (4, 16):  # the 512-wide build of the forward kernel (inference): same packer, its own stream geometry
        m = packing.forward_maps(512, tau)
        assert handle.sr_fwd_stream_elems(512, tau) == m["idx"].size
    for feat, tau in ((256, 4), (256, 16), (512, 4), (512, 16)):  # the dX stream, both widths
        assert handle.sr_bwd_stream_elems(feat, tau) == packing.backward_maps(feat, tau)["idx"].size
    assert handle.sr_fwd_stream_elems(384, 4) == -1 and handle.sr_bwd_stream_elems(384, 4) == -1
    assert handle.sr_fwd_stream_elems(256, 25) == -1
    assert handle.sr_act_elems_per_tile(256, 16) == 186 * 512 and handle.sr_dpre_elems_per_tile(256, 16) == 186 * 512
    assert handle.sr_act_elems_per_tile(256, 8) == 95 * 512 and handle.sr_dpre_elems_per_tile(256, 8) == 101 * 512  # the 8-bit workspaces
    assert handle.sr_act_elems_per_tile(512, 8) == packing.act8_units(2, 512) * 512  # 512 wide: 8-bit workspaces only
    assert handle.sr_dpre_elems_per_tile(512, 8) == packing.dpre8_units(512) * 512 and handle.sr_act_elems_per_tile(512, 16) == -1
    assert handle.sr_act_elems_per_tile(256, 4) == -1 and handle.sr_wgrad8_load_ints() == packing.WG8_LOAD_INTS


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SatRenderError):
        _lib.lib()


def test_wgrad_plan_host_only(handle):
    """sr_wgrad_plan is host logic: slices >= 1 per block, <= tiles, consecutive numbering, at most n_wg in total."""
    import ctypes

    import numpy as np

    from satnerf_amd import packing

    blocks0 = packing.backward_maps(256, 4)["blocks"]
    cases = ((65536, 256), (96 * 64, 256), (32, 256), (65536, 64), (1 << 20, 256), (65536, 3))
    for (n_points, n_wg), fmt in [(c, f) for c in cases for f in (16, 8)]:
        blocks = np.ascontiguousarray(blocks0.copy())
        n = ctypes.c_int(0)
        rc = handle.sr_wgrad_plan(blocks.ctypes.data_as(ctypes.c_void_p), blocks.shape[0], n_points, n_wg, fmt, ctypes.byref(n))
        assert rc == 0
        ns, first = blocks[:, 9], blocks[:, 10]
        tiles = (n_points + 31) // 32
        assert (ns >= 1).all() and (ns <= tiles).all()
        assert (first == np.concatenate([[0], np.cumsum(ns)[:-1]])).all() and n.value == ns.sum()
        span = int(blocks[0, 11])
        if span > 0:   # stream-K (fmt 8, uneven equal split): spans of `span` tile units over <= n_wg workgroups; slices = the (block,
            assert fmt == 8 and -(-blocks.shape[0] * tiles // span) <= n_wg   # workgroup) pairs that meet
            assert n.value <= n_wg + blocks.shape[0] - 1
            w_first, w_last = (np.arange(blocks.shape[0]) * tiles) // span, ((np.arange(blocks.shape[0]) + 1) * tiles - 1) // span
            assert (ns == w_last - w_first + 1).all()
        else:
            assert n.value <= max(n_wg, blocks.shape[0])
        if tiles >= 1024 and n_wg >= 2 * blocks.shape[0]:
            assert n.value > n_wg - blocks.shape[0]  # the launch fills the chip
        assert (blocks[:, :9] == blocks0[:, :9]).all()
        if fmt == 16:
            assert (ns == ns[0]).all()  # equal slices
        elif n_points == 65536 and n_wg == 256:
            # every workgroup used; the default 4-wave kernel (wgrad9.hip) runs one stream for every FULL block: their slices differ by at
            # most one; r06: the two one-row blocks (12, 13) run thin streams and are handed workgroups in proportion to their cost per tile
            # (the cost-weighted split of the r02 kernel is selected together with it, SATNERF_WGRAD_V1=1)
            thin = sorted(packing.wgrad9_thin_blocks(256, 4))
            full = np.setdiff1d(np.arange(blocks.shape[0]), thin)
            assert n.value == 256 and ns[full].max() - ns[full].min() <= 1
            assert thin == [12, 13] and (ns[thin] < ns[full].min()).all() and (ns[thin] >= 4).all()
            worst_full = -(-tiles // ns[full].min())
            assert all(0.25 * worst_full <= -(-tiles // ns[b]) * c <= 1.05 * worst_full for b, c in zip(thin, (0.5, 0.4)))
    # width 512 (47 blocks, five of them thin): the cost-weighted split would leave 410-tile slices beside 342-tile ones, a stream-K plan gives
    # every workgroup 376 tile units -- the plan takes whichever makespan is shorter (r06), here stream-K; SATNERF_WGRAD_STREAMK=0 forces weights
    b512 = np.ascontiguousarray(packing.backward_maps(512, 4)["blocks"].copy())
    n = ctypes.c_int(0)
    assert handle.sr_wgrad_plan(b512.ctypes.data_as(ctypes.c_void_p), b512.shape[0], 65536, 256, 8, ctypes.byref(n)) == 0
    assert int(b512[0, 11]) == -(-47 * 2048 // 256) and n.value <= 256 + 46
    os.environ["SATNERF_WGRAD_STREAMK"] = "0"
    try:
        b512w = np.ascontiguousarray(packing.backward_maps(512, 4)["blocks"].copy())
        assert handle.sr_wgrad_plan(b512w.ctypes.data_as(ctypes.c_void_p), b512w.shape[0], 65536, 256, 8, ctypes.byref(n)) == 0
        thin512 = sorted(packing.wgrad9_thin_blocks(512, 4))
        full512 = np.setdiff1d(np.arange(47), thin512)
        assert int(b512w[0, 11]) == 0 and n.value == 256 and (b512w[thin512, 9] <= b512w[full512, 9].min()).all()
    finally:
        del os.environ["SATNERF_WGRAD_STREAMK"]
    bad = np.ascontiguousarray(blocks0.copy())
    bad[0, 1] = 17
    assert handle.sr_wgrad_plan(bad.ctypes.data_as(ctypes.c_void_p), bad.shape[0], 65536, 256, 16, ctypes.byref(n)) != 0
