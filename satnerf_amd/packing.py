"""Host-side geometry of the fused-MLP weight streams (numpy mirror of ``csrc/mlp_layout.h``).

The HIP kernels consume the Sat-NeRF weights as a linear stream of 1-KiB MFMA A-fragment "pieces" in
consumption order, with the K axis of every layer permuted into the *slot* order in which the previous
layer's accumulators sit in registers.  This module builds, once per network shape, the integer gather
maps that ``sr_pack_stream`` applies to the flat fp32 parameter vector every time the weights change:

    stream[i] = bf16( flat[idx[i]] * scale[i] )          (idx < 0 -> 0)

Parameter names/shapes are those of the reference ``SatNeRF`` ``state_dict`` (models/satnerf.py:104-153).
"""
from __future__ import annotations

import functools
import os
import math
from collections import OrderedDict

import numpy as np

INV_2PI = np.float32(1.0 / (2.0 * math.pi))  # sin stages run in revolutions (v_sin_f32)
W0_FIRST = 30.0  # Siren(w0=30) after fc_net.0 only (models/satnerf.py:106)


def aux_steps(tau: int) -> int:
    return (8 + ((tau + 7) // 8) * 8 + 15) // 16


def slot_to_feat(sigma):
    """Feature held in slot ``sigma`` (= 16*kstep + 8*lane_half + element) of a B fragment -- mlp_layout.h."""
    sigma = np.asarray(sigma)
    s, h, j = sigma >> 4, (sigma >> 3) & 1, sigma & 7
    g = j + 8 * (s & 1)
    return 32 * (s >> 1) + (g & 3) + 8 * (g >> 2) + 4 * h


def satnerf_param_shapes(feat=256, tau=4, layers=8, skips=(4,)):
    """Ordered ``{state_dict key: shape}`` of SatNeRF (module registration order)."""
    half = feat // 2
    sh = OrderedDict()
    for i in range(layers):
        fan_in = 3 if i == 0 else (feat + 3 if i in skips else feat)
        sh[f"fc_net.{2 * i}.weight"] = (feat, fan_in)
        sh[f"fc_net.{2 * i}.bias"] = (feat,)
    sh["sigma_from_xyz.0.weight"] = (1, feat)
    sh["sigma_from_xyz.0.bias"] = (1,)
    sh["feats_from_xyz.weight"] = (feat, feat)
    sh["feats_from_xyz.bias"] = (feat,)
    sh["rgb_from_xyzdir.0.weight"] = (half, feat)
    sh["rgb_from_xyzdir.0.bias"] = (half,)
    sh["rgb_from_xyzdir.2.weight"] = (3, half)
    sh["rgb_from_xyzdir.2.bias"] = (3,)
    sh["sun_v_net.0.weight"] = (half, feat + 3)
    sh["sun_v_net.0.bias"] = (half,)
    for j in (2, 4):
        sh[f"sun_v_net.{j}.weight"] = (half, half)
        sh[f"sun_v_net.{j}.bias"] = (half,)
    sh["sun_v_net.6.weight"] = (1, half)
    sh["sun_v_net.6.bias"] = (1,)
    sh["sky_color.0.weight"] = (half, 3)
    sh["sky_color.0.bias"] = (half,)
    sh["sky_color.2.weight"] = (3, half)
    sh["sky_color.2.bias"] = (3,)
    sh["beta_from_xyz.0.weight"] = (half, feat + tau)
    sh["beta_from_xyz.0.bias"] = (half,)
    sh["beta_from_xyz.2.weight"] = (1, half)
    sh["beta_from_xyz.2.bias"] = (1,)
    return sh


def param_offsets(shapes):
    off, out = 0, OrderedDict()
    for k, shp in shapes.items():
        out[k] = (off, shp)
        off += int(np.prod(shp))
    return out, off


class _Mat:
    """A stage's dense matrix in slot space: rows x (K slots + 16*auxs aux slots) of (source index, scale)."""

    AUX_SUN, AUX_ONE, AUX_XYZ, AUX_T = 0, 3, 4, 8

    def __init__(self, rows, k_slots, auxs, offsets):
        self.k, self.offsets = k_slots, offsets
        self.idx = np.full((rows, k_slots + 16 * auxs), -1, np.int64)
        self.scl = np.zeros((rows, k_slots + 16 * auxs), np.float32)

    def _flat(self, name, r, c):
        off, shp = self.offsets[name]
        return off + (np.asarray(r) * shp[1] + np.asarray(c) if len(shp) == 2 else np.asarray(r))

    def put_block(self, row0, name, n_rows, slot0, n_slots, col0, scale):
        """rows row0.. <- W[name][0..n_rows, col0 + phi(local slot)] over n_slots slots starting at slot0."""
        feat = slot_to_feat(np.arange(n_slots))
        r = np.arange(n_rows)[:, None]
        self.idx[row0:row0 + n_rows, slot0:slot0 + n_slots] = self._flat(name, r, col0 + feat[None, :])
        self.scl[row0:row0 + n_rows, slot0:slot0 + n_slots] = scale

    def put_aux(self, row0, name, n_rows, aux_col0, cols, scale):
        """rows row0.. aux slots aux_col0.. <- W[name][:, cols] (2-D) or the bias vector (cols is None)."""
        r = np.arange(n_rows)
        if cols is None:
            self.idx[row0:row0 + n_rows, self.k + aux_col0] = self._flat(name, r, 0)
            self.scl[row0:row0 + n_rows, self.k + aux_col0] = scale
        else:
            cols = np.asarray(cols)
            self.idx[row0:row0 + n_rows, self.k + aux_col0:self.k + aux_col0 + len(cols)] = self._flat(name, r[:, None], cols[None, :])
            self.scl[row0:row0 + n_rows, self.k + aux_col0:self.k + aux_col0 + len(cols)] = scale


def _serialize(mat: _Mat):
    """tile -> piece (k-step) -> unit L=(h*32+r) -> element j : the order the kernel's ds_read_b128 expects."""
    rows, kt = mat.idx.shape
    nt, ns = rows // 32, kt // 16
    # [tile, r, s, h, j] -> [tile, s, h, r, j]
    def ser(a):
        return a.reshape(nt, 32, ns, 2, 8).transpose(0, 2, 3, 1, 4).reshape(-1)
    return ser(mat.idx), ser(mat.scl)


@functools.lru_cache(maxsize=8)
def forward_maps(feat=256, tau=4):
    """Gather maps of the forward stream and of the fc_net.0 table.

    Returns dict(idx int32 [n], scale fp32 [n], l0_idx int32 [feat*4], l0_scale fp32 [feat*4], n_params, auxs).
    """
    if feat not in (256, 512):
        raise ValueError(f"feat={feat} unsupported by this build (256; 512 for the inference kernel)")
    if not 1 <= tau <= 24:
        raise ValueError(f"t_embedding tau={tau} unsupported (1..24)")
    half, auxs = feat // 2, aux_steps(tau)
    offsets, n_params = param_offsets(satnerf_param_shapes(feat, tau))
    c = INV_2PI
    mats = []
    for l in range(1, 8):  # fc_net.2 .. fc_net.14
        name = f"fc_net.{2 * l}"
        m = _Mat(feat, feat, auxs, offsets)
        m.put_block(0, name + ".weight", feat, 0, feat, 3 if l == 4 else 0, c)  # skip layer input = [xyz | h]
        m.put_aux(0, name + ".bias", feat, _Mat.AUX_ONE, None, c)
        if l == 4:
            m.put_aux(0, name + ".weight", feat, _Mat.AUX_XYZ, [0, 1, 2], c)
        mats.append(m)
    g1 = _Mat(feat + 32, feat, auxs, offsets)  # feats (identity) + one tile whose row 0 is sigma
    g1.put_block(0, "feats_from_xyz.weight", feat, 0, feat, 0, 1.0)
    g1.put_aux(0, "feats_from_xyz.bias", feat, _Mat.AUX_ONE, None, 1.0)
    g1.put_block(feat, "sigma_from_xyz.0.weight", 1, 0, feat, 0, 1.0)
    g1.put_aux(feat, "sigma_from_xyz.0.bias", 1, _Mat.AUX_ONE, None, 1.0)
    mats.append(g1)
    def hidden(name, aux_col=None, aux_cols=None):  # a 128-row hidden layer on [feats; aux]
        m = _Mat(half, feat, auxs, offsets)
        m.put_block(0, name + ".weight", half, 0, feat, 0, c)
        m.put_aux(0, name + ".bias", half, _Mat.AUX_ONE, None, c)
        if aux_col is not None:
            m.put_aux(0, name + ".weight", half, aux_col, aux_cols, c)
        return m

    def head(name, row0, n_rows, with_bias):  # rows of the 5-row output tile fed by one hidden vector
        m = _Mat(32, half, auxs if with_bias else 0, offsets)
        m.put_block(row0, name + ".weight", n_rows, 0, half, 0, 1.0)
        return m

    mats.append(hidden("rgb_from_xyzdir.0"))
    mats.append(head("rgb_from_xyzdir.2", 0, 3, False))  # rows 0..2: albedo logits
    mats.append(hidden("sun_v_net.0", _Mat.AUX_SUN, [feat, feat + 1, feat + 2]))  # cat([feats, sun]) :199
    for j in (2, 4):  # sun_v_net.2, sun_v_net.4
        m = _Mat(half, half, auxs, offsets)
        m.put_block(0, f"sun_v_net.{j}.weight", half, 0, half, 0, c)
        m.put_aux(0, f"sun_v_net.{j}.bias", half, _Mat.AUX_ONE, None, c)
        mats.append(m)
    mats.append(head("sun_v_net.6", 3, 1, False))  # row 3: sun-visibility logit
    mats.append(hidden("beta_from_xyz.0", _Mat.AUX_T, [feat + i for i in range(tau)]))  # cat([feats, t]) :204
    hb = head("beta_from_xyz.2", 4, 1, True)  # row 4: beta pre-softplus; the aux k-step carries all five biases
    hb.put_aux(0, "rgb_from_xyzdir.2.bias", 3, _Mat.AUX_ONE, None, 1.0)
    hb.put_aux(3, "sun_v_net.6.bias", 1, _Mat.AUX_ONE, None, 1.0)
    hb.put_aux(4, "beta_from_xyz.2.bias", 1, _Mat.AUX_ONE, None, 1.0)
    mats.append(hb)
    parts = [_serialize(m) for m in mats]
    idx = np.concatenate([p[0] for p in parts]).astype(np.int32)
    scale = np.concatenate([p[1] for p in parts]).astype(np.float32)
    # fc_net.0 table in slot order: rows [w_x, w_y, w_z, b] * 30/(2 pi)
    feat_of_slot = slot_to_feat(np.arange(feat))
    w_off, _ = offsets["fc_net.0.weight"]
    b_off, _ = offsets["fc_net.0.bias"]
    l0_idx = np.stack([w_off + 3 * feat_of_slot, w_off + 3 * feat_of_slot + 1, w_off + 3 * feat_of_slot + 2, b_off + feat_of_slot], 1)
    l0_scale = np.full(l0_idx.shape, np.float32(W0_FIRST) * INV_2PI, np.float32)
    return dict(idx=idx, scale=scale, l0_idx=l0_idx.reshape(-1).astype(np.int32), l0_scale=l0_scale.reshape(-1),
                n_params=n_params, auxs=auxs, offsets=offsets)


# ------------------------------------------------------------------------------------------------ backward
KIND_BF16, KIND_PHASE = 0, 1
WG_BLOCK_FLOATS = 256 * 256 + 256 * 32  # ... plus its 32 aux columns


def feat_to_slot(n):
    """Inverse of slot_to_feat over n features (n a multiple of 32)."""
    inv = np.empty(n, np.int64)
    inv[slot_to_feat(np.arange(n))] = np.arange(n)
    return inv


WG_TABLE_INTS = 12  # ints per job block: rf0 nr0 rf1 nr1 | cf0 nc0 cf1 nc1 | col_kind n_slices first_slice 0


class _Blocks:
    """Job table of the weight-gradient kernel.  A block = up to 16 dpre ROW fragments x up to 16 activation COLUMN fragments
    (each given as at most two contiguous fragment ranges) + the rows x aux-slot columns; a workgroup's time per point tile does
    not depend on how full its block is (profiles/r01_ab_variants.txt), so narrow layers are packed together -- the cross
    products nobody asked for are computed and ignored."""

    def __init__(self):
        self.rows, self.cols, self.kind = [], [], []

    def add(self, row_ranges, col_ranges, kind):
        rows = [f for f0, n in row_ranges for f in range(f0, f0 + n)]
        cols = [f for f0, n in col_ranges for f in range(f0, f0 + n)]
        assert 1 <= len(rows) <= 16 and len(cols) <= 16 and len(row_ranges) <= 2 and len(col_ranges) <= 2
        self.rows.append(rows), self.cols.append(cols), self.kind.append(kind)
        self._ranges = getattr(self, "_ranges", []) + [(list(row_ranges), list(col_ranges))]

    def table(self):
        out = np.zeros((len(self.rows), WG_TABLE_INTS), np.int32)
        for b, (rr, cr) in enumerate(self._ranges):
            rr = rr + [(0, 0)] * (2 - len(rr))
            cr = cr + [(0, 0)] * (2 - len(cr))
            out[b, 0:4] = [rr[0][0], rr[0][1], rr[1][0], rr[1][1]]
            out[b, 4:8] = [cr[0][0], cr[0][1], cr[1][0], cr[1][1]]
            out[b, 8] = self.kind[b]
        return out

    def find(self, row_frag, col_frag):
        """(block, row position, column position) of the block holding both fragments (col_frag None: any block with the row)."""
        for b, (rows, cols) in enumerate(zip(self.rows, self.cols)):
            if row_frag in rows and (col_frag is None or col_frag in cols):
                return b, rows.index(row_frag), (cols.index(col_frag) if col_frag is not None else 0)
        raise KeyError((row_frag, col_frag))


class _Job:
    """Weight-gradient GEMMs of one layer group: dW[row slot][col slot] = sum_points dpre[row] * act[col]; ``rf0`` = first
    dpre fragment of its rows, ``col_segs`` = first activation fragment of each input segment.  Positions are looked up in the
    block table, wherever the packing put the fragments."""

    def __init__(self, table, rf0, col_segs):
        self.table, self.rf0, self.col_segs = table, rf0, col_segs

    def pos(self, row_slot, seg, col_slot):
        row_slot, col_slot = np.broadcast_arrays(np.asarray(row_slot), np.asarray(col_slot))
        out = np.empty(row_slot.shape, np.int64)
        for i in np.ndindex(row_slot.shape):
            r, c = int(row_slot[i]), int(col_slot[i])
            b, rp, cp = self.table.find(self.rf0 + r // 16, self.col_segs[seg] + c // 16)
            out[i] = b * WG_BLOCK_FLOATS + (16 * rp + r % 16) * 256 + 16 * cp + c % 16
        return out

    def pos_aux(self, row_slot, aux_slot):
        row_slot, aux_slot = np.broadcast_arrays(np.asarray(row_slot), np.asarray(aux_slot))
        out = np.empty(row_slot.shape, np.int64)
        for i in np.ndindex(row_slot.shape):
            r = int(row_slot[i])
            b, rp, _ = self.table.find(self.rf0 + r // 16, None)
            out[i] = b * WG_BLOCK_FLOATS + 256 * 256 + (16 * rp + r % 16) * 32 + int(aux_slot[i])
        return out


@functools.lru_cache(maxsize=8)
def backward_maps(feat=256, tau=4):
    """Gather maps of the transposed (dX) stream, the weight-gradient job table and the gradient scatter map.

    Returns dict(idx, scale: bwd stream;  blocks int32 [n_blocks, 12] = (rf0, nr0, rf1, nr1, cf0, nc0, cf1, nc1, col_kind, 0, 0, 0):
    up to two row / two column fragment ranges per block (block_rows / block_cols list the fragments);
    gidx int32 [n_params] position of each parameter's gradient in the block-partial buffer (-1: not produced here),
    gscale fp32 [n_params]).  feat 256: the hand-packed 14-block table; feat 512 (8-bit workspaces only): 47 blocks generated from
    the same job list (a 512 x 512 layer = four 256 x 256 blocks).
    """
    if feat not in (256, 512):
        raise ValueError(f"feat={feat} unsupported by this build (256, 512)")
    half, auxs = feat // 2, aux_steps(tau)
    KS, HS = feat // 16, half // 16
    offsets, n_params = param_offsets(satnerf_param_shapes(feat, tau))

    def transposed(n_in, parts, in_col0=0):
        """rows = natural input index i < n_in; K slots = concatenated row-slot spaces of `parts` [(weight name, n_out_slots, row_of_slot)]."""
        k = sum(p[1] for p in parts)
        m = _Mat(n_in, k, 0, offsets)
        s0 = 0
        for name, nslots, rows in parts:
            rows = np.asarray(rows)  # forward output row held by each slot, -1 = none
            ok = rows >= 0
            i = np.arange(n_in)[:, None]
            m.idx[:, s0:s0 + nslots] = np.where(ok[None, :], m._flat(name, np.maximum(rows, 0)[None, :], in_col0 + i), -1)
            m.scl[:, s0:s0 + nslots] = np.where(ok[None, :], 1.0, 0.0)
            s0 += nslots
        return m

    phi_h, phi_f, phi16 = slot_to_feat(np.arange(half)), slot_to_feat(np.arange(feat)), slot_to_feat(np.arange(16))
    head_rows = lambda lo, hi: np.where((phi16 >= lo) & (phi16 < hi), phi16 - lo, -1)  # noqa: E731
    mats = []
    # bH: d_head (16 slots, rows 0..4 live) -> d rgb hidden | d sun hidden 3 | d beta hidden (3 * half / 32 tiles x 1 piece)
    for name, lo, hi in (("rgb_from_xyzdir.2.weight", 0, 3), ("sun_v_net.6.weight", 3, 4), ("beta_from_xyz.2.weight", 4, 5)):
        mats.append(transposed(half, [(name, 16, head_rows(lo, hi))]))
    mats.append(transposed(half, [("sun_v_net.4.weight", half, phi_h)]))  # bS3
    mats.append(transposed(half, [("sun_v_net.2.weight", half, phi_h)]))  # bS2
    mats.append(transposed(feat, [("rgb_from_xyzdir.0.weight", half, phi_h), ("sun_v_net.0.weight", half, phi_h),
                                  ("beta_from_xyz.0.weight", half, phi_h)]))  # bG2
    dt = transposed(32, [("beta_from_xyz.0.weight", half, phi_h)], in_col0=feat)  # bDT: rows = t index
    dt.idx[tau:, :] = -1
    dt.scl[tau:, :] = 0.0
    mats.append(dt)
    sig_rows = np.where(phi16 == 0, 0, -1)
    mats.append(transposed(feat, [("feats_from_xyz.weight", feat, phi_f), ("sigma_from_xyz.0.weight", 16, sig_rows)]))  # bG1
    for l in range(7, 0, -1):
        mats.append(transposed(feat, [(f"fc_net.{2 * l}.weight", feat, phi_f)], in_col0=3 if l == 4 else 0))
    parts = [_serialize(m) for m in mats]
    idx = np.concatenate([p[0] for p in parts]).astype(np.int32)
    scale = np.concatenate([p[1] for p in parts]).astype(np.float32)

    # ---- weight-gradient jobs: rows = fragments of the dpre workspace, cols = fragments of the saved activations
    A = auxs  # activation fragment offsets (mlp_layout.h)
    a_frag = lambda l: A + KS * l  # noqa: E731
    ACT_FEATS = A + 8 * KS
    ACT_RGBH, ACT_S1, ACT_E1, ACT_S2, ACT_S3 = (ACT_FEATS + KS + k * HS for k in range(5))
    DP_FEATS, DP_SIGMA = 8 * KS, 9 * KS
    DP_RGBH = DP_SIGMA + 1
    DP_S1, DP_E1, DP_S2, DP_S3, DP_HEAD = (DP_RGBH + k * HS for k in range(1, 6))
    tab = _Blocks()
    todo = []  # (row ranges, column ranges, kind, sharing key): added below in an order that puts blocks sharing operands 8 positions apart
    if feat == 256:
        for l in range(1, 8):
            todo.append(([(16 * l, 16)], [(a_frag(l - 1), 16)], KIND_PHASE, None))
        todo.append(([(DP_FEATS, 16)], [(a_frag(7), 16)], KIND_PHASE, "a7"))                          # feats_from_xyz
        # first half of fc_net.0 (aux columns only) + the sigma head.  Every 32-row pair of a block holds ONE double fragment or ONE raw
        # bf16 fragment: the 4-wave kernel fits fp16's range per pair (wgrad9_pair_groups), so the lone sigma fragment sits in a pair of its own
        todo.append(([(0, 8), (DP_SIGMA, 1)], [(a_frag(7), 16)], KIND_PHASE, "a7"))
        todo.append(([(DP_RGBH, 16)], [(ACT_FEATS, 16)], KIND_BF16, "feats"))                         # rgb hidden + sun hidden 1
        todo.append(([(DP_RGBH + 16, 8), (8, 8)], [(ACT_FEATS, 16)], KIND_BF16, "feats"))             # beta hidden + second half of fc_net.0
        todo.append(([(DP_S2, 8), (DP_S3, 8)], [(ACT_S1, 8), (ACT_S2, 8)], KIND_PHASE, None))         # sun hidden 2 and 3
        todo.append(([(DP_HEAD, 1)], [(ACT_RGBH, 8), (ACT_S3, 8)], KIND_PHASE, None))                 # rgb and sun output rows
        todo.append(([(DP_HEAD, 1)], [(ACT_E1, 8)], KIND_PHASE, None))                                # beta output row
    else:  # every job cut into blocks of <= 16 row fragments x <= 16 column fragments
        def tile_job(r0, nr, c0, nc, kind, key=None):
            for rb in range(0, nr, 16):
                for cb in range(0, max(nc, 1), 16):
                    todo.append(([(r0 + rb, min(16, nr - rb))], [(c0 + cb, min(16, nc - cb))] if nc else [], kind, key))
        for l in range(1, 8):
            tile_job(KS * l, KS, a_frag(l - 1), KS, KIND_PHASE, f"L{l}")
        tile_job(DP_FEATS, KS, a_frag(7), KS, KIND_PHASE, "G1")  # feats_from_xyz
        tile_job(DP_SIGMA, 1, a_frag(7), KS, KIND_PHASE)         # sigma head
        tile_job(0, KS, 0, 0, KIND_PHASE)                        # fc_net.0: aux columns only
        tile_job(DP_RGBH, 3 * HS, ACT_FEATS, KS, KIND_BF16)      # rgb / sun-1 / beta hidden layers
        tile_job(DP_S2, HS, ACT_S1, HS, KIND_PHASE)
        tile_job(DP_S3, HS, ACT_S2, HS, KIND_PHASE)
        for c0 in (ACT_RGBH, ACT_S3, ACT_E1):                    # output rows
            tile_job(DP_HEAD, 1, c0, HS, KIND_PHASE)
    # Table order.  The 4-wave weight-gradient kernel (csrc/wgrad9.hip) maps workgroup i to slice i // n_blocks of block i % n_blocks, and
    # workgroup i runs on XCD i % 8 (MI355X_MICROARCH.md: observed dispatch order): blocks 8 table positions apart work on the same tiles on
    # the same XCD at the same time, so an operand both of them read (a7: feats_from_xyz and the sigma head; feats: the three hidden heads;
    # at width 512 the four 256 x 256 blocks of a layer share rows and columns pairwise) comes from HBM once and from that XCD's L2 after.
    # Members of a sharing group therefore go to positions p, p + 8, p + 16, ...; everything else fills the gaps in its old order.
    groups = {}
    for i, t in enumerate(todo):
        if t[3] is not None:
            groups.setdefault(t[3], []).append(i)
    order = [None] * len(todo)
    free = lambda p: p < len(order) and order[p] is None  # noqa: E731
    for members in groups.values():
        p = next((p for p in range(len(order)) if all(free(p + 8 * j) for j in range(len(members)))), None)
        if p is None:
            continue  # (no aligned run left: these blocks keep whatever position remains)
        for j, i in enumerate(members):
            order[p + 8 * j] = i
    rest = iter(i for i in range(len(todo)) if i not in order)
    order = [i if i is not None else next(rest) for i in order]
    assert sorted(order) == list(range(len(todo)))
    for i in order:
        tab.add(*todo[i][:3])
    jobs = {"L0": _Job(tab, 0, [])}
    for l in range(1, 8):
        jobs[f"L{l}"] = _Job(tab, KS * l, [a_frag(l - 1)])
    jobs["G1"] = _Job(tab, DP_FEATS, [a_frag(7)])      # rows 0..feat-1 feats, feat.. sigma (DP_SIGMA = DP_FEATS + KS)
    jobs["G2"] = _Job(tab, DP_RGBH, [ACT_FEATS])
    jobs["S2"] = _Job(tab, DP_S2, [ACT_S1])
    jobs["S3"] = _Job(tab, DP_S3, [ACT_S2])
    jobs["H"] = _Job(tab, DP_HEAD, [ACT_RGBH, ACT_S3, ACT_E1])

    gidx = np.full(n_params, -1, np.int64)
    gscale = np.zeros(n_params, np.float32)
    inv_f, inv_h, inv16 = feat_to_slot(feat), feat_to_slot(half), feat_to_slot(32)[:16]

    def put(name, job, row_slots, seg, col_slots, scale_=1.0, cols=None):
        """grad of W[name][r, cols[c]] (or bias[r] when 1-D) lives at job.pos(row_slots[r], seg, col_slots[c])."""
        off, shp = offsets[name]
        row_slots, col_slots = np.asarray(row_slots), np.asarray(col_slots)
        where = (lambda r, c: job.pos_aux(r, c)) if seg == "aux" else (lambda r, c: job.pos(r, seg, c))
        if len(shp) == 1:
            gidx[off + np.arange(shp[0])] = where(row_slots, np.full_like(row_slots, int(col_slots)))
            gscale[off:off + shp[0]] = scale_
            return
        cols = np.arange(shp[1]) if cols is None else np.asarray(cols)
        r = np.arange(shp[0])[:, None]
        flat = off + r * shp[1] + cols[None, :]
        gidx[flat] = where(np.broadcast_to(row_slots[:, None], flat.shape), np.broadcast_to(col_slots[None, :], flat.shape))
        gscale[flat] = scale_

    AUXC = _Mat
    put("fc_net.0.weight", jobs["L0"], inv_f, "aux", AUXC.AUX_XYZ + np.arange(3), W0_FIRST)
    put("fc_net.0.bias", jobs["L0"], inv_f, "aux", AUXC.AUX_ONE, W0_FIRST)
    for l in range(1, 8):
        name, job = f"fc_net.{2 * l}", jobs[f"L{l}"]
        if l == 4:
            put(name + ".weight", job, inv_f, "aux", AUXC.AUX_XYZ + np.arange(3), cols=[0, 1, 2])
            put(name + ".weight", job, inv_f, 0, inv_f, cols=3 + np.arange(feat))
        else:
            put(name + ".weight", job, inv_f, 0, inv_f)
        put(name + ".bias", job, inv_f, "aux", AUXC.AUX_ONE)
    put("feats_from_xyz.weight", jobs["G1"], inv_f, 0, inv_f)
    put("feats_from_xyz.bias", jobs["G1"], inv_f, "aux", AUXC.AUX_ONE)
    put("sigma_from_xyz.0.weight", jobs["G1"], np.array([feat + inv16[0]]), 0, inv_f)
    put("sigma_from_xyz.0.bias", jobs["G1"], np.array([feat + inv16[0]]), "aux", AUXC.AUX_ONE)
    for k, (name, aux0) in enumerate((("rgb_from_xyzdir.0", None), ("sun_v_net.0", AUXC.AUX_SUN), ("beta_from_xyz.0", AUXC.AUX_T))):
        rows = half * k + inv_h
        put(name + ".weight", jobs["G2"], rows, 0, inv_f, cols=np.arange(feat))
        put(name + ".bias", jobs["G2"], rows, "aux", AUXC.AUX_ONE)
        if aux0 is not None:
            extra = offsets[name + ".weight"][1][1] - feat
            put(name + ".weight", jobs["G2"], rows, "aux", aux0 + np.arange(extra), cols=feat + np.arange(extra))
    put("sun_v_net.2.weight", jobs["S2"], inv_h, 0, inv_h)
    put("sun_v_net.2.bias", jobs["S2"], inv_h, "aux", AUXC.AUX_ONE)
    put("sun_v_net.4.weight", jobs["S3"], inv_h, 0, inv_h)
    put("sun_v_net.4.bias", jobs["S3"], inv_h, "aux", AUXC.AUX_ONE)
    put("rgb_from_xyzdir.2.weight", jobs["H"], inv16[[0, 1, 2]], 0, inv_h)
    put("rgb_from_xyzdir.2.bias", jobs["H"], inv16[[0, 1, 2]], "aux", AUXC.AUX_ONE)
    put("sun_v_net.6.weight", jobs["H"], inv16[[3]], 1, inv_h)
    put("sun_v_net.6.bias", jobs["H"], inv16[[3]], "aux", AUXC.AUX_ONE)
    put("beta_from_xyz.2.weight", jobs["H"], inv16[[4]], 2, inv_h)
    put("beta_from_xyz.2.bias", jobs["H"], inv16[[4]], "aux", AUXC.AUX_ONE)
    return dict(idx=idx, scale=scale, blocks=tab.table(), block_rows=tab.rows, block_cols=tab.cols, gidx=gidx.astype(np.int32), gscale=gscale,
                n_params=n_params, auxs=auxs, offsets=offsets, feat=feat)


# ------------------------------------------------------------------------------------------------ 8-bit workspaces
# csrc/mlp_layout.h (SR_FMT8): logical 16-bit fragment f of a workspace -> (unit, codec[, scale unit, scale byte]).
SRC_DPRE, SRC_ACTS = 1, 2
RAW16, PHASE8, MX8 = 0, 1, 2
WG8_OLD_INTS = 20          # load table of the 16- / 8-wave kernels (wgrad8.hip, wgrad8f.hip)
WG9_DUTY_INTS = 4 * 5 * 4  # duty table of the 4-wave kernel (wgrad9.hip): 4 waves x 5 duties x (source, unit, LDS fragment, scale)
WG9_SCAN_INTS = 8          # ... + the exponent group (byte of an entry of the dX kernel's table of exponent maxima) of each of the block's 8 row pairs; -1 = none
EMAX_FEATS, EMAX_RAW = 14, 15   # ... bytes 0..13 = the dpre scale groups (mlp_layout.h), 14 = the saved feats (columns of the MX8 blocks), 15 = the bf16 rows d_sigma_pre / d_head
WG9_MASK_INTS = 1          # ... + the mask of the 128 x 128 quadrants (bit = wave = 2 row half + column half) somebody reads
WG9_VARIANT_INTS = 4       # ... + the instruction stream of each wave: 0 = the full stream (its aux-only form where the quadrant mask says so), k > 0 = thin stream WG9_THIN[k - 1]
WG8_LOAD_INTS = WG8_OLD_INTS + WG9_DUTY_INTS + WG9_SCAN_INTS + WG9_MASK_INTS + WG9_VARIANT_INTS
# thin streams of csrc/gen/wgrad9_loop.py (THIN), in the order csrc/wgrad9.hip numbers them: (column double fragments, raw fragment, MFMAs)
WG9_THIN = ((1, True, "thin"), (3, False, "none"), (0, True, "thin"), (2, True, "none"), (2, False, "none"), (0, False, "none"))
WG9_DUMP_FRAG = 34         # LDS fragment an unused duty decodes into (csrc/gen/wgrad9_loop.py: 16 rows + 16 columns + 2 aux + 2 dump)
def fmt8_geometry(feat=256):
    """Unit (1 KiB) offsets of the 8-bit workspaces (csrc/mlp_layout.h kD8* / kA8*) and the 16-bit fragment numbers they map."""
    KS, HS = feat // 16, feat // 32
    MT = KS // 2
    body = (9 * KS + 5 * HS) // 2
    gpu = 16 // MT                                   # scale groups per scale unit
    d8_scale = body + 2
    return dict(KS=KS, HS=HS, MT=MT, MTH=HS // 2, D8_SIGMA=body, D8_HEAD=body + 1, D8_SCALE=d8_scale, GROUPS_PER_UNIT=gpu,
                D8_UNITS=d8_scale + (14 + gpu - 1) // gpu, A8_SCALE=body,
                DP_FEATS=8 * KS, DP_SIGMA=9 * KS, DP_RGBH=9 * KS + 1, DP_HEAD=9 * KS + 1 + 5 * HS, ACT_FEATS=8 * KS)


D8_SIGMA, D8_HEAD, D8_SCALE, D8_UNITS, A8_SCALE = (fmt8_geometry(256)[k] for k in ("D8_SIGMA", "D8_HEAD", "D8_SCALE", "D8_UNITS", "A8_SCALE"))


def act8_units(auxs, feat=256):
    return auxs + fmt8_geometry(feat)["A8_SCALE"] + 1


def dpre8_units(feat=256):
    return fmt8_geometry(feat)["D8_UNITS"]


def dpre8_source(f, feat=256):
    """Logical dpre fragment -> dict(unit, codec, half[, scale_unit, scale_byte]) in the 8-bit layout."""
    g8 = fmt8_geometry(feat)
    if f == g8["DP_SIGMA"]:
        return dict(unit=g8["D8_SIGMA"], codec=RAW16, half=0, group=EMAX_RAW)
    if f == g8["DP_HEAD"]:
        return dict(unit=g8["D8_HEAD"], codec=RAW16, half=0, group=EMAX_RAW)
    if f < g8["DP_SIGMA"]:
        u, half = f >> 1, f & 1
        g, k = u // g8["MT"], u % g8["MT"]                 # trunk layers 0..7, d_feats = 8
    else:
        u, half = (f - 1) >> 1, (f - 1) & 1
        h = u - 9 * g8["MT"]
        g, k = 9 + h // g8["MTH"], h % g8["MTH"]
    gpu = g8["GROUPS_PER_UNIT"]
    return dict(unit=u, codec=MX8, half=half, scale_unit=g8["D8_SCALE"] + g // gpu, scale_byte=(g % gpu) * g8["MT"] + k, group=g)


def act8_source(f, auxs, feat=256):
    """Logical activation fragment (aux offset included, as in the job table) -> its 8-bit source."""
    g8 = fmt8_geometry(feat)
    a = f - auxs
    assert a >= 0, "aux fragments are fetched by the kernel itself"
    if g8["ACT_FEATS"] <= a < g8["ACT_FEATS"] + g8["KS"]:  # feats: identity stage -> MX8
        return dict(unit=auxs + (a >> 1), codec=MX8, half=a & 1, scale_unit=auxs + g8["A8_SCALE"], scale_byte=(a - g8["ACT_FEATS"]) >> 1)
    return dict(unit=auxs + (a >> 1), codec=PHASE8, half=a & 1)


@functools.lru_cache(maxsize=8)
def wgrad8_loads(feat=256, tau=4):
    """Load tables of the 8-bit weight-gradient kernels for the job blocks of ``backward_maps``: int32 [n_blocks, 100].

    Per block: ints 0..15 = the primary load of wave w of csrc/wgrad8.hip (0 = none; bits 0-1 source, 2-3 codec, 4-11 unit, 12-17 operand
    fragment, 18-19 scale area, 20-23 scale byte), ints 16..18 = the scale unit fetched into scale area 0..2 (0 = none); ints 20..99 =
    the duty table of csrc/wgrad9.hip (``wgrad9_duties``)."""
    bm = backward_maps(feat, tau)
    auxs = bm["auxs"]
    out = np.zeros((len(bm["block_rows"]), WG8_LOAD_INTS), np.int32)
    for b, (rows, cols) in enumerate(zip(bm["block_rows"], bm["block_cols"])):
        loads, areas = [], []

        def area_of(src, unit):
            if (src, unit) not in areas:
                areas.append((src, unit))
            return areas.index((src, unit))

        for base, frags, src, lookup in ((0, rows, SRC_DPRE, lambda f: dpre8_source(f, feat)), (16, cols, SRC_ACTS, lambda f: act8_source(f, auxs, feat))):
            pos = 0
            while pos < len(frags):
                d = lookup(frags[pos])
                desc = src | (d["codec"] << 2) | (d["unit"] << 4) | ((base + pos) << 12)
                if d["codec"] == RAW16:
                    pos += 1
                else:
                    assert d["half"] == 0 and pos + 1 < len(frags) and frags[pos + 1] == frags[pos] + 1, (b, frags, pos)
                    if d["codec"] == MX8:
                        desc |= (area_of(src, d["scale_unit"]) << 18) | (d["scale_byte"] << 20)
                    pos += 2
                loads.append(desc)
        assert len(loads) <= 16 and len(areas) <= 3, (b, len(loads), len(areas))
        # waves 0, 1 also fetch the aux fragments and waves 2..4 the scale units: hand the primaries to the others first
        order = list(range(5, 16)) + [4, 3, 2, 1, 0]
        for w, desc in zip(order, loads):
            out[b, w] = desc
        for k, (src, unit) in enumerate(areas):
            out[b, 16 + k] = src | (unit << 4)
    duties = wgrad9_duties(feat, tau)
    out[:, WG8_OLD_INTS:WG8_OLD_INTS + WG9_DUTY_INTS] = duties
    out[:, WG8_OLD_INTS + WG9_DUTY_INTS:WG8_OLD_INTS + WG9_DUTY_INTS + WG9_SCAN_INTS] = wgrad9_pair_groups(feat, tau)
    out[:, WG8_OLD_INTS + WG9_DUTY_INTS + WG9_SCAN_INTS + WG9_MASK_INTS:] = wgrad9_variants(feat, tau)
    # quadrant mask: which (row half, column half) of each 256 x 256 block holds a gradient the scatter map reads
    g = bm["gidx"][bm["gidx"] >= 0].astype(np.int64)
    blk, w = g // WG_BLOCK_FLOATS, g % WG_BLOCK_FLOATS
    main = w < 256 * 256
    quad = 2 * ((w[main] // 256) // 128) + (w[main] % 256) // 128
    for b, q in zip(blk[main], quad):
        out[b, WG8_OLD_INTS + WG9_DUTY_INTS + WG9_SCAN_INTS] |= 1 << int(q)
    return out


@functools.lru_cache(maxsize=8)
def wgrad9_pair_groups(feat=256, tau=4):
    """int32 [n_blocks, 8]: the exponent group of each 32-row pair of a block's row operand (-1 = the block has no such pair).

    csrc/wgrad9.hip contracts fp16 operands; fp16's range is fitted PER ROW PAIR (= per 32 x 32 accumulator tile): rows are decoded times
    2^(138 - Emax) where Emax is the largest exponent of the pair's source over the workgroup's slice of points, read from the table the dX
    kernel leaves behind the dpre workspace (one byte per group and 4 tiles; mlp_layout.h).  That needs every pair to have ONE source:
    a double fragment at an even fragment position or a raw bf16 fragment alone -- asserted here."""
    bm = backward_maps(feat, tau)
    out = np.full((len(bm["block_rows"]), WG9_SCAN_INTS), -1, np.int32)
    for b, rows in enumerate(bm["block_rows"]):
        pos = 0
        while pos < len(rows):
            d = dpre8_source(rows[pos], feat)
            assert pos % 2 == 0, (b, rows, "a row operand must start a 32-row pair")
            out[b, pos // 2] = d["group"]
            if d["codec"] == RAW16:
                assert pos + 1 == len(rows), (b, rows, "a raw row fragment must be the last of its block")
                pos += 1
            else:
                pos += 2
    return out


def wgrad9_thin_blocks(feat=256, tau=4):
    """{block: number of column double fragments} of the job blocks that run THIN streams (csrc/gen/wgrad9_loop.py, r06): the row operand is
    ONE raw bf16 fragment (d_head / d_sigma_pre: a single live 32-row pair), the columns are 4 or 8 PHASE8 double fragments, and there is one
    aux fragment (tau <= 8).  The full stream spends a full block's time per tile on them (two dump row duties decoded at full price)."""
    bm = backward_maps(feat, tau)
    out = {}
    if os.environ.get("SATNERF_WGRAD_THIN", "1") == "0" or bm["auxs"] != 1:
        return out
    for b, (rows, cols) in enumerate(zip(bm["block_rows"], bm["block_cols"])):
        if len(rows) == 1 and dpre8_source(rows[0], feat)["codec"] == RAW16 and bm["blocks"][b, 8] == KIND_PHASE and len(cols) in (8, 16):
            if all(act8_source(c, bm["auxs"], feat)["codec"] == PHASE8 for c in cols):
                out[b] = len(cols) // 2
    return out


@functools.lru_cache(maxsize=8)
def wgrad9_variants(feat=256, tau=4):
    """int32 [n_blocks, 4]: the instruction stream wave w of a block's workgroup runs -- 0 = the full stream, k > 0 = WG9_THIN[k - 1].
    8 column double fragments: waves 0, 1 own the two live quadrants (raw fragment + 1 double fragment + the MFMAs of row pair 0), waves 2, 3
    decode three double fragments each.  4 of them: wave 0 (raw row fragment, MFMAs), wave 1 (aux fragment + 2), wave 2 (2), wave 3 idle."""
    thin = wgrad9_thin_blocks(feat, tau)
    out = np.zeros((len(backward_maps(feat, tau)["block_rows"]), WG9_VARIANT_INTS), np.int32)
    vid = lambda v: 1 + WG9_THIN.index(v)  # noqa: E731
    for b, n in thin.items():
        out[b] = ([vid((1, True, "thin")), vid((1, True, "thin")), vid((3, False, "none")), vid((3, False, "none"))] if n == 8 else
                  [vid((0, True, "thin")), vid((2, True, "none")), vid((2, False, "none")), vid((0, False, "none"))])
    return out


@functools.lru_cache(maxsize=8)
def wgrad9_duties(feat=256, tau=4):
    """Duty table of csrc/wgrad9.hip: int32 [n_blocks, 80] = 4 waves x 5 duties x (source, unit, LDS fragment, scale).

    A workgroup of four waves owns one job block; per 32-point tile every wave fetches, decodes and writes to the LDS slot
      duties 0, 1: one MX8 double fragment of dpre rows each     duties 2, 3: one double fragment of activation columns each (PHASE8, or
      MX8 for the feats columns)     duty 4: one raw bf16 fragment (an aux fragment, or the bf16 row fragment d_sigma_pre / d_head).
    ``source`` 1 = dpre, 2 = acts workspace; ``unit`` = 1-KiB unit within the tile; ``LDS fragment`` 0..15 rows, 16..31 columns, 32 / 33
    aux, WG9_DUMP_FRAG = unused duty (it fetches a valid unit and decodes it into the dump fragments); ``scale`` = 16 * unit + byte of
    the lane's MX8 exponent (same source).  Rows are always read from dpre and columns from acts (the kernel steps one per-lane offset per
    workspace), the raw duty from either."""
    bm = backward_maps(feat, tau)
    auxs = bm["auxs"]
    out = np.zeros((len(bm["block_rows"]), 4, 5, 4), np.int32)
    thin = wgrad9_thin_blocks(feat, tau)
    for b, (rows, cols) in enumerate(zip(bm["block_rows"], bm["block_cols"])):
        out[b, :, 0:2] = [SRC_DPRE, 0, WG9_DUMP_FRAG, 0]
        out[b, :, 2:4] = [SRC_ACTS, auxs, WG9_DUMP_FRAG, 16 * auxs]
        out[b, :, 4] = [SRC_ACTS, 0, WG9_DUMP_FRAG, 0]
        raw_wave = 0
        for a in range(auxs):                                    # aux fragments: waves 1, 2
            out[b, 1 + a, 4] = [SRC_ACTS, a, 32 + a, 0]
        if b in thin:   # thin streams: duty slots 0.. hold COLUMN double fragments (wgrad9_variants says which wave takes how many)
            d = dpre8_source(rows[0], feat)
            out[b, 0, 4] = [SRC_DPRE, d["unit"], 0, 0]
            share = ([1, 1, 3, 3] if thin[b] == 8 else [0, 2, 2, 0])
            j = 0
            for w, k in enumerate(share):
                for slot in range(k):
                    c = act8_source(cols[2 * j], auxs, feat)
                    assert c["codec"] == PHASE8 and c["half"] == 0 and cols[2 * j + 1] == cols[2 * j] + 1
                    out[b, w, slot] = [SRC_ACTS, c["unit"], 16 + 2 * j, 16 * auxs]
                    j += 1
            assert j == thin[b]
            continue
        for base, frags, src, lookup, d0 in ((0, rows, SRC_DPRE, lambda f: dpre8_source(f, feat), 0),
                                             (16, cols, SRC_ACTS, lambda f: act8_source(f, auxs, feat), 2)):
            pos, n_df = 0, 0
            while pos < len(frags):
                d = lookup(frags[pos])
                if d["codec"] == RAW16:
                    assert src == SRC_DPRE and raw_wave == 0, (b, frags)   # one bf16 row fragment per block: wave 0 (aux: waves 1, 2)
                    out[b, 0, 4] = [src, d["unit"], base + pos, 0]
                    raw_wave = 1
                    pos += 1
                    continue
                assert d["half"] == 0 and pos + 1 < len(frags) and frags[pos + 1] == frags[pos] + 1, (b, frags, pos)
                assert (d["codec"] == MX8) == (src == SRC_DPRE or bm["blocks"][b, 8] == KIND_BF16), (b, frags, pos)
                assert n_df < 8, (b, frags)
                scale = 16 * d["scale_unit"] + d["scale_byte"] if d["codec"] == MX8 else 16 * auxs
                out[b, n_df % 4, d0 + n_df // 4] = [src, d["unit"], base + pos, scale]
                n_df += 1
                pos += 2
    return out.reshape(len(bm["block_rows"]), WG9_DUTY_INTS)


PACK_POS_BITS, PACK_L0_FLAG = 26, 1 << 28


@functools.lru_cache(maxsize=8)
def pack_scatter_map(feat=256, tau=4):
    """The inverse of the gather maps ``sr_pack_all`` runs (forward stream | transposed stream | fp32 fc_net.0 table), for the launch
    that updates the parameters (sr_grad_tail_adam's ``pack``): int32 [n_params, 2] = the (at most two) places parameter i is copied
    to, each  position | scale index << 26 | (1 << 28: the fc_net.0 table),  -1 = none; positions count through ``forward_maps`` idx
    followed by ``backward_maps`` idx, exactly the buffer ``SatNeRF.repack(backward=True)`` packs.  Returns (map, scales[4]): the distinct
    non-zero scale factors of the three maps (1, 1 / 2 pi, 30 / 2 pi).  A stream element whose gather index is negative is a constant
    zero: the first sr_pack_all wrote it and nothing changes it."""
    fm, bm = forward_maps(feat, tau), backward_maps(feat, tau)
    n = int(bm["n_params"])
    idx = np.concatenate([fm["idx"], bm["idx"]]).astype(np.int64)
    scale = np.concatenate([fm["scale"], bm["scale"]]).astype(np.float32)
    l0_idx, l0_scale = fm["l0_idx"].astype(np.int64), fm["l0_scale"].astype(np.float32)
    assert idx.size < (1 << PACK_POS_BITS) and l0_idx.size < (1 << PACK_POS_BITS)
    live = (idx >= 0) & (scale != 0)
    live0 = (l0_idx >= 0) & (l0_scale != 0)
    values = sorted(set(np.unique(scale[live]).tolist()) | set(np.unique(l0_scale[live0]).tolist()))
    assert 1 <= len(values) <= 4, values
    scales = np.zeros(4, np.float32)
    scales[:len(values)] = values
    values_arr = np.array(values, np.float32)
    pos, pos0 = np.nonzero(live)[0], np.nonzero(live0)[0]
    params = np.concatenate([idx[pos], l0_idx[pos0]])
    words = np.concatenate([pos | (np.searchsorted(values_arr, scale[pos]).astype(np.int64) << PACK_POS_BITS),
                            pos0 | (np.searchsorted(values_arr, l0_scale[pos0]).astype(np.int64) << PACK_POS_BITS) | PACK_L0_FLAG])
    order = np.argsort(params, kind="stable")
    params, words = params[order], words[order]
    first = np.r_[True, params[1:] != params[:-1]]
    rank = np.arange(params.size) - np.maximum.accumulate(np.where(first, np.arange(params.size), 0))  # 0, 1, .. within a parameter's run
    assert rank.max() < 2, "a parameter is copied to more than two places"
    out = np.full((n, 2), -1, np.int32)
    out[params, rank] = words
    return out, scales
