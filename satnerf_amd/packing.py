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
    if feat != 256:
        raise ValueError(f"feat={feat} unsupported by this build (256)")
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
