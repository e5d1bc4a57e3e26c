"""The generated instruction stream of the fused forward core (csrc/gen/fwd_core.py -> csrc/mlp_fwd_core_a*.inc): the committed files
are current, and the instruction list -- executed on a lane-accurate numpy model of the VGPR file, the LDS ring, the LDS-DMA rows and
v_mfma_f32_32x32x16_bf16 -- reproduces the forward pass of tests/mfma_emulator.py (itself held to the oracle) on the packed weight
stream.  This validates register allocation, piece addressing, operand order, the lgkmcnt / vmcnt counts and the ring protocol
(no piece read before its rendezvous, no ring slot overwritten before every wave consumed it) and the hazards the assembler does not
pad inside inline asm (XDL write -> VALU read, VALU write -> MFMA operand, trans -> VALU use, M0 write -> LDS-DMA) without a GPU."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import satnerf_oracle as O
from tests import mfma_emulator as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "satnerf_amd", "csrc", "gen", "fwd_core.py")


def _gen():
    spec = importlib.util.spec_from_file_location("fwd_core_gen", GEN)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("auxs", [1, 2])
@pytest.mark.parametrize("save", [0, 8])
def test_generated_files_are_current(auxs, save):
    g = _gen()
    want = g.Core(auxs, save=save).inc_file()
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", f"mlp_fwd_core_a{auxs}{'s8' if save else ''}.inc")) as f:
        assert f.read() == want, "re-run satnerf_amd/csrc/gen/fwd_core.py"
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", "mlp_fwd_core_clobbers_s8.inc" if save else "mlp_fwd_core_clobbers.inc")) as f:
        assert f.read() == g.Core.clobber_file(save)


@pytest.mark.parametrize("tau,save", [(4, 0), (16, 0), (4, 8), (16, 8)])
def test_instruction_stream_computes_the_forward_pass(tau, save):
    g = _gen()
    auxs = g.aux_steps(tau)
    # (SR_FWD_ABLATE: run the lane model on an experimental ordering of the same stream, e.g. phasefirst, before it goes to the GPU)
    core = g.Core(auxs, save=save, ablate=tuple(x for x in os.environ.get("SR_FWD_ABLATE", "").split(",") if x))
    params = O.procedural_satnerf_params(256, tau, seed=3)
    flat = np.concatenate([v.numpy().reshape(-1) for v in params.values()]).astype(np.float32)
    em = E.Emulator(flat, 256, tau, bf16=True, l0_split=True)
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-1, 1, (32, 3))
    sun = rng.normal(size=(32, 3))
    sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    t = rng.uniform(-1, 1, (32, tau))
    albedo, sigma, sun_v, beta = em.forward_tile(xyz, sun, t)

    # packed stream as the kernel sees it: [piece, lane = h * 32 + row, 8 bf16] -> 4 dwords per lane
    st = em.stream.reshape(-1, 64, 8).astype(np.float32)
    bits = g.bf16_bits(st)
    l0b = g.bf16_bits(g.l0_pieces(em.l0.astype(np.float32)))  # fc_net.0's 16 pieces: the kernel prologue writes them into the ring
    bits = np.concatenate([l0b, bits])
    stream_bits = (bits[:, :, 0::2] | (bits[:, :, 1::2] << 16)).astype(np.uint32)
    assert stream_bits.shape[0] == core.n_pieces

    m = g.Machine(core, stream_bits)
    for s, frag in enumerate(g.l0_b_frags(xyz.astype(np.float32))):  # the split coordinates (the C++ prologue's job) and the aux fragment(s)
        m.v[g.L0B + 4 * s:g.L0B + 4 * s + 4] = g.f32_to_frag(frag)
    for a in range(auxs):
        m.v[g.AUX + 4 * a:g.AUX + 4 * a + 4] = g.f32_to_frag(em.saved["aux"][a])
    m.v[g.KMAGIC] = np.full(64, 49152.0, np.float32).view(np.uint32)
    m.v[g.K128] = np.full(64, 128.0, np.float32).view(np.uint32)
    m.run()
    if save:
        # PHASE8 units: byte g of the lane's 16 = round(frac(pre-activation) * 256) mod 256 of fragment pair (2t, 2t+1)
        def unit_bytes(u):
            w = m.stores[u]  # [4, 64]
            return np.stack([(w[q] >> np.uint32(8 * j)) & 0xFF for q in range(4) for j in range(4)], 1).astype(np.int64)  # [64, 16]
        stages = [(f"a{l}", 8 * l, 8) for l in range(0, 8)] + [("rgbh", 72, 4), ("s1", 76, 4), ("e1", 80, 4), ("s2", 84, 4), ("s3", 88, 4)]
        for tag, u0, nt in stages:
            pre = em.saved["pre"][tag]
            for t in range(nt):
                want = np.rint((np.concatenate([pre[2 * t], pre[2 * t + 1]], 1) % 1.0) * 256).astype(np.int64) % 256
                d = (unit_bytes(auxs + u0 + t) - want) % 256
                assert np.minimum(d, 256 - d).max() <= 1, (tag, t)
        # MX8 feats: (u - 128) * 2^(E - 133) within one quantum of the value, E in byte t of the scale unit
        sc = m.stores[auxs + 92]
        for t in range(8):
            e = ((sc[t >> 2] >> np.uint32(8 * (t & 3))) & 0xFF).astype(np.int64)  # [64]
            val = (unit_bytes(auxs + 64 + t) - 128) * np.exp2(e - 133.0)[:, None]
            ref = np.concatenate([em.saved["feats"][2 * t], em.saved["feats"][2 * t + 1]], 1)
            assert (np.abs(val - ref) <= 1.01 * np.exp2(e - 133.0)[:, None] + 2.0 ** -8 * np.abs(ref)).all(), t
            assert (np.abs(ref).max(1) <= 127.01 * np.exp2(e - 133.0)).all() and (e >= 6).all()
        assert sorted(m.stores) == sorted([auxs + u for u in list(range(0, 92)) + [92]])

    head = np.stack([m.f(g.HEAD + r) for r in range(5)], 1).astype(np.float64)  # [lane, row]
    sig = m.f(g.SIG).astype(np.float64)
    sigmoid = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    softplus = lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20))))  # noqa: E731
    got = (sigmoid(head[:32, 0:3]) * 1.002 - 0.001, softplus(sig[:32]), sigmoid(head[:32, 3]), softplus(head[32:, 0]))
    for name, a, b in zip(("albedo", "sigma", "sun_v", "beta"), got, (albedo, sigma, sun_v, beta)):
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err < 2e-5, (name, err)  # same bf16 operands; fp32 vs fp64 accumulation and sin


# ---- the parity-mode (bf16x3) core: csrc/gen/fwd_core3.py -> csrc/mlp_fwd3_core_a*.inc ---------------------------------------------------
GEN3 = os.path.join(ROOT, "satnerf_amd", "csrc", "gen", "fwd_core3.py")


def _gen3():
    spec = importlib.util.spec_from_file_location("fwd_core3_gen", GEN3)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("auxs", [1, 2])
@pytest.mark.parametrize("save", [0, 16])
def test_generated_parity_core_is_current(auxs, save):
    g = _gen3()
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", f"mlp_fwd3_core_a{auxs}{'s16' if save else ''}.inc")) as f:
        assert f.read() == g.Core3(auxs, save=save).inc_file(), "re-run satnerf_amd/csrc/gen/fwd_core3.py"
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", "mlp_fwd3_core_clobbers_s16.inc" if save else "mlp_fwd3_core_clobbers.inc")) as f:
        assert f.read() == g.Core3.clobber_file(save)


@pytest.mark.parametrize("tau,save", [(4, 0), (16, 0), (4, 16), (16, 16)])
def test_parity_core_stream_computes_the_forward_pass(tau, save):
    """hi / lo planes everywhere, three MFMAs per k-step: the instruction list on the unified 512-register model against the fp64
    emulator (exact operands) -- the split representation carries ~16 bits, so 1e-5 of the outputs"""
    g = _gen3()
    auxs = g.aux_steps(tau)
    core = g.Core3(auxs, save=save)
    params = O.procedural_satnerf_params(256, tau, seed=3)
    flat = np.concatenate([v.numpy().reshape(-1) for v in params.values()]).astype(np.float32)
    em = E.Emulator(flat, 256, tau, bf16=False)
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-1, 1, (32, 3))
    sun = rng.normal(size=(32, 3))
    sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    t = rng.uniform(-1, 1, (32, tau))
    albedo, sigma, sun_v, beta = em.forward_tile(xyz, sun, t)

    def planes(v):  # fp32 values -> (hi, lo) bf16 bit planes, hi = RNE(v), lo = RNE(v - hi)
        v = np.asarray(v, np.float32)
        hb = g.bf16_bits(v)
        lb = g.bf16_bits((v - g.bf16_to_f32(hb)).astype(np.float32))
        return hb, lb

    def words(b):  # [..., 8] bf16 bits -> [..., 4] dwords (element 0 in the low half)
        return (b[..., 0::2] | (b[..., 1::2] << 16)).astype(np.uint32)

    st = em.stream.reshape(-1, 64, 8).astype(np.float32)
    sh, sl = planes(st)
    assert st.shape[0] == core.n_units
    m = g.Machine3(core, words(sh), words(sl))
    for k in range(16):  # the C++ prologue's outputs: fc_net.0 activations (hi plane in v[0:63], lo plane arrives in v[64:127])
        hb, lb = planes(em.saved["a"][0][k])
        m.v[g.XH + 4 * k:g.XH + 4 * k + 4] = words(hb).T
        m.v[g.IN_XL + 4 * k:g.IN_XL + 4 * k + 4] = words(lb).T
    for a in range(auxs):
        hb, lb = planes(em.saved["aux"][a])
        m.v[g.IN_AUXH + 4 * a:g.IN_AUXH + 4 * a + 4] = words(hb).T
        m.v[g.IN_AUXL + 4 * a:g.IN_AUXL + 4 * a + 4] = words(lb).T
    m.run()
    head = np.stack([m.f(g.OUT_HEAD + r) for r in range(5)], 1).astype(np.float64)
    sig = m.f(g.SIG).astype(np.float64)
    sigmoid = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    softplus = lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20))))  # noqa: E731
    got = (sigmoid(head[:32, 0:3]) * 1.002 - 0.001, softplus(sig[:32]), sigmoid(head[:32, 3]), softplus(head[32:, 0]))
    for name, a, b in zip(("albedo", "sigma", "sun_v", "beta"), got, (albedo, sigma, sun_v, beta)):
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err < 3e-5, (name, err)  # hi + lo planes carry 16-17 bits of every operand
    if save:
        # SR_FMT16 workspaces (mlp_layout.h): fragment pair (F + 2 t, F + 2 t + 1) of a sin stage = unorm16 phases of its pre-activations,
        # of feats = the bf16 values; word q of a fragment = values 2 q, 2 q + 1 of the lane
        def halves(frag):
            w = m.stores[frag]  # [4, 64]
            return np.stack([(w[q] >> np.uint32(16 * j)) & 0xFFFF for q in range(4) for j in range(2)], 1).astype(np.int64)  # [64, 8]
        stages = [(f"a{l}", 16 * l, 8) for l in range(1, 8)] + [("rgbh", 144, 4), ("s1", 152, 4), ("e1", 160, 4), ("s2", 168, 4), ("s3", 176, 4)]
        for tag, f0, nt in stages:
            for k in range(2 * nt):
                want = np.rint((em.saved["pre"][tag][k] % 1.0) * 65535).astype(np.int64)
                d = np.abs(halves(auxs + f0 + k) - want)
                assert np.minimum(d, 65535 - d).max() <= 24, (tag, k)  # fp32 accumulation of ~16-bit operands vs fp64: 3e-4 of a revolution
        for k in range(16):
            got = g.bf16_to_f32(halves(auxs + 128 + k).astype(np.uint32))
            ref = em.saved["feats"][k]
            assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-4).all(), k
        assert sorted(m.stores) == sorted(auxs + f for f in list(range(16, 144)) + list(range(144, 184)))


# ---- the width-512 core: csrc/gen/fwd_core512.py -> csrc/mlp_fwd512_core_a*.inc ---------------------------------------------------------
GEN512 = os.path.join(ROOT, "satnerf_amd", "csrc", "gen", "fwd_core512.py")


def _gen512():
    spec = importlib.util.spec_from_file_location("fwd_core512_gen", GEN512)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("auxs", [1, 2])
@pytest.mark.parametrize("save", [0, 8])
def test_generated_width512_core_is_current(auxs, save):
    g = _gen512()
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", f"mlp_fwd512_core_a{auxs}{'s8' if save else ''}.inc")) as f:
        assert f.read() == g.Core512(auxs, save=save).inc_file(), "re-run satnerf_amd/csrc/gen/fwd_core512.py"
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", "mlp_fwd512_core_clobbers_s8.inc" if save else "mlp_fwd512_core_clobbers.inc")) as f:
        assert f.read() == g.Core512.clobber_file(save)


@pytest.mark.parametrize("tau,save", [(4, 0), (16, 0), (4, 8), (16, 8)])
def test_width512_core_stream_computes_the_forward_pass(tau, save):
    g = _gen512()
    auxs = g.aux_steps(tau)
    core = g.Core512(auxs, save=save)
    params = O.procedural_satnerf_params(512, tau, seed=3)
    flat = np.concatenate([v.numpy().reshape(-1) for v in params.values()]).astype(np.float32)
    em = E.Emulator(flat, 512, tau, bf16=True)
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-1, 1, (32, 3))
    sun = rng.normal(size=(32, 3))
    sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    t = rng.uniform(-1, 1, (32, tau))
    albedo, sigma, sun_v, beta = em.forward_tile(xyz, sun, t)
    bits = g.bf16_bits(em.stream.reshape(-1, 64, 8).astype(np.float32))
    stream_bits = (bits[:, :, 0::2] | (bits[:, :, 1::2] << 16)).astype(np.uint32)
    assert stream_bits.shape[0] == core.n_pieces
    m = g.Machine512(core, stream_bits)
    for k in range(32):  # fc_net.0 output (the C++ prologue's job) and the aux fragment(s), which arrive in VGPRs
        m.v[g.X + 4 * k:g.X + 4 * k + 4] = g.f32_to_frag(em.saved["a"][0][k])
    for a in range(auxs):
        m.v[g.IN_AUX + 4 * a:g.IN_AUX + 4 * a + 4] = g.f32_to_frag(em.saved["aux"][a])
    m.v[g.KMAGIC] = np.full(64, 49152.0, np.float32).view(np.uint32)
    m.v[g.K128] = np.full(64, 128.0, np.float32).view(np.uint32)
    m.run()
    if save:  # the 8-bit workspaces at width 512 (mlp_layout.h): a_l at 16 l + t, feats 128 + t, rgbh 144, s1 152, e1 160, s2 168, s3 176, scales 184
        def unit_bytes(u):
            w = m.stores[u]
            return np.stack([(w[q] >> np.uint32(8 * j)) & 0xFF for q in range(4) for j in range(4)], 1).astype(np.int64)  # [64, 16]
        stages = [(f"a{l}", 16 * l, 16) for l in range(1, 8)] + [("rgbh", 144, 8), ("s1", 152, 8), ("e1", 160, 8), ("s2", 168, 8), ("s3", 176, 8)]
        for tag, u0, nt in stages:
            pre = em.saved["pre"][tag]
            for t in range(nt):
                want = np.rint((np.concatenate([pre[2 * t], pre[2 * t + 1]], 1) % 1.0) * 256).astype(np.int64) % 256
                d = (unit_bytes(auxs + u0 + t) - want) % 256
                assert np.minimum(d, 256 - d).max() <= 1, (tag, t)
        sc = unit_bytes(auxs + 184)  # [64, 16]: byte t = E of feats tile t
        for t in range(16):
            e = sc[:, t]
            val = (unit_bytes(auxs + 128 + t) - 128) * np.exp2(e - 133.0)[:, None]
            ref = np.concatenate([em.saved["feats"][2 * t], em.saved["feats"][2 * t + 1]], 1)
            assert (np.abs(val - ref) <= 1.01 * np.exp2(e - 133.0)[:, None] + 2.0 ** -8 * np.abs(ref)).all(), t
            assert (np.abs(ref).max(1) <= 127.01 * np.exp2(e - 133.0)).all() and (e >= 6).all()
        assert sorted(m.stores) == sorted([auxs + u for u in list(range(16, 184)) + [184]])
    head = np.stack([m.f(g.OUT_HEAD + r) for r in range(5)], 1).astype(np.float64)
    sig = m.f(g.SIG).astype(np.float64)
    sigmoid = lambda v: 1 / (1 + np.exp(-v))  # noqa: E731
    softplus = lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20))))  # noqa: E731
    got = (sigmoid(head[:32, 0:3]) * 1.002 - 0.001, softplus(sig[:32]), sigmoid(head[:32, 3]), softplus(head[32:, 0]))
    for name, a, b in zip(("albedo", "sigma", "sun_v", "beta"), got, (albedo, sigma, sun_v, beta)):
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err < 5e-4, (name, err)  # K = 512 fp32 accumulation vs fp64: a few bf16 roundings of intermediate activations flip
