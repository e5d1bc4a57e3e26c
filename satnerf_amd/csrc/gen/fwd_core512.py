#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of the WIDTH-512 forward core (csrc/mlp_fwd512g.inc), inference and the
one-launch render pass in the single-pass modes (bf16 / f16 operands).

Same method as fwd_core.py (read that docstring first); the machine shape is fwd_core3.py's: at width 512 one activation vector of a
32-point wave is 128 registers, so the wave runs alone on its SIMD with the unified 512-register file --
  * VGPRs: activation vector X (the trunk's even layers' input, feats), the two tile accumulators, temporaries;
  * AGPRs (numbered 256.. here): activation vector Y (odd layers' input, the two 256-wide head hidden vectors; written by
    v_accvgpr_write behind the cvt_pk), the A-fragment ring (ds_read_b128 straight into AGPRs), aux fragments, head accumulator;
  * 4 waves per workgroup, LDS ring of 144 pieces (1 KiB = one k-step of one 32-row output tile) fed by rows of 4 (one request per
    wave), one rendezvous per output tile (33 MFMAs; two tiles per rendezvous would need a ring of 5 tiles), A fragments PF MFMAs
    ahead, the epilogue of tile t-1 (16 v_sin, 8 v_cvt_pk, 8 v_accvgpr_write when the output vector lives in AGPRs) in the gaps of tile t.
save = 8 (the training forward): the PHASE8 byte of every sin stage's pre-activation (SDWA add into its byte of the store quad before
the in-place sine), MX8 feats + scale bytes, non-temporal stores in the MFMA gaps -- fwd_core.py's scheme with the unit numbers of
width 512 (mlp_layout.h: a_l at 16 l + t, feats 128, rgbh 144, s1 152, e1 160, s2 168, s3 176, scale unit 184, all + auxs).
5,370 MFMAs per 32 points (tau <= 8).  ``python fwd_core512.py`` writes csrc/mlp_fwd512_core_a{1,2}.inc and the clobber list;
tests/test_fwd_core.py checks they are current and executes the list on the lane-accurate model below against the emulator.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fwd_core import LANE, ROW_OF, Ins, aux_steps, bf16_bits, f32_to_frag, frag_to_f32  # noqa: E402,F401
from fwd_core3 import A0, rn  # noqa: E402

FEAT = 512
KS, HS, MT, MTH, NW = FEAT // 16, FEAT // 32, FEAT // 32, FEAT // 64, 4
X = 0                         # 128 VGPRs
ACC = (128, 144)
TMP = (160, 161, 162, 163)    # cvt_pk results on their way to AGPRs
SIG = 168
VL = (169, 170, 171)          # LDS read bases: ring + lane * 16 (+ 64 KiB, + 128 KiB)
VOFF = 172
N_VGPR = 173
SV = (174, 178)               # save = 8: two quads of store data (register tuples are 64-bit aligned on gfx950)
SOFF, KMAGIC, EB, MXT, K128 = 182, 183, 184, 188, 192   # workspace offset, PHASE8 constant, feats scale bytes (4 regs), MX8 temporaries (4), 128.0f
N_VGPR_SAVE = 193
Y = A0 + 0                    # 128 AGPRs; the head hidden vectors H0 / H1 are its halves
AR0, NA = A0 + 128, 6
AUX = A0 + 152                # 2 quads
HEAD = A0 + 160
N_AGPR = 176
IN_AUX, OUT_HEAD = 144, 128   # operands arrive / leave in VGPRs: aux fragments in v[144:151], head accumulator out in v[128:143]


class Tile:
    def __init__(self, p0, bregs, n_aux, acc, c0, epi, out, name, save_unit=None):
        self.save_unit = save_unit  # save = 8: workspace unit of this tile's 16 values per lane
        self.p0, self.n = p0, len(bregs)
        order = list(range(self.n - n_aux, self.n)) + list(range(self.n - n_aux))  # aux k-steps first
        self.pieces = [p0 + k for k in order]
        self.bregs = [bregs[k] for k in order]
        self.acc, self.c0, self.epi, self.out, self.name = acc, c0, epi, out, name


def stage_list(auxs):
    """the chunk list of FwdStream<AUXS> at SR_FEAT = 512 (mlp_layout.h) with this kernel's register assignment"""
    tiles, p, tno = [], 0, 0
    auxb = [AUX + 4 * a for a in range(auxs)]

    def dense(inp, ks, ntiles, epi, outbase, name, unit0=None):
        nonlocal p, tno
        for t in range(ntiles):
            b = [inp + 4 * k for k in range(ks)] + auxb
            out = [outbase + 8 * t + q for q in range(8)] if outbase is not None else None
            tiles.append(Tile(p, b, auxs, ACC[tno & 1], True, epi, out, f"{name}.{t}", None if unit0 is None else auxs + unit0 + t))
            p += len(b)
            tno += 1

    started = [False]

    def head(inp, with_aux, name):
        nonlocal p
        b = [inp + 4 * k for k in range(HS)] + (auxb if with_aux else [])
        tiles.append(Tile(p, b, auxs if with_aux else 0, HEAD, not started[0], None, None, name))
        started[0] = True
        p += len(b)

    H0, H1 = Y, Y + 4 * HS
    for l in range(7):
        dense(X if l % 2 == 0 else Y, KS, MT, "sin", Y if l % 2 == 0 else X, f"L{l + 1}", MT * (l + 1))
    dense(Y, KS, MT, "id", X, "feats", 8 * MT)
    dense(Y, KS, 1, "sigma", None, "sigma")
    dense(X, KS, MTH, "sin", H0, "rgbh", 9 * MT)
    head(H0, False, "Hr")
    dense(X, KS, MTH, "sin", H1, "s1", 9 * MT + MTH)
    dense(H1, HS, MTH, "sin", H0, "s2", 9 * MT + 3 * MTH)
    dense(H0, HS, MTH, "sin", H1, "s3", 9 * MT + 4 * MTH)
    head(H1, False, "Hs")
    dense(X, KS, MTH, "sin", H0, "e1", 9 * MT + 2 * MTH)
    head(H0, True, "Hb")
    return tiles, p


class Core512:
    def __init__(self, auxs, R=144, PF=5, GROUP=1, FILL=None, ablate=(), save=0):
        assert R % NW == 0 and R <= 192 and PF + 1 <= NA and save in (0, 8)
        if FILL is None:
            FILL = 3 if save else 2
        self.save = save
        self.auxs, self.R, self.PF, self.GROUP, self.FILL = auxs, R, PF, GROUP, FILL
        self.ablate = set(ablate)
        self.tiles, self.n_pieces = stage_list(auxs)
        self.ins = []
        self._build()

    def _e(self, op, a, text):
        self.ins.append(Ins(op, a, text))

    def mfma(self, acc, areg, breg, c0):
        c = "0" if c0 else rn(acc, 16)
        self._e("mfma", (acc, areg, breg, c0), f"MF {rn(acc, 16)}, {rn(areg, 4)}, {rn(breg, 4)}, {c}")

    def dsread(self, dst, slot):
        self._e("dsread", (dst, slot), f"ds_read_b128 {rn(dst, 4)}, v{VL[slot >> 6]} offset:{(slot & 63) * 1024}")

    def waitl(self, n):
        self._e("waitl", (n,), f"s_waitcnt lgkmcnt({n})")

    def dma_row(self, j, partial):
        imm = ((NW * j) % self.R) * 1024
        if partial is not None:
            self._e("dma_pred", (j, partial), f"s_cmp_lt_u32 %[wave], {partial}")
            self._e("dma_br", (j,), f"s_cbranch_scc0 .Lskip5_row{j}_%=")
        self._e("m0", (j,), f"s_add_u32 m0, %[wb], {imm}")
        self._e("nop", (0,), "s_nop 0")
        self._e("dma", (j, partial), f"global_load_lds_dwordx4 v{VOFF}, %[sb]")
        if partial is not None:
            self._e("label", (j,), f".Lskip5_row{j}_%=:")
        self._e("voff", (), f"v_add_u32 v{VOFF}, 0x1000, v{VOFF}")

    def _build(self):
        T, R, PF, G = self.tiles, self.R, self.PF, self.GROUP
        mf = [(ti, k) for ti, t in enumerate(T) for k in range(t.n)]
        N = len(mf)
        assert N == self.n_pieces
        last_of_tile = {}
        for i, (ti, k) in enumerate(mf):
            last_of_tile[ti] = i
        n_rows = (N + NW - 1) // NW
        partial_row = n_rows - 1 if N % NW else None
        partial_n = N % NW
        self.rows_issued = 0
        pending_rows = []

        def allow_rows(free_below):
            j = self.rows_issued + len(pending_rows)
            while j < n_rows and NW * (j + 1) - R <= free_below:
                pending_rows.append(j)
                j += 1

        def emit_row():
            j = pending_rows.pop(0)
            self.dma_row(j, partial_n if j == partial_row else None)
            self.rows_issued += 1

        def sync_for(first_tile):
            last = min(first_tile + G, len(T)) - 1
            need = (T[last].p0 + T[last].n + NW - 1) // NW
            while pending_rows:
                emit_row()
            issued = self.rows_issued
            assert issued >= need, (first_tile, issued, need)
            vm = issued - need
            if partial_row is not None and issued > partial_row and need <= partial_row:
                vm -= 1
            vm = max(vm, 0)
            assert vm <= 63
            self._e("sync", (vm, need), f"s_waitcnt vmcnt({vm})")
            self._e("barrier", (), "s_barrier")

        self.n_saves, self.cur_unit = 0, 0
        epi_q = []  # [earliest MFMA index, kind, args, registers written (hazard tracking / forced flushes), accumulator read or None]
        written_at, trans_at = {}, {}

        def queue_epilogue(ti):
            t = T[ti]
            g0, a = last_of_tile[ti] + 2, t.acc
            if t.epi == "sigma":
                epi_q.append([g0, "mov", (SIG, a), {SIG}, a])
                return
            sin = t.epi == "sin"
            to_agpr = t.out[0] >= A0

            def out_pk(q):  # cvt_pk of values 2 q, 2 q + 1 into the output vector (through a temporary when it lives in AGPRs)
                d = TMP[q & 3] if to_agpr else t.out[q]
                epi_q.append([g0, "pk", (d, a + 2 * q, a + 2 * q + 1), set() if to_agpr else {d}, a])
                if to_agpr:
                    epi_q.append([g0, "accw", (t.out[q], d), {t.out[q]}, None])
            if self.save and t.save_unit is not None:
                sv = SV[self.n_saves & 1]
                self.n_saves += 1
                if sin:
                    # PHASE8: byte = low mantissa byte of (pre-activation + 1.5 * 2^15), written straight into its place of the store quad
                    # by an SDWA add BEFORE the value's sin overwrites the accumulator; cvt_pk one pair behind (trans -> VALU use)
                    for q in range(8):
                        for g in (2 * q, 2 * q + 1):
                            epi_q.append([g0, "phase", (sv + (g >> 2), g & 3, a + g), set(), a])
                            epi_q.append([g0, "sin", (a + g,), set(), a])
                        if q > 0:
                            out_pk(q - 1)
                    epi_q.append([g0, "store", (sv, t.save_unit), set(), None])
                    out_pk(7)
                else:
                    # MX8: E = exponent of 1.0079 max|v| clamped to [6, 254]; u = cvt_pk_u8(v * 2^(133 - E) + 128); byte t of EB = E
                    tt = t.save_unit - (self.auxs + 8 * MT)
                    epi_q.append([g0, "mx_max", (a, 0, True), set(), a])
                    for g in range(2, 16, 2):
                        epi_q.append([g0, "mx_max", (a, g, False), set(), a])
                    epi_q.append([g0, "mx_exp", (tt,), set(), None])
                    for g in range(16):
                        epi_q.append([g0, "mx_q", (sv + (g >> 2), g & 3, a + g), set(), a])
                    epi_q.append([g0, "store", (sv, t.save_unit), set(), None])
                    for q in range(8):
                        out_pk(q)
                    if tt == MT - 1:
                        epi_q.append([g0, "store_scale", (self.auxs + 9 * MT + 5 * MTH,), set(), None])
                return
            if sin:
                epi_q.append([g0, "sin", (a + 0,), set(), a])
                epi_q.append([g0, "sin", (a + 1,), set(), a])
            for q in range(8):
                v0, v1 = a + 2 * q, a + 2 * q + 1
                if sin and q < 7:
                    epi_q.append([g0, "sin", (v0 + 2,), set(), a])
                d = TMP[q & 3] if to_agpr else t.out[q]
                epi_q.append([g0, "pk", (d, v0, v1), set() if to_agpr else {d}, a])
                if sin and q < 7:
                    epi_q.append([g0, "sin", (v1 + 2,), set(), a])
                if to_agpr:
                    epi_q.append([g0, "accw", (t.out[q], d), {t.out[q]}, None])

        def emit_epi(item):
            _, kind, args, _, _ = item
            n = len(self.ins)
            if kind == "sin":
                self._e("sin", args, f"v_sin_f32 v{args[0]}, v{args[0]}")
                trans_at[args[0]] = n
            elif kind == "pk":
                d, s0, s1 = args
                for s in (s0, s1):
                    if s in trans_at and len(self.ins) - trans_at[s] < 2:
                        self._e("nop", (0,), "s_nop 0")
                self._e("pk", args, f"PK v{d}, v{s0}, v{s1}")
            elif kind == "accw":
                self._e("accw", args, f"v_accvgpr_write_b32 {rn(args[0])}, v{args[1]}")
            elif kind == "mov":
                self._e("mov", args, f"v_mov_b32 v{args[0]}, v{args[1]}")
            elif kind == "phase":
                d, byte, src = args
                self._e("phase", args, f"v_add_f32_sdwa v{d}, v{src}, v{KMAGIC} dst_sel:BYTE_{byte} dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
            elif kind in ("store", "store_scale"):
                reg, unit = (args[0], args[1]) if kind == "store" else (EB, args[0])
                assert reg % 2 == 0
                delta = (unit - self.cur_unit) * 1024
                self.cur_unit = unit
                self._e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self._e("store", (reg, unit), f"global_store_dwordx4 v{SOFF}, v[{reg}:{reg + 3}], %[ab]" + (" nt" if kind == "store" else ""))
            elif kind == "mx_max":
                a_, g, first = args
                if first:
                    self._e("mx_max", (MXT, a_ + g, a_ + g + 1, None), f"v_max_f32 v{MXT}, |v{a_ + g}|, |v{a_ + g + 1}|")
                else:
                    self._e("mx_max", (MXT, a_ + g, a_ + g + 1, MXT), f"v_max3_f32 v{MXT}, |v{a_ + g}|, |v{a_ + g + 1}|, v{MXT}")
            elif kind == "mx_exp":
                (tt,) = args
                m_, e, inv = MXT, MXT + 1, MXT + 2
                self._e("mx_e1", (m_,), f"v_fmac_f32 v{m_}, 0x3c000000, v{m_}")
                self._e("mx_e2", (e, m_), f"v_lshrrev_b32 v{e}, 23, v{m_}")
                self._e("mx_e3a", (e,), f"v_max_u32 v{e}, 6, v{e}")
                self._e("mx_e3", (e,), f"v_min_u32 v{e}, 0xfe, v{e}")
                self._e("mx_e4", (inv, e), f"v_sub_u32 v{inv}, 0x104, v{e}")
                self._e("mx_e5", (inv,), f"v_lshlrev_b32 v{inv}, 23, v{inv}")
                if tt & 3:
                    self._e("mx_e6", (EB + (tt >> 2), e, 8 * (tt & 3), False), f"v_lshl_or_b32 v{EB + (tt >> 2)}, v{e}, {8 * (tt & 3)}, v{EB + (tt >> 2)}")
                else:
                    self._e("mx_e6", (EB + (tt >> 2), e, 0, True), f"v_mov_b32 v{EB + (tt >> 2)}, v{e}")
            elif kind == "mx_q":
                d, byte, src = args
                self._e("mx_q1", (MXT + 3, src, MXT + 2), f"v_fma_f32 v{MXT + 3}, v{src}, v{MXT + 2}, v{K128}")
                self._e("mx_q2", (d, MXT + 3, byte), f"v_cvt_pk_u8_f32 v{d}, v{MXT + 3}, {byte}, v{d}")
            for r in item[3]:
                written_at[r] = len(self.ins) - 1

        def flush_producers(regs, i):
            last = -1
            for qi, it in enumerate(epi_q):
                if it[3] & regs:
                    last = qi
            for _ in range(last + 1):
                it = epi_q.pop(0)
                assert it[0] <= i + 1, ("epilogue needed before its accumulator is ready", it, i)
                emit_epi(it)
            return last + 1

        def flush_acc(acc):
            last = -1
            for qi, it in enumerate(epi_q):
                if it[4] == acc:
                    last = qi
            for _ in range(last + 1):
                emit_epi(epi_q.pop(0))
            return last + 1

        # ---- preamble: the aux fragments move to AGPRs, the first rows are requested, the A pipeline starts
        self._e("savem0", (), "s_mov_b32 %[m0save], m0")
        for i in range(4 * self.auxs):
            self._e("accw", (AUX + i, IN_AUX + i), f"v_accvgpr_write_b32 {rn(AUX + i)}, v{IN_AUX + i}")
        allow_rows(0)
        sync_done_for = -1
        forced = 0

        def read_for(i):
            nonlocal sync_done_for
            ti, k = mf[i]
            if ti > sync_done_for and ti % G == 0 and k == 0:
                sync_for(ti)
                sync_done_for = ti + G - 1
                return True
            return False

        for i in range(min(PF, N)):
            read_for(i)
            ti, k = mf[i]
            self.dsread(AR0 + 4 * (i % NA), T[ti].pieces[k] % R)
        for i in range(N):
            ti, k = mf[i]
            t = T[ti]
            breg = t.bregs[k]
            if k == 0 and t.c0:
                forced += flush_acc(t.acc)
            forced += flush_producers(set(range(breg, breg + 4)), i - 1)
            dist = len(self.ins) - max(written_at.get(r, -10) for r in range(breg, breg + 4))
            if dist < 3:
                self._e("nop", (3 - dist,), f"s_nop {3 - dist}")
            self.waitl(min(PF - 1, N - 1 - i))
            self.mfma(t.acc, AR0 + 4 * (i % NA), breg, t.c0 and k == 0)
            if k == t.n - 1 and t.epi is not None:
                queue_epilogue(ti)
            if i + PF < N:
                if read_for(i + PF):
                    allow_rows(T[ti].p0)
                tj, kj = mf[i + PF]
                self.dsread(AR0 + 4 * ((i + PF) % NA), T[tj].pieces[kj] % R)
            if pending_rows:
                emit_row()
            n = 0
            while epi_q and n < self.FILL and epi_q[0][0] <= i:
                emit_epi(epi_q.pop(0))
                n += 1
        assert not epi_q and not pending_rows and self.rows_issued == n_rows
        self._e("nop", (15,), "s_nop 15")
        self._e("nop", (3,), "s_nop 3")
        for g in range(16):
            self._e("accr", (OUT_HEAD + g, HEAD + g), f"v_accvgpr_read_b32 v{OUT_HEAD + g}, {rn(HEAD + g)}")
        self._e("restm0", (), "s_mov_b32 m0, %[m0save]")
        self.stats = dict(mfma=N, instructions=len(self.ins), forced_epilogue=forced, rows=n_rows,
                          barriers=sum(1 for x in self.ins if x.op == "barrier"), nops=sum(1 for x in self.ins if x.op == "nop"))

    def text(self):
        ab, drop = self.ablate, set()
        if "nodma" in ab:
            drop |= {"m0", "dma", "voff", "dma_pred", "dma_br", "label"}
        if "nobarrier" in ab:
            drop |= {"barrier"}
        if "noepi" in ab:
            drop |= {"sin", "pk"}
        if "nostore" in ab:
            drop |= {"store", "soff"}
        out, seen = [], False
        for x in self.ins:
            if x.op == "sync":
                seen = True
            if x.op in drop and (seen or x.op not in ("m0", "dma", "voff")):
                continue
            out.append(x.text)
        return out

    def inc_file(self):
        s = self.stats
        lines = ["// GENERATED by csrc/gen/fwd_core512.py -- do not edit (tests/test_fwd_core.py checks it is current).",
                 f"// width-512 forward core, AUXS = {self.auxs}: {s['mfma']} MFMAs, {s['instructions']} instructions, {s['barriers']} rendezvous, "
                 f"{s['rows']} LDS-DMA rows, ring of {self.R} pieces, A fragments {self.PF} ahead.",
                 "// Operands: %[sb] stream base (SGPR pair), %[wb] LDS ring address + wave * 1024, %[wave] wave index, %[m0save] scratch SGPR"
                 + (", %[ab] activation workspace (SGPR pair)." if self.save else ".")]
        lines += ['"' + t + '\\n"' for t in self.text()]
        return "\n".join(lines) + "\n"

    @staticmethod
    def clobber_file(save=0):
        # operands: v[0:127] (X in), v[128:143] (head out), v[144:151] (aux in), SIG, VL, VOFF[, KMAGIC, SOFF, K128]
        regs = [f"v{r}" for r in list(range(ACC[1], ACC[1] + 16)) + list(TMP) if not IN_AUX <= r < IN_AUX + 8]
        if save:
            regs += [f"v{r}" for r in range(SV[0], N_VGPR_SAVE) if r not in (KMAGIC, SOFF, K128)]
        regs += [f"a{r}" for r in range(N_AGPR)]
        return ("// GENERATED by csrc/gen/fwd_core512.py: clobber list of the width-512 forward core\n" + ", ".join(f'"{r}"' for r in regs)
                + ', "memory", "scc"\n')


class Machine512:
    """Lane-accurate execution of one wave's instruction list on the unified 512-register file, with fwd_core.Machine's ring checks."""

    def __init__(self, core, stream_bits):
        self.c = core
        self.stream = stream_bits  # [n_pieces, 64, 4] uint32
        self.v = np.zeros((512, 64), np.uint32)
        self.ring_piece = [-1] * core.R
        self.ring = np.zeros((core.R, 64, 4), np.uint32)
        self.synced_rows = 0
        self.consumed = np.zeros(core.n_pieces, bool)
        self.consumed_before_barrier = np.zeros(core.n_pieces, bool)
        self.pending = []
        self.ar = {}
        self.issued = {"full": [], "skip": []}
        self.last_write = {}
        self.voff_rows = 0
        self.soff = 0
        self.stores = {}
        self._xdl, self._n_mfma, self._nops = {}, 0, 0
        self._trans, self._pc, self._m0_at = {}, 0, -9  # v_sin results, instruction index, last M0 write

    def f(self, r):
        return self.v[r].view(np.float32)

    def setf(self, r, x):
        self.v[r] = np.asarray(x, np.float32).view(np.uint32)

    # XDL write -> VALU read: the generators start a tile's epilogue two MFMAs after its last one (each later MFMA holds the issue port for
    # 8 wait states; 11 are needed after an 8-pass MFMA) or behind an s_nop chain; checked for every VALU source register
    def _mfma_wrote(self, acc):
        for g in range(16):
            self._xdl[acc + g] = self._n_mfma
        self._n_mfma += 1
        self._nops = 0

    def _valu_reads(self, *regs):
        for r in regs:
            k = self._xdl.get(r)
            assert k is None or self._n_mfma - 1 - k >= 2 or self._nops >= 12, ("VALU reads an MFMA result too early", r)
            assert self._pc - self._trans.get(r, -9) >= 2, ("trans result used by the next instruction", r)  # trans -> VALU: 1 wait state

    def run(self):
        c = self.c
        for n, ins in enumerate(c.ins):
            op, a = ins.op, ins.a
            self._pc = n
            if op == "m0":
                self._m0_at = n
            elif op == "dma":
                assert n - self._m0_at >= 2, "M0 write -> LDS-DMA needs one wait state"
            if op == "mfma":
                acc, areg, breg, c0 = a
                assert not any(d == areg for d, _, _ in self.pending), "MFMA reads an A fragment still in flight"
                for r in range(breg, breg + 4):
                    assert n - self.last_write.get(r, -10) >= 3, ("VALU write -> MFMA operand hazard", ins.text)
                piece = self.ar[areg]
                assert not self.consumed[piece]
                self.consumed[piece] = True
                A = frag_to_f32(self.v[areg:areg + 4]).astype(np.float64)
                B = frag_to_f32(self.v[breg:breg + 4]).astype(np.float64)
                Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
                for h in range(2):
                    Am[:, 8 * h:8 * h + 8] = A[32 * h:32 * h + 32]
                    Bm[8 * h:8 * h + 8, :] = B[32 * h:32 * h + 32].T
                d = (Am @ Bm)[ROW_OF, (LANE & 31)[:, None]]
                for g in range(16):
                    prev = np.zeros(64, np.float32) if c0 else self.f(acc + g).copy()
                    self.setf(acc + g, (prev.astype(np.float64) + d[:, g]).astype(np.float32))
                self._mfma_wrote(acc)
            elif op == "nop":
                self._nops += a[0] + 1
            elif op == "dsread":
                dst, slot = a
                piece = self.ring_piece[slot]
                assert piece >= 0 and piece // NW < self.synced_rows, ("piece read before the rendezvous that covers its row", piece, self.synced_rows)
                self.pending.append((dst, slot, piece))
            elif op == "waitl":
                while len(self.pending) > a[0]:
                    dst, slot, piece = self.pending.pop(0)
                    assert self.ring_piece[slot] == piece
                    self.v[dst:dst + 4] = self.ring[slot].T
                    self.ar[dst] = piece
            elif op == "sync":
                vm, need = a
                for cls, rows in self.issued.items():
                    landed = set(rows[:max(len(rows) - vm, 0)])
                    missing = [r for r in range(need) if r in rows and r not in landed]
                    assert not missing, ("vmcnt lets a needed row stay in flight", cls, vm, need, missing)
                self.synced_rows = max(self.synced_rows, need)
            elif op == "barrier":
                self.consumed_before_barrier = self.consumed.copy()
            elif op == "dma":
                j, partial = a
                assert self.voff_rows == j
                self.issued["full"].append(j)
                if partial is None:
                    self.issued["skip"].append(j)
                for w in range(NW if partial is None else partial):
                    p = NW * j + w
                    slot = p % c.R
                    old = self.ring_piece[slot]
                    assert old < 0 or self.consumed_before_barrier[old], ("DMA overwrites a piece not yet consumed by every wave", old, p)
                    self.ring_piece[slot] = p
                    self.ring[slot] = self.stream[p]
            elif op == "voff":
                self.voff_rows += 1
            elif op == "sin":
                self._valu_reads(a[0])
                self._trans[a[0]] = self._pc
                self.setf(a[0], np.sin(2 * np.pi * self.f(a[0]).astype(np.float64)))
            elif op == "pk":
                d, s0, s1 = a
                self._valu_reads(s0, s1)
                self.v[d] = (bf16_bits(self.f(s0)) | (bf16_bits(self.f(s1)) << 16)).astype(np.uint32)
                self.last_write[d] = n
            elif op in ("accw", "accr", "mov"):
                self._valu_reads(a[1])
                self.v[a[0]] = self.v[a[1]]
                self.last_write[a[0]] = n
            elif op == "phase":
                d, byte, src = a
                self._valu_reads(src)
                b = (self.f(src) + self.f(KMAGIC)).astype(np.float32).view(np.uint32) & np.uint32(0xFF)
                self.v[d] = (self.v[d] & np.uint32(~(0xFF << (8 * byte)) & 0xFFFFFFFF)) | (b << np.uint32(8 * byte))
            elif op == "soff":
                self.soff += a[0]
            elif op == "store":
                reg, unit = a
                assert self.soff == unit * 1024 and unit not in self.stores, (self.soff, unit)
                self.stores[unit] = self.v[reg:reg + 4].copy()
            elif op == "mx_max":
                m_, x, y, z = a
                self._valu_reads(x, y)
                r = np.maximum(np.abs(self.f(x)), np.abs(self.f(y)))
                self.setf(m_, r if z is None else np.maximum(r, self.f(z)))
            elif op == "mx_e1":
                self.setf(a[0], self.f(a[0]).astype(np.float64) * 0.0078125 + self.f(a[0]).astype(np.float64))
            elif op == "mx_e2":
                self.v[a[0]] = self.v[a[1]] >> np.uint32(23)
            elif op == "mx_e3":
                self.v[a[0]] = np.clip(self.v[a[0]], 6, 254).astype(np.uint32)
            elif op == "mx_e4":
                self.v[a[0]] = (np.uint32(260) - self.v[a[1]]).astype(np.uint32)
            elif op == "mx_e5":
                self.v[a[0]] = (self.v[a[0]] << np.uint32(23)).astype(np.uint32)
            elif op == "mx_e6":
                d, e, sh, first = a
                self.v[d] = self.v[e].copy() if first else ((self.v[e] << np.uint32(sh)) | self.v[d]).astype(np.uint32)
            elif op == "mx_q1":
                t_, src, inv = a
                self._valu_reads(src)
                self.setf(t_, self.f(src).astype(np.float64) * self.f(inv).astype(np.float64) + self.f(K128).astype(np.float64))
            elif op == "mx_q2":
                d, t_, byte = a
                b = np.clip(np.rint(self.f(t_).astype(np.float64)), 0, 255).astype(np.uint32)
                self.v[d] = (self.v[d] & np.uint32(~(0xFF << (8 * byte)) & 0xFFFFFFFF)) | (b << np.uint32(8 * byte))
        assert self.consumed.all()


def main():
    import argparse

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ablate", default="")
    ap.add_argument("--PF", type=int, default=5)
    ap.add_argument("--GROUP", type=int, default=1)
    ap.add_argument("--FILL", type=int, default=None)
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = a.out or os.path.dirname(here)
    for auxs in (1, 2):
        for save in (0, 8):
            c = Core512(auxs, PF=a.PF, GROUP=a.GROUP, FILL=a.FILL, ablate=[x for x in a.ablate.split(",") if x], save=save)
            path = os.path.join(out_dir, f"mlp_fwd512_core_a{auxs}{'s8' if save else ''}.inc")
            with open(path, "w") as f:
                f.write(c.inc_file())
            print(path, c.stats)
    for save, name in ((0, "mlp_fwd512_core_clobbers.inc"), (8, "mlp_fwd512_core_clobbers_s8.inc")):
        with open(os.path.join(out_dir, name), "w") as f:
            f.write(Core512.clobber_file(save))


if __name__ == "__main__":
    sys.exit(main())
