#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of the PARITY-mode (bf16x3) forward core (csrc/mlp_fwd3.inc).

Same method as fwd_core.py (read that docstring first), other machine shape: in the parity mode every operand is a pair of bf16
planes (hi = RNE(v), lo = RNE(v - hi)) and every k-step is three MFMAs -- A_lo x B_hi, A_hi x B_lo, A_hi x B_hi, small terms first --
so a wave's activations are 4 x 64 registers.  The wave therefore runs alone on its SIMD with the unified 512-register file:
  * architectural VGPRs: the hi planes of both activation vectors (VALU writes them), the two tile accumulators (VALU reads them),
    epilogue temporaries;
  * accumulation VGPRs (a0..a255, numbered 256.. here): the lo planes (written by v_accvgpr_write), the A-fragment ring (ds_read_b128
    straight into AGPRs), the aux fragments, the head accumulator.  MFMA takes A / B / C / D from either file.
  * 4 waves per workgroup; a "unit" = one k-step of one output tile = 1 KiB of the hi stream + 1 KiB of the lo stream (the two packed
    streams of packing.py, same piece order); LDS ring of R units (2 KiB slots: hi, lo) fed by rows of 4 units, two 1-KiB requests per
    wave and row; one rendezvous per output tile (51 MFMAs), placed PF k-steps before the first read that needs it.
  * epilogue of tile t-1 in the gaps of tile t: v_fract + v_sin (the hardware sine on the exact fraction: 1.6e-6 of the reference end
    to end, profiles/r03_ab_variants.txt), hi = cvt_pk, lo = cvt_pk(v - float(hi)) -> v_accvgpr_write: 88 VALU per 51 MFMAs.
  * save = 16 (training in the parity mode, SR_FMT16 workspaces of mlp_layout.h): the phase of every sin stage's pre-activation as
    unorm16 (v_cvt_pknorm_u16 of the exact fractions, before the in-place sine) and the bf16 hi plane of feats, two non-temporal
    1-KiB-per-wave stores per tile in the MFMA gaps.
``python fwd_core3.py`` writes csrc/mlp_fwd3_core_a{1,2}.inc and mlp_fwd3_core_clobbers.inc; tests/test_fwd_core.py checks they are
current and executes the list on the lane-accurate model below against the fp64 emulator.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fwd_core import LANE, ROW_OF, Ins, aux_steps, bf16_bits, bf16_to_f32, f32_to_frag, frag_to_f32  # noqa: E402,F401

A0 = 256                      # unified numbering: 0..255 = v, 256..511 = a
XH, YH = 0, 64                # hi planes (VGPR)
ACC = (128, 144)
TMP = (160, 164)              # two sets of epilogue temporaries (3 used of each)
SIG = 168
VL0, VL1, VOFF = 169, 170, 171
N_VGPR = 172
SV = (172, 180)               # save = 16: two sets of two store quads (unorm16 phases of a tile)
SOFF = 188                    # this lane's workspace offset (tile base + lane * 16, advanced fragment by fragment)
N_VGPR_SAVE = 189
XL, YL = A0 + 0, A0 + 64      # lo planes (AGPR)
AR0, NA = A0 + 128, 4         # A-fragment ring: NA k-steps x (lo quad, hi quad)
AUXH, AUXL = A0 + 160, A0 + 168
HEAD = A0 + 176
N_AGPR = 192
# operands arrive in VGPRs and are moved by the stream's first instructions: X lo plane in v[64:127], aux hi / lo in v[144:151] / v[152:159]
IN_XL, IN_AUXH, IN_AUXL = 64, 144, 152
OUT_HEAD = 128                # the head accumulator is copied to v[128:143] at the end
KS, HS, NW = 16, 8, 4


def rn(r, n=1):
    f, i = ("v", r) if r < A0 else ("a", r - A0)
    return f"{f}{i}" if n == 1 else f"{f}[{i}:{i + n - 1}]"


class Tile:
    def __init__(self, p0, bh, bl, n_aux, acc, c0, epi, out_h, out_l, name, save_frag=None):
        self.save_frag = save_frag  # SR_FMT16 workspace fragment (1 KiB per tile of 32 points) of this tile's values 0..7; 8..15 go to the next one
        self.p0, self.n = p0, len(bh)
        order = list(range(self.n - n_aux, self.n)) + list(range(self.n - n_aux))  # aux k-steps first
        self.units = [p0 + k for k in order]
        self.bh, self.bl = [bh[k] for k in order], [bl[k] for k in order]
        self.acc, self.c0, self.epi, self.out_h, self.out_l, self.name = acc, c0, epi, out_h, out_l, name


def stage_list(auxs):
    tiles, p, tno = [], 0, 0
    auxh = [AUXH + 4 * a for a in range(auxs)]
    auxl = [AUXL + 4 * a for a in range(auxs)]

    def dense(ih, il, ks, ntiles, epi, oh, ol, name, frag0=None):
        nonlocal p, tno
        for t in range(ntiles):
            bh = [ih + 4 * k for k in range(ks)] + auxh
            bl = [il + 4 * k for k in range(ks)] + auxl
            out_h = [oh + 8 * t + q for q in range(8)] if oh is not None else None
            out_l = [ol + 8 * t + q for q in range(8)] if ol is not None else None
            tiles.append(Tile(p, bh, bl, auxs, ACC[tno & 1], True, epi, out_h, out_l, f"{name}.{t}", None if frag0 is None else auxs + frag0 + 2 * t))
            p += len(bh)
            tno += 1

    started = [False]

    def head(ih, il, with_aux, name):
        nonlocal p
        bh = [ih + 4 * k for k in range(HS)] + (auxh if with_aux else [])
        bl = [il + 4 * k for k in range(HS)] + (auxl if with_aux else [])
        tiles.append(Tile(p, bh, bl, auxs if with_aux else 0, HEAD, not started[0], None, None, None, name))
        started[0] = True
        p += len(bh)

    for l in range(7):
        a, b = ((XH, XL), (YH, YL)) if l % 2 == 0 else ((YH, YL), (XH, XL))
        dense(a[0], a[1], KS, 8, "sin", b[0], b[1], f"L{l + 1}", 16 * (l + 1))   # mlp_layout.h: a_l at fragment 16 l (+ auxs)
    dense(YH, YL, KS, 8, "id", XH, XL, "feats", 128)
    dense(YH, YL, KS, 1, "sigma", None, None, "sigma")
    H0, H1 = (YH, YL), (YH + 32, YL + 32)
    dense(XH, XL, KS, 4, "sin", H0[0], H0[1], "rgbh", 144)
    head(H0[0], H0[1], False, "Hr")
    dense(XH, XL, KS, 4, "sin", H1[0], H1[1], "s1", 152)
    dense(H1[0], H1[1], HS, 4, "sin", H0[0], H0[1], "s2", 168)
    dense(H0[0], H0[1], HS, 4, "sin", H1[0], H1[1], "s3", 176)
    head(H1[0], H1[1], False, "Hs")
    dense(XH, XL, KS, 4, "sin", H0[0], H0[1], "e1", 160)
    head(H0[0], H0[1], True, "Hb")
    return tiles, p


class Core3:
    def __init__(self, auxs, R=64, PF=2, FILL=2, ablate=(), save=0):
        assert R % NW == 0 and R <= 64 and PF + 1 <= NA and save in (0, 16)
        self.auxs, self.R, self.PF, self.FILL, self.save = auxs, R, PF, FILL, save
        self.ablate = set(ablate)
        self.tiles, self.n_units = stage_list(auxs)
        self.ins = []
        self._build()

    def _e(self, op, a, text):
        self.ins.append(Ins(op, a, text))

    # ---- emission helpers
    def mfma(self, acc, areg, breg, c0):
        c = "0" if c0 else rn(acc, 16)
        self._e("mfma", (acc, areg, breg, c0), f"v_mfma_f32_32x32x16_bf16 {rn(acc, 16)}, {rn(areg, 4)}, {rn(breg, 4)}, {c}")

    def dsread(self, dst, slot, plane):
        base, off = (VL0, slot * 2048) if slot < 32 else (VL1, (slot - 32) * 2048)
        self._e("dsread", (dst, slot, plane), f"ds_read_b128 {rn(dst, 4)}, v{base} offset:{off + 1024 * plane}")

    def waitl(self, n):
        self._e("waitl", (n,), f"s_waitcnt lgkmcnt({n})")

    def dma_half(self, j, plane, partial):
        """one of the two requests of row j (wave w fetches plane `plane` of unit 4 j + w into its ring slot); a ragged last row is
        emitted whole under one branch"""
        imm = ((NW * j) % self.R) * 2048
        if partial is not None and plane == 0:
            self._e("dma_pred", (j, partial), f"s_cmp_lt_u32 %[wave], {partial}")
            self._e("dma_br", (j,), f"s_cbranch_scc0 .Lskip3_row{j}_%=")
        self._e("m0", (j, plane), f"s_add_u32 m0, %[wb], {imm + 1024 * plane}")
        self._e("nop", (0,), "s_nop 0")
        self._e("dma", (j, plane, partial), f"global_load_lds_dwordx4 v{VOFF}, %[{'sl' if plane else 'sh'}]")
        if plane == 1:
            if partial is not None:
                self._e("label", (j,), f".Lskip3_row{j}_%=:")
            self._e("voff", (), f"v_add_u32 v{VOFF}, 0x1000, v{VOFF}")

    # ---- schedule
    def _build(self):
        T, R, PF = self.tiles, self.R, self.PF
        ks = [(ti, k) for ti, t in enumerate(T) for k in range(t.n)]
        N = len(ks)
        assert N == self.n_units
        last_of_tile = {}
        for i, (ti, k) in enumerate(ks):
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

        half = [0]  # plane of the next request of pending_rows[0]

        def emit_half():
            j = pending_rows[0]
            part = partial_n if j == partial_row else None
            self.dma_half(j, half[0], part)
            if half[0] == 0 and part is not None:  # ragged row: both requests under one branch
                self.dma_half(j, 1, part)
                half[0] = 1
            half[0] ^= 1
            if half[0] == 0:
                pending_rows.pop(0)
                self.rows_issued += 1

        def emit_row():
            emit_half()
            if half[0]:
                emit_half()

        def sync_for(ti):
            need = (T[ti].p0 + T[ti].n + NW - 1) // NW
            while pending_rows:
                emit_half()
            issued = self.rows_issued
            assert issued >= need, (ti, issued, need)
            vm = 2 * (issued - need)
            if partial_row is not None and issued > partial_row and need <= partial_row:
                vm -= 2
            vm = max(vm, 0)
            assert vm <= 63
            self._e("sync", (vm, need), f"s_waitcnt vmcnt({vm})")
            self._e("barrier", (), "s_barrier")

        self.n_saves, self.cur_frag = 0, 0
        epi_q = []  # items: [earliest MFMA index, kind, args, writes (set of registers), reads_acc (accumulator base or None)]
        written_at, trans_at = {}, {}

        def queue_epilogue(ti, m_last):
            t = T[ti]
            g0, a = m_last + 2, t.acc
            if t.epi == "sigma":
                epi_q.append([g0, "mov", (SIG, a), {SIG}, a])
                return
            sin = t.epi == "sin"
            saving = self.save and t.save_frag is not None
            sv = None
            if saving and sin:
                sv = SV[self.n_saves & 1]
                self.n_saves += 1

            def phase(q):  # unorm16 phases of values 2 q, 2 q + 1 (exact fractions) -> word q of the tile's two store quads
                if sv is not None:
                    epi_q.append([g0, "pknorm", (sv + q, a + 2 * q, a + 2 * q + 1), set(), a])
                    if q in (3, 7):
                        epi_q.append([g0, "store", (sv + (q - 3), t.save_frag + (q >> 2)), set(), None])
            if sin:
                for g in range(16):  # the exact fraction first: v_sin's own range reduction is not trusted at parity tolerances
                    epi_q.append([g0, "fract", (a + g,), set(), a])
                phase(0)
                epi_q.append([g0, "sin", (a + 0,), set(), a])
                epi_q.append([g0, "sin", (a + 1,), set(), a])
            for q in range(8):
                v0, v1 = a + 2 * q, a + 2 * q + 1
                t0, t1, t2 = (TMP[q & 1] + j for j in range(3))
                if sin and q < 7:
                    phase(q + 1)
                    epi_q.append([g0, "sin", (v0 + 2,), set(), a])
                epi_q.append([g0, "pk", (t.out_h[q], v0, v1), {t.out_h[q]}, a])
                if saving and not sin and q in (3, 7):  # identity stage: the bf16 hi plane itself
                    epi_q.append([g0, "store", (t.out_h[q - 3], t.save_frag + (q >> 2)), set(), None])
                if sin and q < 7:
                    epi_q.append([g0, "sin", (v1 + 2,), set(), a])
                epi_q.append([g0, "shl", (t0, t.out_h[q]), set(), None])
                epi_q.append([g0, "and", (t1, t.out_h[q]), set(), None])
                epi_q.append([g0, "sub", (v0, v0, t0), set(), a])
                epi_q.append([g0, "sub", (v1, v1, t1), set(), a])
                epi_q.append([g0, "pk", (t2, v0, v1), set(), a])
                epi_q.append([g0, "accw", (t.out_l[q], t2), {t.out_l[q]}, None])

        def emit_epi(item):
            _, kind, args, _, _ = item
            n = len(self.ins)
            if kind == "fract":
                self._e("fract", args, f"v_fract_f32 v{args[0]}, v{args[0]}")
            elif kind == "sin":
                self._e("sin", args, f"v_sin_f32 v{args[0]}, v{args[0]}")
                trans_at[args[0]] = n
            elif kind == "pk":
                d, s0, s1 = args
                for s in (s0, s1):
                    if s in trans_at and len(self.ins) - trans_at[s] < 2:
                        self._e("nop", (0,), "s_nop 0")
                self._e("pk", args, f"v_cvt_pk_bf16_f32 v{d}, v{s0}, v{s1}")
            elif kind == "shl":
                self._e("shl", args, f"v_lshlrev_b32 v{args[0]}, 16, v{args[1]}")
            elif kind == "and":
                self._e("and", args, f"v_and_b32 v{args[0]}, 0xffff0000, v{args[1]}")
            elif kind == "sub":
                d, x, y = args
                if x in trans_at and len(self.ins) - trans_at[x] < 2:
                    self._e("nop", (0,), "s_nop 0")
                self._e("sub", args, f"v_sub_f32 v{d}, v{x}, v{y}")
            elif kind == "accw":
                self._e("accw", args, f"v_accvgpr_write_b32 {rn(args[0])}, v{args[1]}")
            elif kind == "pknorm":
                self._e("pknorm", args, f"v_cvt_pknorm_u16_f32 v{args[0]}, v{args[1]}, v{args[2]}")
            elif kind == "store":
                reg, frag = args
                delta = (frag - self.cur_frag) * 1024
                self.cur_frag = frag
                self._e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self._e("store", (reg, frag), f"global_store_dwordx4 v{SOFF}, v[{reg}:{reg + 3}], %[ab] nt")
            elif kind == "mov":
                self._e("mov", args, f"v_mov_b32 v{args[0]}, v{args[1]}")
            for r in item[3]:
                written_at[r] = len(self.ins) - 1

        def flush_producers(regs, m):
            last = -1
            for qi, it in enumerate(epi_q):
                if it[3] & regs:
                    last = qi
            for _ in range(last + 1):
                it = epi_q.pop(0)
                assert it[0] <= m + 1, ("epilogue needed before its accumulator is ready", it, m)
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

        def operand_ready(breg):
            recent = max(written_at.get(r, -10) for r in range(breg, breg + 4))
            dist = len(self.ins) - recent
            if dist < 3:
                self._e("nop", (3 - dist,), f"s_nop {3 - dist}")

        # ---- preamble: move the operands that live in AGPRs, request the first rows, start the A pipeline
        self._e("savem0", (), "s_mov_b32 %[m0save], m0")
        for i in range(64):
            self._e("accw", (XL + i, IN_XL + i), f"v_accvgpr_write_b32 {rn(XL + i)}, v{IN_XL + i}")
        for a in range(self.auxs):
            for i in range(4):
                self._e("accw", (AUXH + 4 * a + i, IN_AUXH + 4 * a + i), f"v_accvgpr_write_b32 {rn(AUXH + 4 * a + i)}, v{IN_AUXH + 4 * a + i}")
                self._e("accw", (AUXL + 4 * a + i, IN_AUXL + 4 * a + i), f"v_accvgpr_write_b32 {rn(AUXL + 4 * a + i)}, v{IN_AUXL + 4 * a + i}")
        allow_rows(0)
        synced_tile = -1
        forced = 0

        def reads_for(i):
            """the two ds_reads of k-step i (lo first: the first MFMA of a k-step takes A_lo); a rendezvous first when i opens a tile"""
            nonlocal synced_tile
            ti, k = ks[i]
            did = False
            if k == 0 and ti > synced_tile:
                sync_for(ti)
                synced_tile, did = ti, True
            slot = T[ti].units[k] % R
            base = AR0 + 8 * (i % NA)
            return did, [(base, slot, 1), (base + 4, slot, 0)]

        for i in range(min(PF, N)):
            _, rd = reads_for(i)
            for d, s, pl in rd:
                self.dsread(d, s, pl)
        m = 0  # MFMA index
        for i in range(N):
            ti, k = ks[i]
            t = T[ti]
            bh, bl = t.bh[k], t.bl[k]
            alo, ahi = AR0 + 8 * (i % NA), AR0 + 8 * (i % NA) + 4
            if k == 0 and t.c0:
                forced += flush_acc(t.acc)
            forced += flush_producers(set(range(bh, bh + 4)) | set(range(bl, bl + 4)), m - 1)
            pend = []  # this k-step's gap work: the reads of k-step i + PF
            did_sync = False
            if i + PF < N:
                # a rendezvous (if k-step i + PF opens a tile) goes before this k-step's first MFMA
                did_sync, pend = reads_for(i + PF)
                if did_sync:
                    allow_rows(T[ti].p0)  # the barrier proves every wave is past the tiles before the current one
            ahead = 2 * min(PF - 1, N - 1 - i)  # reads of later k-steps already issued when this k-step starts
            for p, (areg, breg) in enumerate(((alo, bh), (ahi, bl), (ahi, bh))):
                operand_ready(breg)
                if p == 0:
                    self.waitl(ahead + 1)          # everything up to A_lo(i): A_hi(i) and the later k-steps may be in flight
                elif p == 1:
                    self.waitl(ahead + (1 if pend_issued else 0))
                self.mfma(t.acc, areg, breg, t.c0 and k == 0 and p == 0)
                m += 1
                if k == t.n - 1 and p == 2 and t.epi is not None:
                    queue_epilogue(ti, m - 1)
                # ---- gap
                pend_issued = False
                if p == 0 and pend:
                    self.dsread(*pend[0])
                    pend_issued = True
                elif p == 1 and pend:
                    self.dsread(*pend[1])
                elif p == 2 and pending_rows:
                    emit_half()
                n = 0
                while epi_q and n < self.FILL and epi_q[0][0] <= m - 1:
                    emit_epi(epi_q.pop(0))
                    n += 1
        assert not epi_q and not pending_rows and self.rows_issued == n_rows, (len(epi_q), pending_rows, self.rows_issued, n_rows)
        self._e("nop", (15,), "s_nop 15")
        self._e("nop", (3,), "s_nop 3")
        for g in range(16):
            self._e("accr", (OUT_HEAD + g, HEAD + g), f"v_accvgpr_read_b32 v{OUT_HEAD + g}, {rn(HEAD + g)}")
        self._e("restm0", (), "s_mov_b32 m0, %[m0save]")
        self.stats = dict(mfma=m, instructions=len(self.ins), forced_epilogue=forced, rows=n_rows,
                          barriers=sum(1 for x in self.ins if x.op == "barrier"), nops=sum(1 for x in self.ins if x.op == "nop"))

    def text(self):
        ab, drop = self.ablate, set()
        if "nodma" in ab:
            drop |= {"m0", "dma", "voff", "dma_pred", "dma_br", "label"}
        if "nobarrier" in ab:
            drop |= {"barrier"}
        if "noepi" in ab:
            drop |= {"sin", "fract", "pk", "shl", "and", "sub"}
        if "nostore" in ab:
            drop |= {"store", "soff", "pknorm"}
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
        lines = ["// GENERATED by csrc/gen/fwd_core3.py -- do not edit (tests/test_fwd_core.py checks it is current).",
                 f"// parity-mode (bf16x3) forward core, AUXS = {self.auxs}: {s['mfma']} MFMAs, {s['instructions']} instructions, "
                 f"{s['barriers']} rendezvous, {s['rows']} LDS-DMA rows, ring of {self.R} units, A fragments {self.PF} k-steps ahead.",
                 "// Operands: %[sh] / %[sl] hi / lo stream base (SGPR pairs), %[wb] LDS ring address + wave * 2048, %[wave], %[m0save]"
                 + (", %[ab] activation workspace (SGPR pair)." if self.save else ".")]
        lines += ['"' + t + '\\n"' for t in self.text()]
        return "\n".join(lines) + "\n"

    @staticmethod
    def clobber_file(save=0):
        # operands: v[0:127] (X hi, X lo in), v[128:143] (head out), v[144:159] (aux in), SIG, VL0, VL1, VOFF[, SOFF]
        regs = [f"v{r}" for r in range(TMP[0], TMP[1] + 4)]
        if save:
            regs += [f"v{r}" for r in range(SV[0], SV[1] + 8)]
        regs += [f"a{r}" for r in range(N_AGPR)]
        return ("// GENERATED by csrc/gen/fwd_core3.py: clobber list of the parity-mode forward core\n" + ", ".join(f'"{r}"' for r in regs)
                + ', "memory", "scc"\n')


# =================================================================================================================================
class Machine3:
    """Lane-accurate execution of one wave's instruction list (unified 512-register file), with the ring-protocol checks of
    fwd_core.Machine: no unit read before the rendezvous that covers its DMA row, no ring slot overwritten before the MFMAs that
    consume it were issued ahead of a barrier, vmcnt / lgkmcnt counts sufficient."""

    def __init__(self, core, hi_bits, lo_bits):
        self.c = core
        self.stream = (hi_bits, lo_bits)  # each [n_units, 64, 4] uint32
        self.v = np.zeros((512, 64), np.uint32)
        self.ring_unit = [[-1, -1] for _ in range(core.R)]
        self.ring = np.zeros((core.R, 2, 64, 4), np.uint32)
        self.synced_rows = 0
        self.uses = np.zeros(core.n_units, np.int32)  # MFMAs issued on the unit (3 = consumed)
        self.consumed_before_barrier = np.zeros(core.n_units, bool)
        self.pending = []
        self.ar = {}
        self.issued = {"full": [], "skip": []}
        self.last_write = {}
        self.soff = 0       # save = 16: byte offset of SOFF relative to the tile's workspace base
        self.stores = {}    # fragment -> [4, 64] uint32
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
                assert not any(d == areg for d, *_ in self.pending), "MFMA reads an A fragment still in flight"
                for r in list(range(breg, breg + 4)):
                    assert n - self.last_write.get(r, -10) >= 3, ("VALU write -> MFMA operand hazard", ins.text)
                unit, plane = self.ar[areg]
                self.uses[unit] += 1
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
                dst, slot, plane = a
                unit = self.ring_unit[slot][plane]
                assert unit >= 0 and unit // NW < self.synced_rows, ("unit read before the rendezvous that covers its row", unit, self.synced_rows)
                self.pending.append((dst, slot, plane, unit))
            elif op == "waitl":
                while len(self.pending) > a[0]:
                    dst, slot, plane, unit = self.pending.pop(0)
                    assert self.ring_unit[slot][plane] == unit
                    self.v[dst:dst + 4] = self.ring[slot, plane].T
                    self.ar[dst] = (unit, plane)
            elif op == "sync":
                vm, need = a
                for cls, reqs in self.issued.items():
                    landed = set(reqs[:max(len(reqs) - vm, 0)])
                    missing = [(r, p) for r in range(need) for p in (0, 1) if (r, p) in reqs and (r, p) not in landed]
                    assert not missing, ("vmcnt lets a needed request stay in flight", cls, vm, need, missing[:4])
                self.synced_rows = max(self.synced_rows, need)
            elif op == "barrier":
                self.consumed_before_barrier = self.uses == 3
            elif op == "dma":
                j, plane, partial = a
                self.issued["full"].append((j, plane))
                if partial is None:
                    self.issued["skip"].append((j, plane))
                for w in range(NW if partial is None else partial):
                    u = NW * j + w
                    slot = u % c.R
                    old = self.ring_unit[slot][plane]
                    assert old < 0 or self.consumed_before_barrier[old], ("DMA overwrites a unit not yet consumed by every wave", old, u)
                    self.ring_unit[slot][plane] = u
                    self.ring[slot, plane] = self.stream[plane][u]
            elif op == "fract":
                self._valu_reads(a[0])
                x = self.f(a[0]).astype(np.float64)
                self.setf(a[0], x - np.floor(x))
            elif op == "sin":
                self._valu_reads(a[0])
                self._trans[a[0]] = self._pc
                self.setf(a[0], np.sin(2 * np.pi * self.f(a[0]).astype(np.float64)))
            elif op == "pk":
                d, s0, s1 = a
                self._valu_reads(s0, s1)
                self.v[d] = (bf16_bits(self.f(s0)) | (bf16_bits(self.f(s1)) << 16)).astype(np.uint32)
                self.last_write[d] = n
            elif op == "shl":
                self.v[a[0]] = (self.v[a[1]] << np.uint32(16)).astype(np.uint32)
            elif op == "and":
                self.v[a[0]] = self.v[a[1]] & np.uint32(0xFFFF0000)
            elif op == "sub":
                d, x, y = a
                self._valu_reads(x, y)
                self.setf(d, self.f(x) - self.f(y))
            elif op == "pknorm":  # v_cvt_pknorm_u16_f32: round(clamp(x, 0, 1) * 65535), RNE
                d, s0, s1 = a
                self._valu_reads(s0, s1)
                u = [np.rint(np.clip(self.f(r).astype(np.float64), 0, 1) * 65535).astype(np.uint32) for r in (s0, s1)]
                self.v[d] = u[0] | (u[1] << np.uint32(16))
            elif op == "soff":
                self.soff += a[0]
            elif op == "store":
                reg, frag = a
                assert self.soff == frag * 1024 and frag not in self.stores, (self.soff, frag)
                self.stores[frag] = self.v[reg:reg + 4].copy()
            elif op in ("accw", "accr", "mov"):
                self._valu_reads(a[1])
                self.v[a[0]] = self.v[a[1]]
                self.last_write[a[0]] = n
        assert (self.uses == 3).all()



def main():
    import argparse

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ablate", default="")
    ap.add_argument("--PF", type=int, default=2)
    ap.add_argument("--FILL", type=int, default=2)
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = a.out or os.path.dirname(here)
    for auxs in (1, 2):
        for save in (0, 16):
            c = Core3(auxs, PF=a.PF, FILL=a.FILL, ablate=[x for x in a.ablate.split(",") if x], save=save)
            path = os.path.join(out_dir, f"mlp_fwd3_core_a{auxs}{'s16' if save else ''}.inc")
            with open(path, "w") as f:
                f.write(c.inc_file())
            print(path, c.stats)
    for save, name in ((0, "mlp_fwd3_core_clobbers.inc"), (16, "mlp_fwd3_core_clobbers_s16.inc")):
        with open(os.path.join(out_dir, name), "w") as f:
            f.write(Core3.clobber_file(save))


if __name__ == "__main__":
    sys.exit(main())
