#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of the fused Sat-NeRF forward MLP core (csrc/mlp_fwd2.inc).

Why a generator: profiles/r03_coissue.txt shows that v_mfma_f32_32x32x16_bf16 and the SIREN epilogue's VALU work (v_sin_f32,
v_cvt_pk_bf16_f32) overlap completely on one SIMD when the stream is placed by hand -- 32.3-32.8 cycles per MFMA with the A fragment
read from LDS -- while hipcc's schedule of the same work (csrc/mlp_fwd.inc) runs at ~75.  The stream is long (1,406 MFMAs per 32-point
tile, every register named) and regular, so it is emitted by this script: ``python fwd_core.py`` writes csrc/mlp_fwd_core_a{1,2}.inc
(string literals #included into one ``asm volatile`` statement).  tests/test_fwd_core.py re-generates the files (they must match what
is committed) and EXECUTES the instruction list on a lane-accurate numpy model of the register file / LDS ring / MFMA against the
packed weight stream, so register allocation, piece addressing, operand order and the DMA ring protocol are validated on the CPU.

Structure of the core (one wave = 32 points, 8 waves per workgroup, 2 per SIMD, <= 256 VGPRs; mlp_layout.h for the stream format):
  * MFMA i of the wave consumes piece i of the stream (1 KiB A fragment = one k-step of one 32-row output tile).  Pieces reach LDS by
    LDS-DMA "rows" of 8 pieces (one 1-KiB request per wave and row) into a flat ring of R pieces; ``sync`` points (counted vmcnt +
    s_barrier) every GROUP chunks make a group's pieces visible, after which the rows whose ring slots are free are requested.
  * A fragments are fetched PF MFMAs ahead (ds_read_b128 into a ring of PF + 1 register quads, counted lgkmcnt before each MFMA).
  * the epilogue of tile t-1 (16 v_sin in place on its accumulator, 8 v_cvt_pk into the next stage's B fragments) is spread over the
    gaps of tile t's MFMAs, starting two MFMAs after the tile's last one (XDL write -> VALU read needs 12 wait states).
  * fc_net.0 (K = 3, fp32 inputs) rides the matrix pipe too: the kernel prologue splits x, y, z and the fc_net.0 rows three ways into
    bf16 (h + m + l = 24 bits) and lays the 21 significant cross products over two k-steps (``l0_terms``), so pieces 0..15 of the ring
    are written by the prologue (not by LDS-DMA) and the stream starts with 8 two-MFMA tiles (always the bf16 opcode, MF0).
  * hazards the assembler does not pad inside inline asm are checked here: trans -> VALU use (1 state), VALU write -> MFMA operand
    (2 states), MFMA D -> VALU read (>= 2 MFMAs later), M0 write -> LDS-DMA (1 state).
"""
from __future__ import annotations

import os
import sys

import numpy as np

# ---- register map (VGPR numbers) ------------------------------------------------------------------------------------------------
X, Y = 0, 64            # the two 64-register activation vectors (16 B fragments of 4 registers)
ACC = (128, 144)        # double-buffered tile accumulators
AR0, NA = 160, 6        # ring of A-fragment register quads (PF + 1 entries)
AUX = 184               # aux B fragment(s): 184..187 (and 188..191 when AUXS = 2)
HEAD = 192              # 16 registers: the 5-row output head accumulator
SIG = 208               # sigma pre-activation
VL0, VL1, VOFF = 209, 210, 211  # LDS read bases (ring, ring + 64 KiB) + lane * 16; DMA source offset wave * 1024 + lane * 16 (advanced per row)
N_VGPR = 212
# SAVE8 (training forward, 8-bit workspaces of mlp_layout.h): two quads of store data, the PHASE8 magic constant, the workspace offset
# of this lane (tile base + lane * 16, advanced to the unit being stored), the feats scale bytes, MX8 temporaries
SV = (212, 216)
KMAGIC, SOFF, EB, MXT, K128 = 220, 221, 222, 224, 228   # EB: 2 registers, MXT: 4 (max, exponent, 1/scale, scaled value); K128 = 128.0f
N_VGPR_SAVE = 229
L0B = 232               # two quads: the B fragments of fc_net.0's two k-steps (split coordinates, l0_terms)
L0_PIECES = 16          # ring pieces 0..15: fc_net.0's A fragments, written by the kernel prologue
H0, H1 = 64, 96         # the two 32-register head hidden vectors (in Y once the trunk is done)

KS, HS = 16, 8          # k-steps of a 256-wide / 128-wide input
NW = 8                  # waves per workgroup = pieces per DMA row


def aux_steps(tau):
    return (8 + ((tau + 7) // 8) * 8 + 15) // 16


class Tile:
    """One chunk of the stream = one output tile: pieces [p0, p0 + n), the MFMA order, the B register of every MFMA, what happens to
    the accumulator afterwards."""

    def __init__(self, p0, bregs, n_aux, acc, c0, epi, out, name, save_unit=None, op="MF"):
        self.op = op  # MF = the mode's MFMA (bf16 / f16 operands), MF0 = always bf16 (fc_net.0's split operands)
        self.save_unit = save_unit  # SAVE8: workspace unit (1 KiB per tile of 32 points) of this tile's 16 values per lane, or None
        self.p0, self.n = p0, len(bregs)
        # MFMA order: aux k-steps first (their B operand is always ready: one more gap for the producing epilogue)
        order = list(range(self.n - n_aux, self.n)) + list(range(self.n - n_aux))
        self.pieces = [p0 + k for k in order]
        self.bregs = [bregs[k] for k in order]
        self.acc, self.c0, self.epi, self.out, self.name = acc, c0, epi, out, name


def stage_list(auxs):
    """The chunk list of FwdStream<AUXS> (mlp_layout.h) with the register assignment of this kernel."""
    tiles, p, tno = [], 0, 0
    auxb = [AUX + 4 * a for a in range(auxs)]

    def dense(inp, ks, ntiles, epi, outbase, name, unit0=None, first=False):
        nonlocal p, tno
        for t in range(ntiles):
            # fc_net.0 (first): two k-steps of split operands, no aux k-step; the small cross terms (k-step 1) go first
            b = [inp + 4 * k for k in range(ks)] + ([] if first else auxb)
            out = [outbase + 8 * t + q for q in range(8)] if outbase is not None else None
            tiles.append(Tile(p, b, 1 if first else auxs, ACC[tno & 1], True, epi, out, f"{name}.{t}", None if unit0 is None else auxs + unit0 + t,
                              "MF0" if first else "MF"))
            p += len(b)
            tno += 1

    head_started = [False]

    def head(inp, with_aux, name):
        nonlocal p
        b = [inp + 4 * k for k in range(HS)] + (auxb if with_aux else [])
        tiles.append(Tile(p, b, auxs if with_aux else 0, HEAD, not head_started[0], None, None, name))
        head_started[0] = True
        p += len(b)

    # workspace units (mlp_layout.h SR_FMT8, after the aux fragments): a_l at 8 l + t, feats 64 + t, rgbh 72, s1 76, e1 80, s2 84, s3 88
    dense(L0B, 2, 8, "sin", X, "L0", 0, first=True)
    assert p == L0_PIECES
    for l in range(7):
        dense(X if l % 2 == 0 else Y, KS, 8, "sin", Y if l % 2 == 0 else X, f"L{l + 1}", 8 * (l + 1))
    dense(Y, KS, 8, "id", X, "feats", 64)      # a7 (in Y) -> feats (in X)
    dense(Y, KS, 1, "sigma", None, "sigma")
    dense(X, KS, 4, "sin", H0, "rgbh", 72)
    head(H0, False, "Hr")
    dense(X, KS, 4, "sin", H1, "s1", 76)
    dense(H1, HS, 4, "sin", H0, "s2", 84)
    dense(H0, HS, 4, "sin", H1, "s3", 88)
    head(H1, False, "Hs")
    dense(X, KS, 4, "sin", H0, "e1", 80)
    head(H0, True, "Hb")
    return tiles, p


class Ins:
    __slots__ = ("op", "a", "text")

    def __init__(self, op, a, text):
        self.op, self.a, self.text = op, a, text


class Core:
    def __init__(self, auxs, R=128, PF=5, GROUP=2, FILL=None, ablate=(), save=0):
        assert R % NW == 0 and PF + 1 <= NA and save in (0, 8)
        if FILL is None:
            FILL = 3 if save else 2  # epilogue instructions per MFMA gap (a saving tile's epilogue has 42 instead of 24)
        self.save = save
        self.auxs, self.R, self.PF, self.GROUP, self.FILL = auxs, R, PF, GROUP, FILL
        self.ablate = set(ablate)  # timing experiments only (results are wrong): nodma, nobarrier, noepi, noread, nowaitl
        self.tiles, self.n_pieces = stage_list(auxs)
        self.ins = []
        self.stats = {}
        self._build()

    # ---- emission helpers ------------------------------------------------------------------------------------------------------
    def _e(self, op, a, text):
        self.ins.append(Ins(op, a, text))

    def mfma(self, acc, areg, breg, c0, op="MF"):
        c = "0" if c0 else f"v[{acc}:{acc + 15}]"
        self._e("mfma", (acc, areg, breg, c0), f"{op} v[{acc}:{acc + 15}], v[{areg}:{areg + 3}], v[{breg}:{breg + 3}], {c}")

    def dsread(self, dst, slot):
        base, off = (VL0, slot * 1024) if slot < 64 else (VL1, (slot - 64) * 1024)
        self._e("dsread", (dst, slot), f"ds_read_b128 v[{dst}:{dst + 3}], v{base} offset:{off}")

    def waitl(self, n):
        self._e("waitl", (n,), f"s_waitcnt lgkmcnt({n})")

    def sync(self, vm, need_rows):
        self._e("sync", (vm, need_rows), f"s_waitcnt vmcnt({vm})")
        self._e("barrier", (), "s_barrier")

    def dma_row(self, j, partial):
        # row j: wave w fetches piece 8 j + w into ring slot (8 j + w) % R from stream offset (8 j + w - 16) * 1024: VOFF holds
        # w * 1024 + lane * 16 + (j - 2) * 8192 (rows 0, 1 are fc_net.0's pieces, never requested)
        imm = ((NW * j) % self.R) * 1024
        if partial is not None:  # ragged last row: waves >= partial skip it (s_cbranch around the request; SCC from s_cmp)
            self._e("dma_pred", (j, partial), f"s_cmp_lt_u32 %[wave], {partial}")
            self._e("dma_br", (j,), f"s_cbranch_scc0 .Lskip_row{j}_%=")
        self._e("m0", (j,), f"s_add_u32 m0, %[wb], {imm}")
        self._e("nop", (0,), "s_nop 0")
        self._e("dma", (j, partial), f"global_load_lds_dwordx4 v{VOFF}, %[sb]")
        if partial is not None:
            self._e("label", (j,), f".Lskip_row{j}_%=:")
        self._e("voff", (), f"v_add_u32 v{VOFF}, 0x2000, v{VOFF}")

    # ---- the schedule ----------------------------------------------------------------------------------------------------------
    def _build(self):
        T, R, PF, G = self.tiles, self.R, self.PF, self.GROUP
        # global MFMA list
        mf = []  # (tile index, k in tile)
        for ti, t in enumerate(T):
            for k in range(t.n):
                mf.append((ti, k))
        N = len(mf)
        assert N == self.n_pieces
        first_of_tile = {}
        last_of_tile = {}
        for i, (ti, k) in enumerate(mf):
            first_of_tile.setdefault(ti, i)
            last_of_tile[ti] = i
        n_rows = (self.n_pieces + NW - 1) // NW
        partial_row = n_rows - 1 if self.n_pieces % NW else None
        partial_n = self.n_pieces % NW

        PRE = L0_PIECES // NW  # rows 0, 1 = fc_net.0's pieces: in the ring before the stream starts, never requested
        self.rows_issued = PRE
        pending_rows = []  # rows allowed but not yet emitted

        def allow_rows(free_below):
            # ring slots of pieces < free_below may be overwritten: row j writes pieces [8 j, 8 j + 8) over pieces 8 j - R ...
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
            need = (T[last].p0 + T[last].n + NW - 1) // NW  # rows 0 .. need-1 must have landed
            # every row allowed so far is emitted before the wait (program order = count order)
            while pending_rows:
                emit_row()
            issued = self.rows_issued
            assert issued >= need, (first_tile, issued, need)
            vm = issued - max(need, PRE)
            if partial_row is not None and issued > partial_row and need <= partial_row:
                vm -= 1  # waves that skipped the ragged row have one request fewer in flight
            vm = max(vm, 0)
            assert vm <= 63
            self.sync(vm, need)

        # epilogue queue: entries (earliest gap, kind, args)
        epi_q = []

        def queue_epilogue(ti):
            t = T[ti]
            g0 = last_of_tile[ti] + 2
            a = t.acc
            if self.save and t.save_unit is not None and t.epi == "sin":
                # PHASE8: byte = low mantissa byte of (pre-activation + 1.5 * 2^15), written straight into its place of the store
                # quad by an SDWA add BEFORE the value's sin overwrites the accumulator; cvt_pk one pair behind (trans -> VALU use)
                sv = SV[self.n_saves & 1]
                self.n_saves += 1
                if "phasefirst" in self.ablate and ti % self.GROUP == self.GROUP - 1:   # (its epilogue runs in a tile that holds no counted wait)
                    # experiment (correct results): all 16 PHASE8 bytes first, the store right behind them -- it is then ~8 MFMAs older when
                    # the next counted vmcnt wait (which cannot tell stores from LDS-DMA loads) comes -- then the sines and packs
                    for g in range(16):
                        epi_q.append([g0, "phase", (sv + (g >> 2), g & 3, a + g)])
                    epi_q.append([g0, "store", (sv, t.save_unit)])
                    for q in range(8):
                        for g in (2 * q, 2 * q + 1):
                            epi_q.append([g0, "sin", (a, g, t.out)])
                        if q > 0:
                            epi_q.append([g0, "pk", (a, q - 1, t.out)])
                    epi_q.append([g0, "pk", (a, 7, t.out)])
                else:
                    for q in range(8):
                        for g in (2 * q, 2 * q + 1):
                            epi_q.append([g0, "phase", (sv + (g >> 2), g & 3, a + g)])
                            epi_q.append([g0, "sin", (a, g, t.out)])
                        if q > 0:
                            epi_q.append([g0, "pk", (a, q - 1, t.out)])
                    epi_q.append([g0, "store", (sv, t.save_unit)])
                    epi_q.append([g0, "pk", (a, 7, t.out)])
            elif self.save and t.save_unit is not None and t.epi == "id":
                # MX8: E = exponent of 1.0079 max|v| clamped to [6, 254]; u = cvt_pk_u8(v * 2^(133 - E) + 128); byte t of EB = E
                sv = SV[self.n_saves & 1]
                self.n_saves += 1
                tt = t.save_unit - (self.auxs + 64)
                epi_q.append([g0, "mx_max", (a, 0, True)])
                for g in range(2, 16, 2):
                    epi_q.append([g0, "mx_max", (a, g, False)])
                epi_q.append([g0, "mx_exp", (tt,)])
                for g in range(16):
                    epi_q.append([g0, "mx_q", (sv + (g >> 2), g & 3, a + g)])
                epi_q.append([g0, "store", (sv, t.save_unit)])
                for q in range(8):
                    epi_q.append([g0, "pk", (a, q, t.out)])
                if tt == 7:
                    epi_q.append([g0, "store_scale", (self.auxs + 92,)])
            elif t.epi == "sin":
                seq = []
                # s0 s1 s2 c0 s3 s4 c1 ...: every cvt_pk is separated from its second sin by one instruction (trans -> VALU use)
                order = [("sin", 0), ("sin", 1)]
                for q in range(8):
                    if q < 7:
                        order.append(("sin", 2 * q + 2))
                    order.append(("pk", q))
                    if q < 7:
                        order.append(("sin", 2 * q + 3))
                for kind, v in order:
                    seq.append((kind, v))
                for kind, v in seq:
                    epi_q.append([g0, kind, (a, v, t.out)])
            elif t.epi == "id":
                for q in range(8):
                    epi_q.append([g0, "pk", (a, q, t.out)])
            elif t.epi == "sigma":
                epi_q.append([g0, "mov", (SIG, a)])

        self.n_saves, self.cur_unit = 0, 0
        written_at = {}   # VGPR -> index in self.ins of the VALU instruction that last wrote it
        trans_at = {}     # VGPR -> index of a v_sin that wrote it

        def emit_epi(item):
            _, kind, args = item
            if kind == "sin":
                a, g, _ = args
                self._e("sin", (a + g,), f"v_sin_f32 v{a + g}, v{a + g}")
                trans_at[a + g] = len(self.ins) - 1
                written_at[a + g] = len(self.ins) - 1
            elif kind == "pk":
                a, q, out = args
                for src in (a + 2 * q, a + 2 * q + 1):  # trans -> VALU use: one instruction in between
                    if src in trans_at and len(self.ins) - trans_at[src] < 2:
                        self._e("nop", (0,), "s_nop 0")
                self._e("pk", (out[q], a + 2 * q, a + 2 * q + 1), f"PK v{out[q]}, v{a + 2 * q}, v{a + 2 * q + 1}")
                written_at[out[q]] = len(self.ins) - 1
            elif kind == "mov":
                d, s = args
                self._e("mov", (d, s), f"v_mov_b32 v{d}, v{s}")
                written_at[d] = len(self.ins) - 1
            elif kind == "phase":
                d, byte, src = args
                self._e("phase", (d, byte, src), f"v_add_f32_sdwa v{d}, v{src}, v{KMAGIC} dst_sel:BYTE_{byte} dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
            elif kind == "store":
                sv, unit = args
                delta = (unit - self.cur_unit) * 1024
                self.cur_unit = unit
                self._e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self._e("store", (sv, unit), f"global_store_dwordx4 v{SOFF}, v[{sv}:{sv + 3}], %[ab] nt")
            elif kind == "store_scale":
                (unit,) = args
                delta = (unit - self.cur_unit) * 1024
                self.cur_unit = unit
                self._e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self._e("store2", (EB, unit), f"global_store_dwordx2 v{SOFF}, v[{EB}:{EB + 1}], %[ab]")
            elif kind == "mx_max":
                a, g, first = args
                m = MXT
                if first:
                    self._e("mx_max", (m, a + g, a + g + 1, None), f"v_max_f32 v{m}, |v{a + g}|, |v{a + g + 1}|")
                else:
                    self._e("mx_max", (m, a + g, a + g + 1, m), f"v_max3_f32 v{m}, |v{a + g}|, |v{a + g + 1}|, v{m}")
            elif kind == "mx_exp":
                (tt,) = args
                m, e, inv = MXT, MXT + 1, MXT + 2
                self._e("mx_e1", (m,), f"v_fmac_f32 v{m}, 0x3c000000, v{m}")                 # m = 2^-7 m + m (one rounding; VOP2 takes the literal)
                self._e("mx_e2", (e, m), f"v_lshrrev_b32 v{e}, 23, v{m}")
                self._e("mx_e3a", (e,), f"v_max_u32 v{e}, 6, v{e}")                          # clamp to [6, 254]
                self._e("mx_e3", (e,), f"v_min_u32 v{e}, 0xfe, v{e}")
                self._e("mx_e4", (inv, e), f"v_sub_u32 v{inv}, 0x104, v{e}")                 # 260 - E
                self._e("mx_e5", (inv,), f"v_lshlrev_b32 v{inv}, 23, v{inv}")                # 2^(133 - E)
                if tt & 3:
                    self._e("mx_e6", (EB + (tt >> 2), e, 8 * (tt & 3), False), f"v_lshl_or_b32 v{EB + (tt >> 2)}, v{e}, {8 * (tt & 3)}, v{EB + (tt >> 2)}")
                else:
                    self._e("mx_e6", (EB + (tt >> 2), e, 0, True), f"v_mov_b32 v{EB + (tt >> 2)}, v{e}")
            elif kind == "mx_q":
                d, byte, src = args
                tmp, inv = MXT + 3, MXT + 2
                self._e("mx_q1", (tmp, src, inv), f"v_fma_f32 v{tmp}, v{src}, v{inv}, v{K128}")   # v * 2^(133 - E) + 128
                self._e("mx_q2", (d, tmp, byte), f"v_cvt_pk_u8_f32 v{d}, v{tmp}, {byte}, v{d}")

        def flush_producers(regs, gap):
            """everything in the queue up to the last instruction that writes one of `regs` must be emitted now"""
            last = -1
            for qi, it in enumerate(epi_q):
                if it[1] == "pk" and it[2][2][it[2][1]] in regs:
                    last = qi
            forced = 0
            for _ in range(last + 1):
                it = epi_q.pop(0)
                assert it[0] <= gap + 1, ("epilogue needed before its accumulator is ready", it, gap)
                emit_epi(it)
                forced += 1
            return forced

        def flush_acc(acc):
            """a tile is about to start on accumulator `acc`: the epilogue that still reads it must be out first"""
            last = -1
            for qi, it in enumerate(epi_q):
                if ((it[1] in ("sin", "pk", "mx_max") and it[2][0] == acc) or (it[1] == "mov" and it[2][1] == acc)
                        or (it[1] in ("phase", "mx_q") and acc <= it[2][2] < acc + 16)):
                    last = qi
            for _ in range(last + 1):
                emit_epi(epi_q.pop(0))
            return last + 1

        # ---- preamble: request the first rows, make the first group visible, start the A-fragment pipeline
        self._e("savem0", (), "s_mov_b32 %[m0save], m0")
        allow_rows(0)
        sync_done_for = -1
        forced_total = 0

        def read_for(i):
            ti, k = mf[i]
            t = T[ti]
            nonlocal sync_done_for
            if ti > sync_done_for and ti % G == 0 and k == 0:
                sync_done_for = ti + G - 1
                last = min(ti + G, len(T)) - 1
                if T[last].p0 + T[last].n <= L0_PIECES:
                    return False  # fc_net.0's tiles: their pieces are in the ring already
                sync_for(ti)
                return True
            return False

        for i in range(min(PF, N)):
            read_for(i)
            ti, k = mf[i]
            self.dsread(AR0 + 4 * (i % NA), T[ti].pieces[k] % R)
        for i in range(N):
            ti, k = mf[i]
            t = T[ti]
            # B operand ready?  (VALU write -> MFMA read: 2 wait states)
            breg = t.bregs[k]
            if k == 0 and t.c0:
                forced_total += flush_acc(t.acc)
            forced_total += flush_producers(set(range(breg, breg + 4)), i - 1)
            recent = [written_at.get(r, -10) for r in range(breg, breg + 4)]
            dist = len(self.ins) - max(recent)
            if dist < 3:
                self._e("nop", (2 - dist + 1,), f"s_nop {2 - dist + 1}")
            self.waitl(min(PF - 1, N - 1 - i))
            self.mfma(t.acc, AR0 + 4 * (i % NA), breg, t.c0 and k == 0, t.op)
            if k == t.n - 1 and t.epi is not None:
                queue_epilogue(ti)
            # ---- gap(i)
            if i + PF < N:
                did_sync = read_for(i + PF)
                if did_sync:
                    # the barrier proves every wave has issued MFMA i: all tiles before the current one are consumed
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
        self._e("nop", (15,), "s_nop 15")  # the head accumulator is read by compiler code right after the statement
        self._e("restm0", (), "s_mov_b32 m0, %[m0save]")
        self.stats = dict(mfma=N, instructions=len(self.ins), forced_epilogue=forced_total, rows=n_rows,
                          barriers=sum(1 for x in self.ins if x.op == "barrier"), nops=sum(1 for x in self.ins if x.op == "nop"))

    # ---- text --------------------------------------------------------------------------------------------------------------------
    def text(self):
        ab = self.ablate
        drop = set()
        if "nodma" in ab:
            drop |= {"m0", "dma", "voff", "dma_pred", "dma_br", "label"}
        if "nobarrier" in ab:
            drop |= {"barrier"}
        if "nosync" in ab:
            drop |= {"barrier", "sync"}
        if "noepi" in ab:
            drop |= {"sin", "pk"}
        if "noread" in ab:
            drop |= {"dsread", "waitl"}
        if "nowaitl" in ab:
            drop |= {"waitl"}
        if "nonop" in ab:
            drop |= {"nop"}
        if "nostore" in ab:   # (saving cores) the workspace stores and their address steps
            drop |= {"store", "store2", "soff"}
        if "storesame" in ab:  # (saving cores) every store of a wave lands on the wave's FIRST unit: the store instructions and their vmcnt
            drop |= {"soff"}   # bookkeeping stay, the 196 MB of HBM traffic become 2 MB of L2 traffic
        if "noenc" in ab:     # (saving cores) the PHASE8 / MX8 encodes of the saved state
            drop |= {"phase", "mx_max", "mx_e1", "mx_e2", "mx_e3a", "mx_e3", "mx_e4", "mx_e5", "mx_e6", "mx_q1", "mx_q2"}
        if "empty" in ab:
            return ["s_mov_b32 %[m0save], m0", "v_mov_b32 v208, 0"] + [f"v_mov_b32 v{HEAD + g}, 0" for g in range(16)]
        out = []
        seen_first_sync = False
        for x in self.ins:
            if x.op == "sync":
                seen_first_sync = True
            if x.op in drop and (seen_first_sync or x.op not in ("m0", "dma", "voff")):
                continue
            out.append(x.text)
        return out

    def inc_file(self):
        lines = ["// GENERATED by csrc/gen/fwd_core.py -- do not edit (tests/test_fwd_core.py checks it is current).",
                 f"// fused forward core, AUXS = {self.auxs}: {self.stats['mfma']} MFMAs, {self.stats['instructions']} instructions, "
                 f"{self.stats['barriers']} rendezvous, {self.stats['rows']} LDS-DMA rows, ring of {self.R} pieces, A fragments {self.PF} ahead.",
                 f"// Ring pieces 0..{L0_PIECES - 1} (fc_net.0) are written by the kernel prologue; stream piece p is ring piece p + {L0_PIECES}.",
                 "// Operands: %[sb] stream base (SGPR pair), %[wb] LDS ring address + wave * 1024, %[wave] wave index, %[m0save] scratch SGPR"
                 + (", %[ab] activation workspace (SGPR pair)." if self.save else ".")]
        lines += ['"' + t + '\\n"' for t in self.text()]
        return "\n".join(lines) + "\n"

    @staticmethod
    def clobber_file(save=0):
        """registers the statement writes besides its operands (the aux and fc_net.0 fragments, v[192:211], KMAGIC, K128 and SOFF are operands)"""
        regs = list(range(X, AUX))
        if save:
            regs += [r for r in range(SV[0], N_VGPR_SAVE) if r not in (KMAGIC, SOFF, K128)]
        return ("// GENERATED by csrc/gen/fwd_core.py: clobber list of the forward core\n" + ", ".join(f'"v{r}"' for r in regs)
                + ', "memory", "scc"\n')


# =================================================================================================================================
# Lane-accurate interpreter (CPU validation)
# =================================================================================================================================
LANE = np.arange(64)
ROW_OF = (np.arange(16)[None, :] & 3) + 8 * (np.arange(16)[None, :] >> 2) + 4 * (LANE[:, None] >> 5)  # [lane, g]


def bf16_bits(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint32)


def bf16_to_f32(b):
    return (np.asarray(b, np.uint32) << 16).view(np.float32)


def frag_to_f32(regs4):
    """4 registers [4, 64] of packed bf16 pairs -> [64 lanes, 8] fp32"""
    lo = bf16_to_f32(regs4 & 0xFFFF)
    hi = bf16_to_f32(regs4 >> 16)
    out = np.empty((64, 8), np.float32)
    out[:, 0::2] = lo.T
    out[:, 1::2] = hi.T
    return out


def f32_to_frag(v):
    """[64, 8] fp32 -> 4 registers [4, 64] uint32 (bf16 RNE, element 0 in the low half)"""
    b = bf16_bits(v)
    return (b[:, 0::2] | (b[:, 1::2] << 16)).T.astype(np.uint32)


def split3(v):
    """fp32 -> three bf16-representable fp32 terms h + m + l = v to 24 bits (each RNE of the remainder; the remainders are exact)"""
    v = np.asarray(v, np.float32)
    h = bf16_to_f32(bf16_bits(v))
    r = (v - h).astype(np.float32)
    m = bf16_to_f32(bf16_bits(r))
    l = bf16_to_f32(bf16_bits((r - m).astype(np.float32)))
    return h, m, l


def l0_terms():
    """fc_net.0 as two bf16 k-steps: slot k of k-step s multiplies part A[s][k] of a table entry with part B[s][k] of a coordinate.
    Entries (column of the (w_x, w_y, w_z, b) table row, part of it, coordinate or None = 1.0, part of it); part 0 / 1 / 2 = h / m / l.
    k-step 0 carries w_h (x_h + x_m + x_l) + w_m x_h per coordinate and the bias, k-step 1 the remaining 2^-16 terms w_m x_m + w_l x_h."""
    Z = None
    s0 = []
    for c in range(3):
        s0 += [(c, 0, c, 0), (c, 0, c, 1), (c, 1, c, 0), (c, 0, c, 2)]
    s0 += [(3, 0, Z, 0), (3, 1, Z, 0), (3, 2, Z, 0), Z]
    s1 = []
    for c in range(3):
        s1 += [(c, 1, c, 1), (c, 2, c, 0)]
    s1 += [Z] * 10
    return s0, s1


def l0_b_frags(xyz):
    """B fragments of the two k-steps for 32 points: [2][64 lanes, 8] fp32 (lane = 32 h + point holds slots 8 h .. 8 h + 7)"""
    parts = [split3(xyz[:, c]) for c in range(3)]
    out = []
    for terms in l0_terms():
        f = np.zeros((64, 8), np.float32)
        for k, t in enumerate(terms):
            if t is None:
                continue
            _, _, c, xp = t
            f[32 * (k >> 3):32 * (k >> 3) + 32, k & 7] = 1.0 if c is None else parts[c][xp]
        out.append(f)
    return out


def l0_pieces(l0_table):
    """A fragments of fc_net.0: [16 pieces = (tile t, k-step s)][64 lanes = 32 (k >> 3) + row][8] fp32 from the (slot-ordered, scaled)
    table [256, 4]; row r of tile t is slot 16 (2 t + (g >> 3)) + 8 hh + (g & 7) with g = (r & 3) + 4 (r >> 3), hh = (r >> 2) & 1 -- the
    accumulator layout that makes tile t the B fragments 2 t, 2 t + 1 of the next layer (what the VALU prologue of r02 produced)."""
    r = np.arange(32)
    g, hh = (r & 3) + 4 * (r >> 3), (r >> 2) & 1
    out = np.zeros((16, 64, 8), np.float32)
    for t in range(8):
        slot = 16 * (2 * t + (g >> 3)) + 8 * hh + (g & 7)
        parts = [split3(l0_table[slot, c]) for c in range(4)]
        for s, terms in enumerate(l0_terms()):
            for k, tm in enumerate(terms):
                if tm is not None:
                    out[2 * t + s, 32 * (k >> 3):32 * (k >> 3) + 32, k & 7] = parts[tm[0]][tm[1]]
    return out


class Machine:
    """Executes the instruction list of one wave.  The ring is shared by the 8 waves of the workgroup: a DMA row copies the 8 pieces
    all waves fetch.  Protocol checks: a piece may only be read after a sync that covers its row, and a ring slot may only be
    overwritten once the MFMA that consumes its previous content has been issued (by this wave; the barrier in the sync extends that to
    all waves because a row is only requested after a barrier that follows that MFMA -- checked through `barrier_seen_after`)."""

    def __init__(self, core, stream_bits):
        self.c = core
        self.stream = stream_bits  # [n_pieces, 64 lanes, 4] uint32
        self.v = np.zeros((256, 64), np.uint32)
        self.ring_piece = [-1] * core.R
        self.ring = np.zeros((core.R, 64, 4), np.uint32)
        self.synced_rows = L0_PIECES // NW
        self.consumed = np.zeros(core.n_pieces, bool)   # piece's MFMA issued
        self.consumed_before_barrier = np.zeros(core.n_pieces, bool)
        self.pending_reads = []  # (dst, slot, piece)
        self.ar_piece = {}
        self.issued = {"full": [], "skip": []}  # DMA rows requested by a wave that takes / skips the ragged last row
        for p in range(L0_PIECES):  # fc_net.0's pieces are in the ring when the stream starts (the prologue wrote them, then a barrier)
            self.ring_piece[p], self.ring[p] = p, stream_bits[p]
        for rows in self.issued.values():
            rows += list(range(L0_PIECES // NW))
        self.voff_rows = 0   # rows VOFF has been advanced by: the source of the next request is stream piece 8 * voff_rows + wave
        self.soff = 0        # SAVE8: byte offset of SOFF relative to the tile's workspace base
        self.stores = {}     # unit -> [4, 64] uint32 (or [2, 64] for the scale unit)
        self._xdl, self._n_mfma, self._nops = {}, 0, 0
        self._trans, self._pc, self._m0_at = {}, 0, -9  # v_sin results, instruction index, last M0 write

    def f(self, r):
        return self.v[r].view(np.float32)

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
        for pc, ins in enumerate(c.ins):
            op, a = ins.op, ins.a
            self._pc = pc
            if op == "m0":
                self._m0_at = pc
            elif op == "dma":
                assert pc - self._m0_at >= 2, "M0 write -> LDS-DMA needs one wait state"
            if op == "mfma":
                acc, areg, breg, c0 = a
                assert not any(d == areg for d, _, _ in self.pending_reads), "MFMA reads an A fragment still in flight"
                piece = self.ar_piece[areg]
                assert not self.consumed[piece]
                self.consumed[piece] = True
                A = frag_to_f32(self.v[areg:areg + 4]).astype(np.float64)  # [lane, j] : row = lane & 31, k = 8 h + j
                B = frag_to_f32(self.v[breg:breg + 4]).astype(np.float64)
                Am = np.zeros((32, 16))
                Bm = np.zeros((16, 32))
                for h in range(2):
                    Am[:, 8 * h:8 * h + 8] = A[32 * h:32 * h + 32]
                    Bm[8 * h:8 * h + 8, :] = B[32 * h:32 * h + 32].T
                D = Am @ Bm
                d = D[ROW_OF, (LANE & 31)[:, None]]  # [lane, g]
                for g in range(16):
                    prev = np.zeros(64, np.float32) if c0 else self.f(acc + g).copy()
                    self.v[acc + g] = (prev.astype(np.float64) + d[:, g]).astype(np.float32).view(np.uint32)
                self._mfma_wrote(acc)
            elif op == "nop":
                self._nops += a[0] + 1
            elif op == "dsread":
                dst, slot = a
                piece = self.ring_piece[slot]
                assert piece >= 0, ("read of an empty ring slot", slot)
                assert piece // NW < self.synced_rows, ("piece read before the sync that covers its DMA row", piece, self.synced_rows)
                self.pending_reads.append((dst, slot, piece))
            elif op == "waitl":
                keep = a[0]
                while len(self.pending_reads) > keep:
                    dst, slot, piece = self.pending_reads.pop(0)
                    assert self.ring_piece[slot] == piece, "ring slot overwritten while a read of it was in flight"
                    self.v[dst:dst + 4] = self.ring[slot].T
                    self.ar_piece[dst] = piece
            elif op == "sync":
                vm, need = a
                for cls, rows in self.issued.items():  # loads return in order: all but the vm newest requests have landed
                    landed = set(rows[:max(len(rows) - vm, 0)])
                    missing = [r for r in range(need) if r in rows and r not in landed]
                    assert not missing, ("vmcnt lets a needed row stay in flight", cls, vm, need, missing)
                self.synced_rows = max(self.synced_rows, need)
            elif op == "barrier":
                self.consumed_before_barrier = self.consumed.copy()
            elif op == "dma":
                j, partial = a
                self.issued["full"].append(j)
                if partial is None:
                    self.issued["skip"].append(j)
                assert L0_PIECES + NW * self.voff_rows == NW * j, ("VOFF does not address row j of the stream", j, self.voff_rows)
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
                r = a[0]
                self._valu_reads(r)
                self._trans[r] = self._pc
                self.v[r] = np.sin(2 * np.pi * self.f(r).astype(np.float64)).astype(np.float32).view(np.uint32)
            elif op == "pk":
                d, s0, s1 = a
                self._valu_reads(s0, s1)
                self.v[d] = (bf16_bits(self.f(s0)) | (bf16_bits(self.f(s1)) << 16)).astype(np.uint32)
            elif op == "mov":
                d, s = a
                self._valu_reads(s)
                self.v[d] = self.v[s]
            elif op == "phase":  # v_add_f32_sdwa dst_sel:BYTE_k UNUSED_PRESERVE: the low byte of the fp32 sum
                d, byte, src = a
                self._valu_reads(src)
                ssum = (self.f(src) + self.f(KMAGIC)).astype(np.float32)
                b = ssum.view(np.uint32) & np.uint32(0xFF)
                self.v[d] = (self.v[d] & np.uint32(~(0xFF << (8 * byte)) & 0xFFFFFFFF)) | (b << np.uint32(8 * byte))
            elif op == "soff":
                self.soff += a[0]
            elif op in ("store", "store2"):
                reg, unit = a
                assert self.soff == unit * 1024 and unit not in self.stores, (self.soff, unit)
                self.stores[unit] = self.v[reg:reg + (4 if op == "store" else 2)].copy()
            elif op == "mx_max":
                m, x, y, z = a
                self._valu_reads(x, y)
                r = np.maximum(np.abs(self.f(x)), np.abs(self.f(y)))
                if z is not None:
                    r = np.maximum(r, self.f(z))
                self.v[m] = r.astype(np.float32).view(np.uint32)
            elif op == "mx_e1":
                m = a[0]
                self.v[m] = (self.f(m).astype(np.float64) * 0.0078125 + self.f(m).astype(np.float64)).astype(np.float32).view(np.uint32)
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
                t, src, inv = a
                self._valu_reads(src)
                self.v[t] = (self.f(src).astype(np.float64) * self.f(inv).astype(np.float64) + self.f(K128).astype(np.float64)).astype(np.float32).view(np.uint32)
            elif op == "mx_e3a":
                pass  # (the clamp is applied by mx_e3)
            elif op == "mx_q2":  # v_cvt_pk_u8_f32: round to nearest even, saturate to [0, 255]
                d, t, byte = a
                b = np.clip(np.rint(self.f(t).astype(np.float64)), 0, 255).astype(np.uint32)
                self.v[d] = (self.v[d] & np.uint32(~(0xFF << (8 * byte)) & 0xFFFFFFFF)) | (b << np.uint32(8 * byte))
        assert self.consumed.all()


def main():
    import argparse

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--out", default=None, help="output directory (default: csrc/)")
    ap.add_argument("--ablate", default="", help="comma list of timing ablations: nodma,nobarrier,nosync,noepi,noread,nowaitl,nonop,nostore,storesame,noenc")
    ap.add_argument("--R", type=int, default=128)
    ap.add_argument("--PF", type=int, default=5)
    ap.add_argument("--GROUP", type=int, default=2)
    ap.add_argument("--FILL", type=int, default=None)
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = a.out or os.path.dirname(here)
    os.makedirs(out_dir, exist_ok=True)
    for auxs in (1, 2):
        for save in (0, 8):
            c = Core(auxs, R=a.R, PF=a.PF, GROUP=a.GROUP, FILL=a.FILL, ablate=[x for x in a.ablate.split(",") if x], save=save)
            path = os.path.join(out_dir, f"mlp_fwd_core_a{auxs}{'s8' if save else ''}.inc")
            with open(path, "w") as f:
                f.write(c.inc_file())
            print(path, c.stats)
    for save, name in ((0, "mlp_fwd_core_clobbers.inc"), (8, "mlp_fwd_core_clobbers_s8.inc")):
        with open(os.path.join(out_dir, name), "w") as f:
            f.write(Core.clobber_file(save))


if __name__ == "__main__":
    sys.exit(main())
