#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of the dX kernel's TRUNK (csrc/mlp_bwd.inc, 8-bit workspaces, width 256).

The seven transposed trunk layers bL7 .. bL1 are two thirds of the data-gradient kernel (896 of its 1,308 MFMAs per 32-point tile):
    d a_{l-1} = W_l^T d pre_l  (8 output tiles x 16 k-steps),      d pre_{l-1} = d a_{l-1} * cos(2 pi phase_{l-1}),
d pre_{l-1} handed on in registers as bf16 B fragments and written to the dpre workspace as MX8 (codec8.h) for the weight-gradient kernel.
= autograd's backward through fc_net.2 .. fc_net.14 of SatNeRF.forward (models/satnerf.py:156-180) for every sample point.

Why now.  r03 tried this (tools/experiments/bwd_trunk_gen.py) and measured no gain: its epilogue -- cvt + scale + v_cos, multiply, pack,
fma + cvt_pk_u8 per value: ~120 VALU per 16-MFMA tile -- costs ~556 cycles of VALU issue per tile against 512 matrix cycles, so two
waves per SIMD were VALU-bound at ~70 cycles per MFMA however the stream was placed.  r04 shortened the codecs (one v_perm per phase
byte, v_max3 tree, one-rounding MX8 encode in v_pk_fma_f32 + byte gathers: 84 VALU, ~470 cycles per tile), which moved the VALU time UNDER
the matrix time -- but hipcc's schedule of that code (1,336 instructions per layer: 10.4 per MFMA, 26 SALU and 11 waits per tile, the
epilogue of tile t - 1 in one lump after tile t's chain) gained 3 %.  With the epilogue spread over the MFMA gaps the two pipes overlap,
as they do in the forward core (gen/fwd_core.py, whose recipe this is):
  * MFMA i consumes piece i of the trunk's part of the transposed stream from a flat LDS ring of R pieces fed by LDS-DMA rows of 8 pieces
    (one 1-KiB request per wave and row); a rendezvous (counted vmcnt + s_barrier) every GROUP tiles makes a group's pieces visible, after
    which the rows whose ring slots every wave has consumed are requested;
  * A fragments are read PF MFMAs ahead (ds_read_b128 into a ring of PF + 1 register quads, counted lgkmcnt before each MFMA);
  * the tile's saved phases (one 16-byte load per lane) are fetched two tiles ahead straight into registers; loads retire in order, so
    every vmcnt wait is an exact count of the LOADS (LDS-DMA rows and phase loads) issued after the one needed -- stores are left out of
    the count (they may complete out of order: a wait can only get longer);
  * the epilogue of tile t - 1 sits in the gaps of tile t's MFMAs, FILL instructions per gap, software-pipelined over value pairs so that
    no instruction waits for the transcendental unit: perm perm cos cos | perm perm | pk_mul | cos cos | cvt_pk ...
  * per layer: the eight MX8 exponent bytes of a lane go to the scale unit, and their maximum over the wave (SDWA byte maxima, six DPP
    steps, one readlane) to the wave's cell of the exponent-maxima table (mlp_layout.h; the weight-gradient kernel's fp16 range fit).
The arithmetic per value is exactly mlp_bwd.inc's (bpack / mx8_exponent / mx8_encode): the workspace bytes are bit-identical to the
compiler-scheduled kernel's, which tests/test_hip_backward.py asserts on the GPU; tests/test_bwd_core.py executes the stream on a
lane-accurate CPU model.

Hazards the assembler does not pad inside inline asm: trans -> VALU use (1 state: the pipelining above), VALU write -> MFMA operand (2),
MFMA D -> VALU read (the epilogue starts two MFMAs after the tile's last), M0 write -> LDS-DMA (1), VALU write -> DPP read (2).

``python bwd_core.py`` writes csrc/mlp_bwd_trunk_a{1,2}.inc and csrc/mlp_bwd_trunk_clobbers.inc.

Registers: v[0:63] X, v[64:127] Y (d pre vectors, ping-pong: X is the statement's in/out operand), v[128:159] two accumulators,
v[160:183] A ring, v[184:199] phase ring (4 tiles), v[200:203] temporaries, v[204:211] two store quads, v[212:213] scale bytes,
v214 max, v215 exponent, v[216:217] 2^(133-E) (low register used), v[218:219] the rounding constant, v220 = 0x43000000, v221 / v222 LDS
read bases, v223 stream offset, v224 phase offset, v225 dpre offset, v[226:229] temporaries, v230 LDS address of the wave's maxima cells, v[232:247] the tile's sixteen cosines / rounded MX8 values.
Scalar operands: %[sb] trunk part of the stream, %[wb] ring + wave * 1024, %[ab] activation workspace, %[db] dpre workspace, %[m0save].
"""
from __future__ import annotations

import os
import sys

X, Y = 0, 64
ACC = (128, 144)
AR0, NA = 160, 6
PH0, NPH = 184, 4
T0 = 200
SV = (204, 208)
EB = 212
M, E, INV, MAGIC, K43 = 214, 215, 216, 218, 220
VL0, VL1, VOFF, POFF, SOFF = 221, 222, 223, 224, 225
T1 = 226
VCELL = 230
TT = 232            # sixteen temporaries: the tile's cosines, then the rounded MX8 values (even-aligned pairs)
N_VGPR = 248
IN_REGS = (MAGIC, K43, VL0, VL1, VOFF, POFF, SOFF, VCELL)   # operands of the statement (wired by mlp_bwd.inc)
S_SEL = ("s84", "s85", "s86", "s87")     # v_perm selectors of phase byte k: the byte lands in bits 8..15 of 0x43000000 (codec8.h phase8_rev)
S_B4A, S_B4B, S_MAX = "s88", "s89", "s90"  # bytes4() selectors (codec8.h), the wave maximum
NW, KS, MT, LAYERS = 8, 16, 8, 7
D8_SCALE = 94      # kD8Scale (mlp_layout.h, width 256): unit of scale groups 0, 1
GROUPS_PER_UNIT = 2


class Ins:
    __slots__ = ("op", "a", "text")

    def __init__(self, op, a, text):
        self.op, self.a, self.text = op, a, text


class Trunk:
    def __init__(self, auxs, R=96, PF=5, GROUP=2, FILL=6, ablate=()):
        assert R % NW == 0 and PF + 1 <= NA
        self.auxs, self.R, self.PF, self.GROUP, self.FILL = auxs, R, PF, GROUP, FILL
        self.ablate = set(ablate)   # timing experiments (results wrong): noepi, nodma, nophase, nostore
        self.ins = []
        self.vm = []           # outstanding vector-memory LOADS in issue order (tags)
        self.p_unit = None     # unit POFF / SOFF currently point at
        self.s_unit = None
        self.last_trans = {}   # register -> index of the v_cos that wrote it
        self.last_valu = {}    # register -> index of the VALU instruction that wrote it
        self._build()

    def e(self, op, a, text):
        self.ins.append(Ins(op, a, text))

    # ---- vmcnt bookkeeping (loads only) --------------------------------------------------------------------------------------------
    def vm_issue(self, tag):
        self.vm.append(tag)
        assert len(self.vm) <= 63

    def vm_wait(self, tag):
        if tag not in self.vm:
            return None
        keep = len(self.vm) - 1 - self.vm.index(tag)
        self.vm = self.vm[len(self.vm) - keep:] if keep else []
        return keep

    def dma_row(self, j):
        imm = ((NW * j) % self.R) * 1024
        self.e("m0", (j,), f"s_add_u32 m0, %[wb], {imm}")
        self.e("nop", (0,), "s_nop 0")
        self.e("dma", (j,), f"global_load_lds_dwordx4 v{VOFF}, %[sb]")
        self.e("voff", (), f"v_add_u32 v{VOFF}, 0x2000, v{VOFF}")
        self.vm_issue(("row", j))

    def phase_unit(self, tau):
        l, t = 7 - tau // MT, tau % MT           # layer bL_l multiplies by cos(phase a_{l-1}): unit A + 8 (l - 1) + t
        return self.auxs + 8 * (l - 1) + t

    def phase_load(self, tau):
        unit = self.phase_unit(tau)
        delta = (unit - self.p_unit) * 1024
        self.p_unit = unit
        if delta:
            self.e("poff", (delta,), f"v_add_u32 v{POFF}, 0x{delta & 0xffffffff:x}, v{POFF}")
        r = PH0 + 4 * (tau % NPH)
        self.e("phload", (r, unit), f"global_load_dwordx4 v[{r}:{r + 3}], v{POFF}, %[ab] nt")
        self.vm_issue(("ph", tau))

    # ---- the epilogue of one tile: a list of closures, one instruction each ----------------------------------------------------------
    def epilogue_items(self, tau):
        """tile tau: accumulator ACC[tau & 1], phases PH[tau % 4]; d pre = acc * cos(2 pi u / 256) -> two bf16 B fragments of the next
        layer's input vector, MX8 bytes -> the dpre workspace (codec8.h: bit for bit what bpack / mx8_exponent / mx8_encode compute)"""
        l, t = 7 - tau // MT, tau % MT
        a, ph = ACC[tau & 1], PH0 + 4 * (tau % NPH)
        out = (Y if (7 - l) % 2 == 0 else X) + 8 * t
        sv = SV[tau & 1]
        it = []

        def V(op, args, text):
            it.append(lambda: self.e(op, args, text))

        def wait_phase():
            keep = self.vm_wait(("ph", tau))
            if keep is not None:
                self.e("waitv", (keep,), f"s_waitcnt vmcnt({keep})")
        it.append(wait_phase)
        # sixteen phase bytes -> sixteen temporaries (v_perm), sixteen cosines in place, eight packed multiplies into the accumulator, eight
        # packs: every consumer sits >= 8 instructions behind its producer -- the wave issues in order, so an instruction waiting for the
        # transcendental unit (or for a packed-fp32 result) also holds back the wave's next MFMA (measured: the pair-wise pipelined order
        # perm perm cos cos perm perm pk_mul ... ran the trunk at 58 cycles per MFMA slot against 40 without the epilogue)
        for g in range(16):
            V("perm_ph", (TT + g, ph + (g >> 2), g & 3), f"v_perm_b32 v{TT + g}, v{K43}, v{ph + (g >> 2)}, {S_SEL[g & 3]}")
        for g in range(16):
            V("cos", (TT + g,), f"v_cos_f32 v{TT + g}, v{TT + g}")
        for q in range(8):
            g = 2 * q
            V("pkmul", (a + g, TT + g), f"v_pk_mul_f32 v[{a + g}:{a + g + 1}], v[{a + g}:{a + g + 1}], v[{TT + g}:{TT + g + 1}]")
        for q in range(8):
            g = 2 * q
            V("pk", (out + q, a + g, a + g + 1), f"v_cvt_pk_bf16_f32 v{out + q}, v{a + g}, v{a + g + 1}")
        # maximum of the 16 magnitudes (exact whatever the order): 8 instructions
        m, t1, t2 = M, T1, T1 + 1
        ab = lambda g: f"|v{a + g}|"  # noqa: E731
        V("max3", (m, a + 0, a + 1, a + 2), f"v_max3_f32 v{m}, {ab(0)}, {ab(1)}, {ab(2)}")
        V("max3", (t1, a + 3, a + 4, a + 5), f"v_max3_f32 v{t1}, {ab(3)}, {ab(4)}, {ab(5)}")
        V("max3", (t2, a + 6, a + 7, a + 8), f"v_max3_f32 v{t2}, {ab(6)}, {ab(7)}, {ab(8)}")
        V("max3r", (m, m, t1, t2), f"v_max3_f32 v{m}, v{m}, v{t1}, v{t2}")
        V("max3", (t1, a + 9, a + 10, a + 11), f"v_max3_f32 v{t1}, {ab(9)}, {ab(10)}, {ab(11)}")
        V("max3", (t2, a + 12, a + 13, a + 14), f"v_max3_f32 v{t2}, {ab(12)}, {ab(13)}, {ab(14)}")
        V("max3m", (t1, t1, t2, a + 15), f"v_max3_f32 v{t1}, v{t1}, v{t2}, {ab(15)}")
        V("max2", (m, m, t1), f"v_max_f32 v{m}, v{m}, v{t1}")
        # E = exponent of 1.0079 max|v| clamped to [6, 254]; 2^(133 - E)
        V("mx_e1", (m,), f"v_fmac_f32 v{m}, 0x3c000000, v{m}")
        V("mx_e2", (E, m), f"v_lshrrev_b32 v{E}, 23, v{m}")
        V("mx_e3a", (E,), f"v_max_u32 v{E}, 6, v{E}")
        V("mx_e3", (E,), f"v_min_u32 v{E}, 0xfe, v{E}")
        V("mx_e4", (INV, E), f"v_sub_u32 v{INV}, 0x104, v{E}")
        V("mx_e5", (INV,), f"v_lshlrev_b32 v{INV}, 23, v{INV}")
        if t & 3:
            V("mx_e6", (EB + (t >> 2), E, 8 * (t & 3), False), f"v_lshl_or_b32 v{EB + (t >> 2)}, v{E}, {8 * (t & 3)}, v{EB + (t >> 2)}")
        else:
            V("mx_e6", (EB + (t >> 2), E, 0, True), f"v_mov_b32 v{EB + (t >> 2)}, v{E}")
        # u = low mantissa byte of v * 2^(133 - E) + (1.5 * 2^23 + 128): two values per v_pk_fma_f32, three v_perm per four bytes; again
        # producers first (eight fmas into the sixteen temporaries), then the byte gathers
        for q in range(8):
            g = 2 * q
            V("pkfma", (TT + g, a + g), f"v_pk_fma_f32 v[{TT + g}:{TT + g + 1}], v[{a + g}:{a + g + 1}], v[{INV}:{INV + 1}], v[{MAGIC}:{MAGIC + 1}] op_sel_hi:[1,0,0]")
        for q4 in range(4):
            g = 4 * q4
            V("b4a", (TT + g, TT + g + 1, TT + g), f"v_perm_b32 v{TT + g}, v{TT + g + 1}, v{TT + g}, {S_B4A}")
            V("b4a", (TT + g + 2, TT + g + 3, TT + g + 2), f"v_perm_b32 v{TT + g + 2}, v{TT + g + 3}, v{TT + g + 2}, {S_B4A}")
        for q4 in range(4):
            g = 4 * q4
            V("b4b", (sv + q4, TT + g + 2, TT + g), f"v_perm_b32 v{sv + q4}, v{TT + g + 2}, v{TT + g}, {S_B4B}")
        def store():
            unit = 8 * (l - 1) + t
            delta = (unit - self.s_unit) * 1024
            self.s_unit = unit
            if delta:
                self.e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
            self.e("store", (sv, unit), f"global_store_dwordx4 v{SOFF}, v[{sv}:{sv + 3}], %[db] nt")
        it.append(store)
        if t == MT - 1:
            g = l - 1   # the layer's eight scale bytes: group l - 1 -> unit kD8Scale + (l - 1) / 2, bytes 8 ((l - 1) % 2) ..

            def store_scale():
                unit = D8_SCALE + g // GROUPS_PER_UNIT
                delta = (unit - self.s_unit) * 1024
                self.s_unit = unit
                self.e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self.e("store2", (EB, unit, 8 * (g % GROUPS_PER_UNIT)), f"global_store_dwordx2 v{SOFF}, v[{EB}:{EB + 1}], %[db] offset:{8 * (g % GROUPS_PER_UNIT)}")
            it.append(store_scale)
            # largest of the lane's eight bytes, then of the wave (lane 63 after the row broadcasts), -> cell g of the wave's maxima
            c = T1
            for j, (reg, b0, b1) in enumerate(((EB, 0, 1), (EB, 2, 3), (EB + 1, 0, 1), (EB + 1, 2, 3))):
                V("bmax", (c + j, reg, b0, b1), f"v_max_u32_sdwa v{c + j}, v{reg}, v{reg} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_{b0} src1_sel:BYTE_{b1}")
            V("umax3", (c, c, c + 1, c + 2), f"v_max3_u32 v{c}, v{c}, v{c + 1}, v{c + 2}")
            V("umax", (c, c, c + 3), f"v_max_u32 v{c}, v{c}, v{c + 3}")
            def group(*parts):   # several instructions that must stay together (one epilogue item)
                it.append(lambda: [self.e(op, args, text) for op, args, text in parts])

            for ctrl in ("row_shr:1", "row_shr:2", "row_shr:4", "row_shr:8", "row_bcast:15 row_mask:0xa", "row_bcast:31 row_mask:0xc"):
                group(("nop", (1,), "s_nop 1"),   # VALU write -> DPP read of the same register: two wait states
                      ("dppmax", (c, ctrl), f"v_max_u32_dpp v{c}, v{c}, v{c} {ctrl}" + ("" if "row_mask" in ctrl else " row_mask:0xf") + " bank_mask:0xf"))
            group(("nop", (0,), "s_nop 0"),       # VALU write -> v_readlane of the same register: one wait state (hipcc pads it too)
                  ("readlane", (c,), f"v_readlane_b32 {S_MAX}, v{c}, 63"))
            V("smov", (c + 1,), f"v_mov_b32 v{c + 1}, {S_MAX}")
            # one lane writes the cell; nothing else may issue while EXEC is narrowed (an A-fragment read would load one lane)
            group(("exec1", (), "s_mov_b64 exec, 1"), ("cell", (c + 1, g), f"ds_write_b32 v{VCELL}, v{c + 1} offset:{4 * g}"), ("execall", (), "s_mov_b64 exec, -1"))
        return it

    def emit_item(self, f):
        """one epilogue instruction, with the wait states the assembler does not insert in inline asm"""
        n0 = len(self.ins)
        f()
        for k in range(n0, len(self.ins)):
            x = self.ins[k]
            if x.op == "cos":
                self.last_trans[x.a[0]] = k
        return len(self.ins) - n0

    def _build(self):
        R, PF, G = self.R, self.PF, self.GROUP
        NT = LAYERS * MT                      # 56 tiles = chunks of 16 pieces
        N = NT * KS
        n_rows = N // NW
        self.e("savem0", (), "s_mov_b32 %[m0save], m0")
        for k in range(4):
            self.e("sconst", (S_SEL[k],), f"s_mov_b32 {S_SEL[k]}, 0x{0x070c000c | (k << 8):08x}")
        self.e("sconst", (S_B4A,), f"s_mov_b32 {S_B4A}, 0x0c0c0400")
        self.e("sconst", (S_B4B,), f"s_mov_b32 {S_B4B}, 0x05040100")
        rows_issued, pending = 0, []

        def allow_rows(free_below):
            j = rows_issued + len(pending)
            while j < n_rows and NW * (j + 1) - R <= free_below:
                pending.append(j)
                j += 1

        def emit_row():
            nonlocal rows_issued
            self.dma_row(pending.pop(0))
            rows_issued += 1

        def sync_for(first_tile):
            last = min(first_tile + G, NT) - 1
            need = ((last + 1) * KS + NW - 1) // NW
            while pending:
                emit_row()
            assert rows_issued >= need
            keep = self.vm_wait(("row", need - 1))
            self.e("sync", (keep if keep is not None else len(self.vm), need), f"s_waitcnt vmcnt({keep if keep is not None else len(self.vm)})")
            self.e("barrier", (), "s_barrier")

        def dsread(i):
            slot = i % R
            base, off = (VL0, slot * 1024) if slot < 64 else (VL1, (slot - 64) * 1024)
            d = AR0 + 4 * (i % NA)
            self.e("dsread", (d, slot), f"ds_read_b128 v[{d}:{d + 3}], v{base} offset:{off}")

        # the workspace offsets start at the tile's base (unit 0)
        self.p_unit, self.s_unit = 0, 0
        allow_rows(0)
        sync_done_for = -1

        def read_for(i):
            nonlocal sync_done_for
            ti, k = divmod(i, KS)
            if ti > sync_done_for and ti % G == 0 and k == 0:
                sync_for(ti)
                sync_done_for = ti + G - 1
                return True
            return False

        self.phase_load(0)
        self.phase_load(1)
        for i in range(PF):
            read_for(i)
            dsread(i)
        epi = []          # [earliest gap, closure, tile]
        for i in range(N):
            ti, k = divmod(i, KS)
            l = 7 - ti // MT
            inp = X if (7 - l) % 2 == 0 else Y
            acc = ACC[ti & 1]
            if k == 0:
                # the tile before last's epilogue still reads this accumulator: it must be out (and every B fragment of a new layer
                # is produced by the previous layer's epilogues: tile 7's runs during this tile, k-steps 14, 15 come last)
                while epi and epi[0][2] <= ti - 2:
                    self.emit_item(epi.pop(0)[1])
            if ti % MT == 0 and k >= 14:
                while epi and epi[0][2] < ti:
                    self.emit_item(epi.pop(0)[1])
                if k == 14:
                    self.e("nop", (1,), "s_nop 1")  # VALU write -> MFMA operand: two wait states
            self.e("waitl", (min(PF - 1, N - 1 - i),), f"s_waitcnt lgkmcnt({min(PF - 1, N - 1 - i)})")
            c = "0" if k == 0 else f"v[{acc}:{acc + 15}]"
            areg = AR0 + 4 * (i % NA)
            self.e("mfma", (acc, areg, inp + 4 * k, k == 0), f"v_mfma_f32_32x32x16_bf16 v[{acc}:{acc + 15}], v[{areg}:{areg + 3}], v[{inp + 4 * k}:{inp + 4 * k + 3}], {c}")
            if k == KS - 1:
                for f in self.epilogue_items(ti):
                    epi.append([i + 2, f, ti])
            # ---- gap(i)
            if i + PF < N:
                if read_for(i + PF):
                    allow_rows(ti * KS)   # the barrier proves every wave has issued MFMA i: all tiles before the current one are consumed
                dsread(i + PF)
            if k == 0 and ti + 2 < NT:
                self.phase_load(ti + 2)
            if pending:
                emit_row()
            n = 0
            while epi and n < self.FILL and epi[0][0] <= i:
                n += max(self.emit_item(epi.pop(0)[1]), 1)
        if epi:  # the last tile's epilogue has no MFMAs to hide behind: XDL write -> VALU read of a 16-pass MFMA needs 18 wait states
            self.e("nop", (15,), "s_nop 15")
            self.e("nop", (3,), "s_nop 3")
        while epi:
            self.emit_item(epi.pop(0)[1])
        assert not pending and rows_issued == n_rows
        self.e("waitall", (), "s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.e("restm0", (), "s_mov_b32 m0, %[m0save]")
        self._check_hazards()
        kinds = {}
        for x in self.ins:
            kinds[x.op] = kinds.get(x.op, 0) + 1
        self.stats = kinds

    # ---- static checks of what the assembler would have padded in compiler-scheduled code -------------------------------------------
    def _check_hazards(self):
        VALU_W = {"perm_ph": 0, "cos": 0, "pkmul": None, "pk": 0, "max3": 0, "max3r": 0, "max3m": 0, "max2": 0, "mx_e1": 0, "mx_e2": 0, "mx_e3a": 0,
                  "mx_e3": 0, "mx_e4": 0, "mx_e5": 0, "mx_e6": 0, "pkfma": None, "b4a": 0, "b4b": 0, "bmax": 0, "umax3": 0, "umax": 0, "dppmax": 0,
                  "smov": 0}
        last_cos, last_w, last_mfma_d = {}, {}, {}
        for idx, x in enumerate(self.ins):
            reads, writes = set(), set()
            if x.op in VALU_W:
                if x.op in ("pkmul",):
                    a, tp = x.a
                    reads |= {a, a + 1, tp, tp + 1}
                    writes |= {a, a + 1}
                elif x.op == "pkfma":
                    d, a = x.a
                    reads |= {a, a + 1, INV, MAGIC}
                    writes |= {d, d + 1}
                elif x.op == "perm_ph":
                    writes.add(x.a[0]), reads.add(x.a[1])
                elif x.op == "cos":
                    writes.add(x.a[0]), reads.add(x.a[0])
                else:
                    regs = [r for r in x.a if isinstance(r, int)]
                    writes.add(regs[0])
                    reads |= set(regs[1:]) if x.op not in ("mx_e6", "bmax") else {x.a[1]}
                    if x.op in ("mx_e1", "mx_e3a", "mx_e3", "mx_e5", "dppmax", "umax3", "umax", "max3r", "max3m", "max2"):
                        reads.add(regs[0])
                # trans -> VALU use: at least one instruction in between
                for r in reads:
                    assert idx - last_cos.get(r, -10) >= 2, ("trans -> VALU use", idx, x.text)
                # MFMA D -> VALU read: the tile's last MFMA at least two MFMAs back
                for r in reads:
                    if r in last_mfma_d:
                        n_mf = sum(1 for y in self.ins[last_mfma_d[r] + 1:idx] if y.op == "mfma")
                        states = sum(y.a[0] + 1 for y in self.ins[last_mfma_d[r] + 1:idx] if y.op == "nop")
                        assert n_mf >= 2 or states >= 18, ("MFMA D -> VALU read", idx, x.text)
                if x.op == "dppmax":
                    assert idx - last_w.get(x.a[0], -10) >= 2 and self.ins[idx - 1].op == "nop", ("VALU write -> DPP read", idx)
                for r in writes:
                    last_w[r] = idx
                    last_mfma_d.pop(r, None)
                if x.op == "cos":
                    last_cos[x.a[0]] = idx
            elif x.op == "mfma":
                acc, areg, breg, _ = x.a
                for r in range(breg, breg + 4):
                    d = idx - last_w.get(r, -10)
                    nops = sum(y.a[0] + 1 for y in self.ins[max(idx - 3, 0):idx] if y.op == "nop")
                    assert d + nops >= 3 or d >= 3, ("VALU write -> MFMA operand", idx, r)
                for r in range(acc, acc + 16):
                    last_mfma_d[r] = idx
            elif x.op == "dma":
                assert self.ins[idx - 1].op == "nop" and self.ins[idx - 2].op == "m0", ("M0 write -> LDS-DMA", idx)
            elif x.op == "readlane":
                assert self.ins[idx - 1].op == "nop", ("VALU write -> v_readlane", idx)
            elif x.op == "cell":
                assert self.ins[idx - 1].op == "exec1" and self.ins[idx + 1].op == "execall", ("EXEC narrowed around anything but the cell write", idx)

    def text(self):
        ab = self.ablate
        drop = set()
        if "noepi" in ab:
            drop |= {"perm_ph", "cos", "pkmul", "max3", "max3r", "max3m", "max2", "mx_e1", "mx_e2", "mx_e3a", "mx_e3", "mx_e4", "mx_e5", "mx_e6", "pkfma",
                     "b4a", "b4b"}
        if "nodma" in ab:
            drop |= {"m0", "dma", "voff"}
        if "nophase" in ab:
            drop |= {"phload", "poff"}
        if "nostore" in ab:
            drop |= {"store", "store2", "soff"}
        return [x.text for x in self.ins if x.op not in drop]

    def inc_file(self):
        head = ["// GENERATED by csrc/gen/bwd_core.py -- do not edit (tests/test_bwd_core.py checks it is current).",
                f"// dX trunk, AUXS = {self.auxs}: {self.stats.get('mfma', 0)} MFMAs, {len(self.ins)} instructions ({len(self.ins) / max(self.stats.get('mfma', 1), 1):.2f} per MFMA), "
                f"{self.stats.get('barrier', 0)} rendezvous, ring of {self.R} pieces, A fragments {self.PF} ahead, {self.FILL} epilogue instructions per gap."]
        return "\n".join(head + ['"' + t + '\\n"' for t in self.text()]) + "\n"


def clobber_file():
    regs = [r for r in range(64, N_VGPR) if r not in IN_REGS]
    sregs = list(S_SEL) + [S_B4A, S_B4B, S_MAX]
    return ("// GENERATED by csrc/gen/bwd_core.py: clobber list of the dX trunk statement (X = v[0:63] is in/out, v218 v220..v225 v230 are inputs)\n"
            + ", ".join(f'"v{r}"' for r in regs) + ", " + ", ".join(f'"{s}"' for s in sregs) + ', "memory", "scc"\n')


def main():
    """bwd_core.py [out_dir [suffix [ablation,...]]]"""
    out_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1:
        out_dir = sys.argv[1]
    suffix = sys.argv[2] if len(sys.argv) > 2 else ""
    ablate = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ()
    for auxs in (1, 2):
        t = Trunk(auxs, ablate=ablate)
        with open(os.path.join(out_dir, f"mlp_bwd_trunk_a{auxs}{suffix}.inc"), "w") as f:
            f.write(t.inc_file())
        print(auxs, len(t.ins), "instructions,", t.stats.get("mfma"), "MFMAs;", {k: v for k, v in sorted(t.stats.items())})
    with open(os.path.join(out_dir, "mlp_bwd_trunk_clobbers.inc"), "w") as f:
        f.write(clobber_file())


if __name__ == "__main__":
    sys.exit(main())
