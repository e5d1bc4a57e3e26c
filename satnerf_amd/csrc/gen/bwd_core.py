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

A0 = 256                     # unified register numbering: 0..255 = v, 256..511 = a (the 512-wide kernel: one wave per SIMD, both files)


def rn(r, n=1):
    f, i = ("v", r) if r < A0 else ("a", r - A0)
    return f"{f}{i}" if n == 1 else f"{f}[{i}:{i + n - 1}]"


class Geo:
    """register map and sizes of one trunk width.  256: 8 waves x <= 256 VGPRs, both d pre vectors in VGPRs.  512: 4 waves x (256 VGPRs + 256
    AGPRs): a 32-point vector is 128 registers -- X in v[0:127], Y and the A-fragment ring in AGPRs (MFMA reads A / B from either file,
    ds_read_b128 writes either); what an epilogue produces for Y goes through v_accvgpr_write (as in gen/fwd_core512.py)."""

    def __init__(self, feat):
        assert feat in (256, 512)
        self.feat = feat
        self.KS, self.MT, self.NW = feat // 16, feat // 32, (8 if feat == 256 else 4)
        # g1: the stream starts one stage earlier, at bG1 ([d feats | d sigma_pre] -> d a_7: MT tiles of KS + 1 k-steps, the last reading the
        # d sigma_pre fragment XS).  Built, tested (tests/test_bwd_core.py runs either setting) and OFF: at width 256 the compiler cannot keep
        # the hand-over (68 pinned registers at two waves per SIMD) out of scratch -- 103 spilled registers in the stages before it, the
        # kernel 6 % SLOWER; at width 512 (no spills) 440 us against 426 with the trunk alone (r05, interleaved on one box)
        self.g1 = False
        self.LAYERS = 8 if self.g1 else 7
        self.R, self.GROUP, self.FILL = (96, 2, 6) if feat == 256 else (128, 1, 4)
        self.NEB = self.MT // 4                      # registers of a layer's exponent bytes
        if feat == 256:
            self.X, self.Y = 0, 64
            self.ACC = (128, 144)
            self.AR0 = 160
            self.PH0 = 184
            self.T0 = 200                            # (unused since the producers-first epilogue; kept out of the clobbers' way)
            self.SV = (204, 208)
            self.EB = 212
            self.M, self.E, self.INV, self.MAGIC, self.K43 = 214, 215, 216, 218, 220
            self.VL0, self.VL1, self.VOFF, self.POFF, self.SOFF = 221, 222, 223, 224, 225
            self.T1 = 226
            self.VCELL = 230
            self.TT = 232
            self.N_VGPR, self.N_AGPR = 248, 0
            self.D8_SCALE, self.GROUPS_PER_UNIT = 94, 2
        else:
            self.X, self.Y = 0, A0
            self.ACC = (128, 144)
            self.AR0 = A0 + 128
            self.PH0 = 160
            self.TT = 176
            self.T1 = 192
            self.SV = (196, 200)
            self.EB = 204
            self.M, self.E, self.INV, self.MAGIC, self.K43 = 208, 209, 210, 212, 214
            self.VL0, self.VL1, self.VOFF, self.POFF, self.SOFF = 215, 216, 217, 218, 219
            self.VCELL = 220
            self.N_VGPR, self.N_AGPR = 222, 152
            self.D8_SCALE, self.GROUPS_PER_UNIT = 186, 1
        self.NA, self.NPH = 6, 4
        self.IN_REGS = (self.MAGIC, self.K43, self.VL0, self.VL1, self.VOFF, self.POFF, self.SOFF, self.VCELL)   # operands (wired by mlp_bwd.inc)
        self.XREGS = self.KS * 4                     # registers of a d pre vector
        # g1: the d sigma_pre B fragment = an in/out operand in the LAST quad of Y, which bG1's own last tile's epilogue overwrites only
        # after the layer's last MFMA has read it
        self.XS = self.Y + self.XREGS - 4
        self.L0 = 8 if self.g1 else 7                # `l` of the stream's first layer (8 = bG1)


S_SEL = ("s84", "s85", "s86", "s87")     # v_perm selectors of phase byte k: the byte lands in bits 8..15 of 0x43000000 (codec8.h phase8_rev)
S_B4A, S_B4B, S_MAX = "s88", "s89", "s90"  # bytes4() selectors (codec8.h), the wave maximum


class Ins:
    __slots__ = ("op", "a", "text")

    def __init__(self, op, a, text):
        self.op, self.a, self.text = op, a, text


class Trunk:
    def __init__(self, auxs, feat=256, PF=5, ablate=()):
        self.g = g = Geo(feat)
        self.auxs, self.R, self.PF, self.GROUP, self.FILL = auxs, g.R, PF, g.GROUP, g.FILL
        assert self.R % g.NW == 0 and PF + 1 <= g.NA
        self.ablate = set(ablate)   # timing experiments (results wrong): noepi, nodma, nophase, nostore
        self.ins = []
        self.vm = []           # outstanding vector-memory LOADS in issue order (tags)
        self.p_unit = None     # unit POFF / SOFF currently point at
        self.s_unit = None
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
        g = self.g
        imm = ((g.NW * j) % self.R) * 1024
        self.e("m0", (j,), f"s_add_u32 m0, %[wb], {imm}")
        self.e("nop", (0,), "s_nop 0")
        self.e("dma", (j,), f"global_load_lds_dwordx4 v{g.VOFF}, %[sb]")
        self.e("voff", (), f"v_add_u32 v{g.VOFF}, 0x{g.NW * 1024:x}, v{g.VOFF}")
        self.vm_issue(("row", j))

    def phase_unit(self, tau):
        g = self.g
        l, t = g.L0 - tau // g.MT, tau % g.MT    # layer l (8 = bG1, 7..1 = bL_l) multiplies by cos(phase a_{l-1}): unit A + MT (l - 1) + t
        return self.auxs + g.MT * (l - 1) + t

    def phase_load(self, tau):
        g = self.g
        unit = self.phase_unit(tau)
        delta = (unit - self.p_unit) * 1024
        self.p_unit = unit
        if delta:
            self.e("poff", (delta,), f"v_add_u32 v{g.POFF}, 0x{delta & 0xffffffff:x}, v{g.POFF}")
        r = g.PH0 + 4 * (tau % g.NPH)
        self.e("phload", (r, unit), f"global_load_dwordx4 v[{r}:{r + 3}], v{g.POFF}, %[ab] nt")
        self.vm_issue(("ph", tau))

    # ---- the epilogue of one tile: a list of closures, one instruction each ----------------------------------------------------------
    def epilogue_items(self, tau):
        """tile tau: accumulator ACC[tau & 1], phases PH[tau % 4]; d pre = acc * cos(2 pi u / 256) -> two bf16 B fragments of the next
        layer's input vector, MX8 bytes -> the dpre workspace (codec8.h: bit for bit what bpack / mx8_exponent / mx8_encode compute)"""
        g = self.g
        MT, TT, K43, EB, E, M, INV, MAGIC, T1, SOFF = g.MT, g.TT, g.K43, g.EB, g.E, g.M, g.INV, g.MAGIC, g.T1, g.SOFF
        l, t = g.L0 - tau // MT, tau % MT
        a, ph = g.ACC[tau & 1], g.PH0 + 4 * (tau % g.NPH)
        out = (g.Y if (g.L0 - l) % 2 == 0 else g.X) + 8 * t
        sv = g.SV[tau & 1]
        it = []

        def V(op, args, text):
            it.append(lambda: self.e(op, args, text))

        def group(*parts):   # several instructions that must stay together (one epilogue item)
            it.append(lambda: [self.e(op, args, text) for op, args, text in parts])

        def wait_phase():
            keep = self.vm_wait(("ph", tau))
            if keep is not None:
                self.e("waitv", (keep,), f"s_waitcnt vmcnt({keep})")
        it.append(wait_phase)
        # sixteen phase bytes -> sixteen temporaries (v_perm), sixteen cosines in place, eight packed multiplies into the accumulator, eight
        # packs: every consumer sits >= 8 instructions behind its producer -- the wave issues in order, so an instruction waiting for the
        # transcendental unit (or for a packed-fp32 result) also holds back the wave's next MFMA (measured: the pair-wise pipelined order
        # perm perm cos cos perm perm pk_mul ... ran the trunk at 58 cycles per MFMA slot against 40 without the epilogue)
        for gg in range(16):
            V("perm_ph", (TT + gg, ph + (gg >> 2), gg & 3), f"v_perm_b32 v{TT + gg}, v{K43}, v{ph + (gg >> 2)}, {S_SEL[gg & 3]}")
        for gg in range(16):
            V("cos", (TT + gg,), f"v_cos_f32 v{TT + gg}, v{TT + gg}")
        for q in range(8):
            gg = 2 * q
            V("pkmul", (a + gg, TT + gg), f"v_pk_mul_f32 v[{a + gg}:{a + gg + 1}], v[{a + gg}:{a + gg + 1}], v[{TT + gg}:{TT + gg + 1}]")
        if out < A0:
            for q in range(8):
                gg = 2 * q
                V("pk", (out + q, a + gg, a + gg + 1), f"v_cvt_pk_bf16_f32 v{out + q}, v{a + gg}, v{a + gg + 1}")
        else:   # the output vector lives in AGPRs: pack into the (now dead) cosine temporaries, then eight v_accvgpr_write
            for q in range(8):
                gg = 2 * q
                V("pk", (TT + q, a + gg, a + gg + 1), f"v_cvt_pk_bf16_f32 v{TT + q}, v{a + gg}, v{a + gg + 1}")
            for q in range(8):
                V("accw", (out + q, TT + q), f"v_accvgpr_write_b32 {rn(out + q)}, v{TT + q}")
        # maximum of the 16 magnitudes (exact whatever the order): 8 instructions
        m, t1, t2 = M, T1, T1 + 1
        ab = lambda k: f"|v{a + k}|"  # noqa: E731
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
            gg = 2 * q
            V("pkfma", (TT + gg, a + gg), f"v_pk_fma_f32 v[{TT + gg}:{TT + gg + 1}], v[{a + gg}:{a + gg + 1}], v[{INV}:{INV + 1}], v[{MAGIC}:{MAGIC + 1}] op_sel_hi:[1,0,0]")
        for q4 in range(4):
            gg = 4 * q4
            V("b4a", (TT + gg, TT + gg + 1, TT + gg), f"v_perm_b32 v{TT + gg}, v{TT + gg + 1}, v{TT + gg}, {S_B4A}")
            V("b4a", (TT + gg + 2, TT + gg + 3, TT + gg + 2), f"v_perm_b32 v{TT + gg + 2}, v{TT + gg + 3}, v{TT + gg + 2}, {S_B4A}")
        for q4 in range(4):
            gg = 4 * q4
            V("b4b", (sv + q4, TT + gg + 2, TT + gg), f"v_perm_b32 v{sv + q4}, v{TT + gg + 2}, v{TT + gg}, {S_B4B}")

        def store():
            unit = MT * (l - 1) + t
            delta = (unit - self.s_unit) * 1024
            self.s_unit = unit
            if delta:
                self.e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
            self.e("store", (sv, unit), f"global_store_dwordx4 v{SOFF}, v[{sv}:{sv + 3}], %[db] nt")
        store.is_store = True
        it.append(store)
        if t == MT - 1:
            grp = l - 1   # the layer's MT scale bytes: group l - 1 -> unit kD8Scale + (l - 1) / groups per unit, its slot of the lane's 16 bytes

            def store_scale():
                unit = g.D8_SCALE + grp // g.GROUPS_PER_UNIT
                delta = (unit - self.s_unit) * 1024
                self.s_unit = unit
                off = MT * (grp % g.GROUPS_PER_UNIT)
                self.e("soff", (delta,), f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self.e("store2", (EB, unit, off), f"global_store_dwordx{g.NEB} v{SOFF}, v[{EB}:{EB + g.NEB - 1}], %[db]" + (f" offset:{off}" if g.GROUPS_PER_UNIT > 1 else ""))
            it.append(store_scale)
            # largest of the lane's MT bytes, then of the wave (lane 63 after the row broadcasts), -> cell `grp` of the wave's maxima
            c = T1
            n = 0
            for r in range(g.NEB):
                for b0 in (0, 2):
                    dst = c + (n & 3) if n < 4 else TT + (n - 4)          # (the 512-wide layer has eight pair maxima)
                    V("bmax", (dst, EB + r, b0, b0 + 1), f"v_max_u32_sdwa v{dst}, v{EB + r}, v{EB + r} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_{b0} src1_sel:BYTE_{b0 + 1}")
                    n += 1
            V("umax3", (c, c, c + 1, c + 2), f"v_max3_u32 v{c}, v{c}, v{c + 1}, v{c + 2}")
            V("umax", (c, c, c + 3), f"v_max_u32 v{c}, v{c}, v{c + 3}")
            if g.NEB > 2:
                V("umax3", (c + 1, TT, TT + 1, TT + 2), f"v_max3_u32 v{c + 1}, v{TT}, v{TT + 1}, v{TT + 2}")
                V("umax3", (c, c, c + 1, TT + 3), f"v_max3_u32 v{c}, v{c}, v{c + 1}, v{TT + 3}")
            for ctrl in ("row_shr:1", "row_shr:2", "row_shr:4", "row_shr:8", "row_bcast:15 row_mask:0xa", "row_bcast:31 row_mask:0xc"):
                group(("nop", (1,), "s_nop 1"),   # VALU write -> DPP read of the same register: two wait states
                      ("dppmax", (c, ctrl), f"v_max_u32_dpp v{c}, v{c}, v{c} {ctrl}" + ("" if "row_mask" in ctrl else " row_mask:0xf") + " bank_mask:0xf"))
            group(("nop", (0,), "s_nop 0"),       # VALU write -> v_readlane of the same register: one wait state (hipcc pads it too)
                  ("readlane", (c,), f"v_readlane_b32 {S_MAX}, v{c}, 63"))
            V("smov", (c + 1,), f"v_mov_b32 v{c + 1}, {S_MAX}")
            # one lane writes the cell; nothing else may issue while EXEC is narrowed (an A-fragment read would load one lane)
            group(("exec1", (), "s_mov_b64 exec, 1"), ("cell", (c + 1, grp), f"ds_write_b32 v{g.VCELL}, v{c + 1} offset:{4 * grp}"), ("execall", (), "s_mov_b64 exec, -1"))
        return it

    def emit_item(self, f):
        if "storelate" in self.ablate and getattr(f, "is_store", False) and not self.last_sync_done:
            # experiment (correct results): a tile's dpre store, due now, is held back until the gap behind the NEXT counted vmcnt wait (which
            # cannot tell stores from loads): it is then most of a tile old, not two MFMAs, when the following wait comes.  Its quad (SV, one of
            # two) is not written again before the epilogue after next.
            self.late.append([self.syncs_done + 1, f, self.cur_tile])
            return 0
        n0 = len(self.ins)
        f()
        return len(self.ins) - n0

    def _build(self):
        g = self.g
        R, PF, G, KS, MT, NW = self.R, self.PF, self.GROUP, g.KS, g.MT, g.NW
        NT = g.LAYERS * MT                    # tiles = chunks of KS pieces (bG1's: KS + 1)
        ks_of = lambda ti: KS + 1 if (g.g1 and ti < MT) else KS   # noqa: E731
        mf = [(ti, k) for ti in range(NT) for k in range(ks_of(ti))]
        first_piece = {}
        for i, (ti, k) in enumerate(mf):
            first_piece.setdefault(ti, i)
        N = len(mf)
        assert N % NW == 0
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
            need = (first_piece[last] + ks_of(last) + NW - 1) // NW
            while pending:
                emit_row()
            assert rows_issued >= need
            keep = self.vm_wait(("row", need - 1))
            self.e("sync", (keep if keep is not None else len(self.vm), need), f"s_waitcnt vmcnt({keep if keep is not None else len(self.vm)})")
            self.e("barrier", (), "s_barrier")

        def dsread(i):
            slot = i % R
            base, off = (g.VL0, slot * 1024) if slot < 64 else (g.VL1, (slot - 64) * 1024)
            d = g.AR0 + 4 * (i % g.NA)
            self.e("dsread", (d, slot), f"ds_read_b128 {rn(d, 4)}, v{base} offset:{off}")

        # the workspace offsets start at the tile's base (unit 0)
        self.p_unit, self.s_unit = 0, 0
        allow_rows(0)
        sync_done_for = -1

        def read_for(i):
            nonlocal sync_done_for
            ti, k = mf[i]
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
        self.late, self.syncs_done, self.cur_tile, self.last_sync_done = [], 0, 0, False   # (storelate) deferred stores: [rendezvous count after which they go out, closure]
        for i in range(N):
            ti, k = mf[i]
            l = g.L0 - ti // MT
            inp = g.X if (g.L0 - l) % 2 == 0 else g.Y
            acc = g.ACC[ti & 1]
            self.cur_tile = ti
            if k == 0:
                while self.late and self.late[0][2] <= ti - 2:   # (storelate) no rendezvous came (the stream's end): out before its quad is rewritten
                    self.late.pop(0)[1]()
                # the tile before last's epilogue still reads this accumulator: it must be out (and every B fragment of a new layer
                # is produced by the previous layer's epilogues: the last tile's runs during this tile, its two k-steps come last)
                while epi and epi[0][2] <= ti - 2:
                    self.emit_item(epi.pop(0)[1])
            if ti % MT == 0 and ti > 0 and KS - 2 <= k < KS:
                while epi and epi[0][2] < ti:
                    self.emit_item(epi.pop(0)[1])
                if k == KS - 2:
                    self.e("nop", (1,), "s_nop 1")  # VALU write -> MFMA operand: two wait states
            self.e("waitl", (min(PF - 1, N - 1 - i),), f"s_waitcnt lgkmcnt({min(PF - 1, N - 1 - i)})")
            c = "0" if k == 0 else f"v[{acc}:{acc + 15}]"
            areg = g.AR0 + 4 * (i % g.NA)
            breg = inp + 4 * k if k < KS else g.XS
            self.e("mfma", (acc, areg, breg, k == 0), f"v_mfma_f32_32x32x16_bf16 v[{acc}:{acc + 15}], {rn(areg, 4)}, {rn(breg, 4)}, {c}")
            if k == ks_of(ti) - 1:
                for f in self.epilogue_items(ti):
                    epi.append([i + 2, f, ti])
            # ---- gap(i)
            if i + PF < N:
                if read_for(i + PF):
                    self.syncs_done += 1
                    self.last_sync_done = sync_done_for >= NT - 1   # (storelate) no further rendezvous: stores go out when due again
                    allow_rows(first_piece[ti])   # the barrier proves every wave has issued MFMA i: all tiles before the current one are consumed
                dsread(i + PF)
            while self.late and self.late[0][0] <= self.syncs_done:
                self.late.pop(0)[1]()
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
        while self.late:   # (storelate) the last tiles' stores
            self.late.pop(0)[1]()
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
        g = self.g
        VALU_W = {"perm_ph", "cos", "pkmul", "pk", "accw", "max3", "max3r", "max3m", "max2", "mx_e1", "mx_e2", "mx_e3a", "mx_e3", "mx_e4", "mx_e5",
                  "mx_e6", "pkfma", "b4a", "b4b", "bmax", "umax3", "umax", "dppmax", "smov"}
        last_cos, last_w, last_mfma_d = {}, {}, {}
        for idx, x in enumerate(self.ins):
            reads, writes = set(), set()
            if x.op in VALU_W:
                if x.op == "pkmul":
                    a, tp = x.a
                    reads |= {a, a + 1, tp, tp + 1}
                    writes |= {a, a + 1}
                elif x.op == "pkfma":
                    d, a = x.a
                    reads |= {a, a + 1, g.INV, g.MAGIC}
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
                for r in reads:   # trans -> VALU use: at least one instruction in between
                    assert idx - last_cos.get(r, -10) >= 2, ("trans -> VALU use", idx, x.text)
                for r in reads:   # MFMA D -> VALU read: the tile's last MFMA at least two MFMAs back (or 18 wait states)
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
        g = self.g
        head = ["// GENERATED by csrc/gen/bwd_core.py -- do not edit (tests/test_bwd_core.py checks it is current).",
                f"// dX trunk, width {g.feat}, AUXS = {self.auxs}: {self.stats.get('mfma', 0)} MFMAs, {len(self.ins)} instructions ({len(self.ins) / max(self.stats.get('mfma', 1), 1):.2f} per MFMA), "
                f"{self.stats.get('barrier', 0)} rendezvous, ring of {self.R} pieces, A fragments {self.PF} ahead, {self.FILL} epilogue instructions per gap."]
        return "\n".join(head + ['"' + t + '\\n"' for t in self.text()]) + "\n"


def clobber_file(feat=256):
    g = Geo(feat)
    regs = [r for r in range(g.XREGS if g.Y >= A0 else g.Y, g.N_VGPR) if r not in g.IN_REGS]
    if feat == 256:
        regs = [r for r in range(64, g.N_VGPR) if r not in g.IN_REGS]
    sregs = list(S_SEL) + [S_B4A, S_B4B, S_MAX]
    skip_a = set(range(g.XS - A0, g.XS - A0 + 4)) if g.g1 and g.XS >= A0 else set()
    return (f"// GENERATED by csrc/gen/bwd_core.py: clobber list of the dX trunk statement, width {feat} (the X vector is in/out, the constants and offsets are inputs)\n"
            + ", ".join(f'"v{r}"' for r in regs) + ("".join(f', "a{r}"' for r in range(g.N_AGPR) if r not in skip_a)) + ", " + ", ".join(f'"{s_}"' for s_ in sregs)
            + ', "memory", "scc"\n')


def file_names(feat, auxs, suffix=""):
    stem = "mlp_bwd_trunk" if feat == 256 else "mlp_bwd512_trunk"
    return f"{stem}_a{auxs}{suffix}.inc", f"{stem}_clobbers.inc"


def main():
    """bwd_core.py [out_dir [suffix [ablation,...]]]"""
    out_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1:
        out_dir = sys.argv[1]
    suffix = sys.argv[2] if len(sys.argv) > 2 else ""
    ablate = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ()
    for feat in (256, 512):
        for auxs in (1, 2):
            t = Trunk(auxs, feat=feat, ablate=ablate)
            name, cname = file_names(feat, auxs, suffix)
            with open(os.path.join(out_dir, name), "w") as f:
                f.write(t.inc_file())
            print(feat, auxs, len(t.ins), "instructions,", t.stats.get("mfma"), "MFMAs,", round(len(t.ins) / t.stats.get("mfma"), 2), "per MFMA")
        with open(os.path.join(out_dir, cname), "w") as f:
            f.write(clobber_file(feat))


if __name__ == "__main__":
    sys.exit(main())
