#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of the dX kernel's TRUNK (csrc/mlp_bwd.inc, 8-bit workspaces, width 256).

The seven transposed trunk layers bL7 .. bL1 are two thirds of the data-gradient kernel (896 of its 1,308 MFMAs per 32-point tile):
  d a_{l-1} = W_l^T d pre_l (8 output tiles x 16 k-steps),   d pre_{l-1} = d a_{l-1} * cos(2 pi phase_{l-1}),
with d pre_{l-1} handed on in registers as bf16 B fragments and written to the dpre workspace as MX8 (codec8.h) for the
weight-gradient kernel.  Same recipe as csrc/gen/fwd_core.py (whose measurements justify it: profiles/r03_coissue.txt,
r03_ab_variants.txt): MFMA i consumes piece i of the trunk's part of the transposed stream from a flat LDS ring fed by LDS-DMA rows of 8
pieces; A fragments are read PF MFMAs ahead; the epilogue of tile t-1 -- PHASE8 decode (cvt, scale, v_cos), multiply, pack, MX8 encode,
non-temporal store: ~120 VALU -- sits in the gaps of tile t's MFMAs.  The tile's saved phases (one 16-byte load per lane) are fetched
two tiles ahead straight into registers; loads retire in order, so every vmcnt wait is an exact count of the loads (LDS-DMA rows and
phase loads) issued after the one needed -- stores are left out of the count (they may complete out of order: a wait can only get longer).

`python bwd_core.py` writes csrc/mlp_bwd_trunk.inc and csrc/mlp_bwd_trunk_clobbers.inc.

Registers: v[0:63] X, v[64:127] Y (d pre vectors, ping-pong; operands), v[128:159] two accumulators, v[160:183] A ring,
v[184:199] phase ring (4 tiles), v[200:203] temporaries, v[204:211] two store quads, v[212:213] scale bytes, v[214:217] MX8
temporaries, v218 = 128.0, v219 / v220 LDS read bases, v221 stream offset, v222 phase offset, v223 dpre offset (operands).
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
MXT = 214
K128, VL0, VL1, VOFF, POFF, SOFF = 218, 219, 220, 221, 222, 223
N_VGPR = 224
NW, KS, MT, LAYERS = 8, 16, 8, 7
D8_SCALE = 94  # kD8Scale (mlp_layout.h, width 256)


class Trunk:
    def __init__(self, auxs, R=96, PF=5, GROUP=2, FILL=8):
        self.auxs, self.R, self.PF, self.GROUP, self.FILL = auxs, R, PF, GROUP, FILL
        self.ins = []          # (kind, text)
        self.vm = []           # outstanding vector-memory LOADS in issue order (tags)
        self.p_unit = None     # unit POFF / SOFF currently point at
        self.s_unit = None
        self._build()

    def e(self, kind, text):
        self.ins.append((kind, text))

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
        self.e("salu", f"s_add_u32 m0, %[wb], {imm}")
        self.e("nop", "s_nop 0")
        self.e("dma", f"global_load_lds_dwordx4 v{VOFF}, %[sb]")
        self.e("valu", f"v_add_u32 v{VOFF}, 0x2000, v{VOFF}")
        self.vm_issue(("row", j))

    def phase_load(self, tau):
        l, t = 7 - tau // MT, tau % MT           # layer bL_l multiplies by cos(phase a_{l-1}): unit A + 8 (l - 1) + t
        unit = self.auxs + 8 * (l - 1) + t
        delta = (unit - self.p_unit) * 1024
        self.p_unit = unit
        if delta:
            self.e("valu", f"v_add_u32 v{POFF}, 0x{delta & 0xffffffff:x}, v{POFF}")
        r = PH0 + 4 * (tau % NPH)
        self.e("vmem", f"global_load_dwordx4 v[{r}:{r + 3}], v{POFF}, %[ab] nt")
        self.vm_issue(("ph", tau))

    def epilogue_items(self, tau):
        """closures, one instruction each, of tile tau's epilogue (accumulator ACC[tau & 1], phases PH[tau % 4])"""
        l, t = 7 - tau // MT, tau % MT
        a, ph, out = ACC[tau & 1], PH0 + 4 * (tau % NPH), (Y if (7 - l) % 2 == 0 else X) + 8 * t
        sv = SV[tau & 1]
        m, ex, inv, tmp = MXT, MXT + 1, MXT + 2, MXT + 3
        it = []
        V = lambda text: it.append(lambda: self.e("valu", text))   # noqa: E731

        def wait_phase():
            keep = self.vm_wait(("ph", tau))
            if keep is not None:
                self.e("wait", f"s_waitcnt vmcnt({keep})")
        it.append(wait_phase)
        for q in range(8):   # values 2 q, 2 q + 1: d pre = acc * cos(2 pi u / 256)
            g0, g1 = 2 * q, 2 * q + 1
            ta, tb = T0 + (g0 & 3), T0 + (g1 & 3)
            V(f"v_cvt_f32_ubyte{g0 & 3} v{ta}, v{ph + (g0 >> 2)}")
            V(f"v_cvt_f32_ubyte{g1 & 3} v{tb}, v{ph + (g1 >> 2)}")
            V(f"v_mul_f32 v{ta}, 0x3b800000, v{ta}")
            V(f"v_mul_f32 v{tb}, 0x3b800000, v{tb}")
            V(f"v_cos_f32 v{ta}, v{ta}")
            V(f"v_cos_f32 v{tb}, v{tb}")
            V(f"v_mul_f32 v{a + g0}, v{a + g0}, v{ta}")
            V(f"v_mul_f32 v{a + g1}, v{a + g1}, v{tb}")
            V(f"v_cvt_pk_bf16_f32 v{out + q}, v{a + g0}, v{a + g1}")
            if q == 0:
                V(f"v_max_f32 v{m}, |v{a}|, |v{a + 1}|")
            else:
                V(f"v_max3_f32 v{m}, |v{a + g0}|, |v{a + g1}|, v{m}")
        # MX8: E = exponent of 1.0079 max|v| clamped to [6, 254]; u = cvt_u8(v * 2^(133 - E) + 128)
        V(f"v_fmac_f32 v{m}, 0x3c000000, v{m}")
        V(f"v_lshrrev_b32 v{ex}, 23, v{m}")
        V(f"v_max_u32 v{ex}, 6, v{ex}")
        V(f"v_min_u32 v{ex}, 0xfe, v{ex}")
        V(f"v_sub_u32 v{inv}, 0x104, v{ex}")
        V(f"v_lshlrev_b32 v{inv}, 23, v{inv}")
        if t & 3:
            V(f"v_lshl_or_b32 v{EB + (t >> 2)}, v{ex}, {8 * (t & 3)}, v{EB + (t >> 2)}")
        else:
            V(f"v_mov_b32 v{EB + (t >> 2)}, v{ex}")
        for g in range(16):
            V(f"v_fma_f32 v{tmp}, v{a + g}, v{inv}, v{K128}")
            V(f"v_cvt_pk_u8_f32 v{sv + (g >> 2)}, v{tmp}, {g & 3}, v{sv + (g >> 2)}")

        def store():
            unit = 8 * (l - 1) + t
            delta = (unit - self.s_unit) * 1024
            self.s_unit = unit
            if delta:
                self.e("valu", f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
            self.e("vmem", f"global_store_dwordx4 v{SOFF}, v[{sv}:{sv + 3}], %[db] nt")
        it.append(store)
        if t == MT - 1:  # the layer's eight scale bytes: group l - 1 -> unit kD8Scale + (l - 1) / 2, bytes 8 ((l - 1) % 2) ..
            def store_scale():
                g = l - 1
                unit = D8_SCALE + g // 2
                delta = (unit - self.s_unit) * 1024
                self.s_unit = unit
                self.e("valu", f"v_add_u32 v{SOFF}, 0x{delta & 0xffffffff:x}, v{SOFF}")
                self.e("vmem", f"global_store_dwordx2 v{SOFF}, v[{EB}:{EB + 1}], %[db] offset:{8 * (g % 2)}")
            it.append(store_scale)
        return it

    def _build(self):
        R, PF, G = self.R, self.PF, self.GROUP
        NT = LAYERS * MT                      # 56 tiles = chunks of 16 pieces
        N = NT * KS
        n_rows = N // NW
        self.e("salu", "s_mov_b32 %[m0save], m0")
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
            self.e("wait", f"s_waitcnt vmcnt({keep if keep is not None else len(self.vm)})")
            self.e("barrier", "s_barrier")

        def dsread(i):
            slot = i % R
            base, off = (VL0, slot * 1024) if slot < 64 else (VL1, (slot - 64) * 1024)
            d = AR0 + 4 * (i % NA)
            self.e("lds", f"ds_read_b128 v[{d}:{d + 3}], v{base} offset:{off}")

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
        epi = []          # [earliest gap, closure]
        for i in range(N):
            ti, k = divmod(i, KS)
            l = 7 - ti // MT
            inp = X if (7 - l) % 2 == 0 else Y
            acc = ACC[ti & 1]
            if k == 0:
                # the tile before last's epilogue still reads this accumulator: it must be out (and every B fragment of a new layer
                # is produced by the previous layer's epilogues: tile 7's runs during this tile, k-steps 14, 15 come last)
                while epi and epi[0][2] <= ti - 2:
                    epi.pop(0)[1]()
            if ti % MT == 0 and k >= 14:
                while epi and epi[0][2] < ti:
                    epi.pop(0)[1]()
                if k == 14:
                    self.e("nop", "s_nop 1")  # VALU write -> MFMA operand: two wait states
            self.e("wait", f"s_waitcnt lgkmcnt({min(PF - 1, N - 1 - i)})")
            c = "0" if k == 0 else f"v[{acc}:{acc + 15}]"
            self.e("mfma", f"MF v[{acc}:{acc + 15}], v[{AR0 + 4 * (i % NA)}:{AR0 + 4 * (i % NA) + 3}], v[{inp + 4 * k}:{inp + 4 * k + 3}], {c}")
            if k == KS - 1:
                for f in self.epilogue_items(ti):
                    epi.append([i + 2, f, ti])
            # ---- gap(i)
            if i + PF < N:
                if read_for(i + PF):
                    allow_rows(ti * KS)
                dsread(i + PF)
            if k == 0 and ti + 2 < NT:
                self.phase_load(ti + 2)
            if pending:
                emit_row()
            n = 0
            while epi and n < self.FILL and epi[0][0] <= i:
                epi.pop(0)[1]()
                n += 1
        while epi:
            epi.pop(0)[1]()
        assert not pending and rows_issued == n_rows
        self.e("wait", "s_waitcnt vmcnt(0)")
        self.e("salu", "s_mov_b32 m0, %[m0save]")
        kinds = {}
        for k_, _ in self.ins:
            kinds[k_] = kinds.get(k_, 0) + 1
        self.stats = kinds

    def inc_file(self):
        head = ["// GENERATED by csrc/gen/bwd_core.py -- do not edit (tests/test_fwd_core.py checks it is current).",
                f"// dX trunk, AUXS = {self.auxs}: {self.stats}"]
        return "\n".join(head + ['"' + t + '\\n"' for _, t in self.ins]) + "\n"


def clobber_file():
    regs = [r for r in range(64, N_VGPR) if r not in (K128, VL0, VL1, VOFF, POFF, SOFF)]
    return ("// GENERATED by csrc/gen/bwd_core.py: clobber list of the dX trunk statement (X = v[0:63] and v[218:223] are operands)\n"
            + ", ".join(f'"v{r}"' for r in regs) + ', "memory", "scc"\n')


def main():
    out_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1:
        out_dir = sys.argv[1]
    for auxs in (1, 2):
        t = Trunk(auxs)
        with open(os.path.join(out_dir, f"mlp_bwd_trunk_a{auxs}.inc"), "w") as f:
            f.write(t.inc_file())
        print(auxs, t.stats, len(t.ins))
    with open(os.path.join(out_dir, "mlp_bwd_trunk_clobbers.inc"), "w") as f:
        f.write(clobber_file())


if __name__ == "__main__":
    sys.exit(main())
