#!/usr/bin/env python3
"""Generator of the weight-gradient kernel's instruction stream, third generation (csrc/wgrad9.hip).

Job (unchanged): dW[row][col] = sum over sample points of dpre[row] * act[col] for one 256 x 256 (+ 32 aux columns) job block over a
slice of 32-point tiles -- autograd's grad_weight / grad_bias of every nn.Linear of SatNeRF (models/satnerf.py:104-153).

What bounds such a kernel on gfx950 (tools/probe_lds.hip, profiles/r04_probe_lds.txt): with ONE wave on a SIMD a simple VALU instruction
issues every ~7.8 cycles (2.7 with four waves), a ds_read_b64_tr_b16 every 16.7, a ds_write_b128 every 55, and an LDS instruction holds
the wave's issue while it is sent -- only the MFMA's execution is asynchronous.  A wave's time is therefore (almost) the SUM of its
non-MFMA issue slots, and what counts is the number of instructions per MFMA, not their nominal cycles.  The r02 / r03 kernels (16 thin
or 8 fat waves, 64 x 64 / 128 x 64 register tiles, operands staged by LDS-DMA and decoded in the LDS, ~10 instructions per MFMA) ran at
2,000-2,100 cycles per 32-point tile against 1,152 matrix cycles.  This design minimises the instruction count per MFMA:
  * 4 waves per workgroup, one per SIMD, each owning a 128 x 128 quadrant (4 x 4 MFMA tiles = 256 fp32 accumulators in AGPRs, two aux
    tiles in VGPRs): one transposed operand read per MFMA instead of two;
  * the 8-bit operands go HBM -> VGPRs (global_load_dwordx4 + the lane's scale byte) and are decoded in registers with PACKED fp16
    arithmetic, the MFMAs run on fp16 operands (v_mfma_f32_32x32x16_f16: same rate as bf16, 11 significand bits instead of 8):
      MX8    (u - 128) 2^(E-133):  one v_perm builds the fp16 pair (1024 + u0, 1024 + u1) (0x6400 | u), one v_pk_fma_f16 scales it
                                   and removes the offset: 1 instruction per value instead of 2.5 through fp32;
      PHASE8 sin(2 pi u / 256):    one v_perm builds the pair (4 + u0 / 256, 4 + u1 / 256) (0x4400 | u), two v_sin_f16 (SDWA, in
                                   place) take revolutions: 1.5 instructions per value instead of 2.5;
    the decoded fragments reach the LDS once (ds_write_b128), nothing is staged raw;
  * fp16's range is fitted per workgroup and per 32-row pair: wgrad9.hip reads the largest exponent of every pair's scale group over its
    slice from the table the dX kernel leaves behind the workspace, the rows of a pair are scaled by G = 2^(138 - Emax) (|value| < 2^12;
    a lane 2^-19 below its group's largest flushes to zero) -- %[erow0] / %[erow1] are the references of the wave's two row duties --
    and the fp32 accumulators are unscaled tile by tile when the partial block is written.

The whole slice loop is ONE asm statement (accumulators never leave the AGPRs), unrolled four tiles deep -- LDS slots and staging
registers rotate with period 4, so every address is static:
    iteration i:  L(i+5)  global loads of tile i + 5 -> staging set (i + 5) % 4  (three tiles of lead)
                  M(i)    36 MFMAs (2 k-steps x (4 x 4 + 2 aux tiles)), operands read transposed from LDS slot i % 4; the operands of
                          k-step 1 and of the next tile's k-step 0 are read one k-step ahead, one operand per MFMA gap
                  PUBLISH tile i + 1 (decoded during iteration i - 1; its writes are waited for a few MFMAs into this iteration, when
                          they have long retired): one ds_add on its slot's counter
                  D(i+2)  decode of the wave's four double fragments + one raw bf16 fragment of tile i + 2 -> LDS slot (i + 2) % 4
                  CONSUME before the prefetch of tile i + 1: its slot's counter must show all four waves' publishes (read early, checked
                          by scalar compare; polled on the rare miss).  No s_barrier in the loop: a barrier per tile cost 250 of 2,000
                          cycles in arrival skew.  Four slots and a decode distance of two make the publish order also the
                          write-after-read order (see consume_check)
LDS operations return in order and are waited for by count.

What a wave loads is a table of five "duties" built on the host (packing.wgrad9_duties): two row double fragments (MX8, from dpre), two
column double fragments (PHASE8, or MX8 for the feats columns: the two variants of this stream), one raw bf16 fragment (aux columns or a
bf16 row fragment: d_sigma_pre / d_head; converted to fp16 with the row scale).  A duty the block does not need points at a dump area
behind the operand fragments; narrow blocks simply contract stale LDS rows nobody reads the results of, so ONE stream serves every block.

Registers:  a[0:255] acc(a, c) at 16 (4 a + c);  v[0:31] aux accumulators;  v[32:67] / v[68:103] operands of k-step 0 / 1 (A0-A3, B0-B3, X);
v[104:167] staging (4 sets x 4 DF x 4);  v[168:183] scale bytes (4 x 4);  v[184:199] raw duty (4 x 4);  v[200:215] decode temporaries and
constants;  v[216:231] transposed-read addresses;  v[232:236] LDS write bases;  v[237:239] per-lane global byte offsets (dpre / acts /
raw duty);  v[240:245] inputs (wgrad9.hip);  v246 write address.
``python wgrad9_loop.py`` writes csrc/wgrad9_loop_{p,m}.inc and csrc/wgrad9_loop_clobbers.inc.
"""
from __future__ import annotations

import os
import sys

FRAG = 1088
PAIR = 2 * FRAG
NFRAG = 36                      # 16 row + 16 column + 2 aux + 2 dump fragments per slot
SLOT = NFRAG * FRAG
NSLOT = 4
AUX_PAIR = 16
OPS = ("B0", "A0", "A1", "A2", "A3", "B1", "B2", "B3", "X")    # read order within a k-step
OPREG = {"A0": 0, "A1": 4, "A2": 8, "A3": 12, "B0": 16, "B1": 20, "B2": 24, "B3": 28, "X": 32}
KBUF = (32, 68)
STG, SCS, RAWX = 104, 168, 184
OUT, T0, SCL, BIAS, SX, K64, K44 = 200, 208, 210, 211, 212, 214, 215   # T0 / SX: even-aligned pairs
RD = {"AL": 216, "AH": 220, "B": 224, "X": 228}   # + 2 * ks + (slot >> 1): one address register serves two slots (16-bit offsets)
WB, WBX = 232, 236
VOFF_D, VOFF_A, VOFF_X = 237, 238, 239
IN_RD0, IN_RD1, IN_LANE16, IN_VD, IN_VA, IN_VX = 240, 241, 242, 243, 244, 245
WADDR, VFLAG, VONE, VSEEN = 246, 247, 248, 249   # VFLAG: LDS address of the four per-slot publish counters; VSEEN: a counter read back
N_VGPR = 250
FLAGS_OFF = NSLOT * SLOT                           # the counters live right behind the four slots (zeroed by wgrad9.hip)
XACC = (0, 16)
S_ADV, S_T0, S_T1, S_T2, S_T3, S_SEL01, S_SEL23 = "s90", "s91", "s92", "s93", "s94", "s95", "s96"    # scratch scalars (clobbers)
S_EXP, S_SEEN = "s97", "s98"                       # publishes a complete tile of this trip's slots has seen (4 per tile); a counter read back
SEL01, SEL23 = "0x04010400", "0x04030402"   # v_perm_b32 selectors: bytes (b0, K, b1, K) / (b2, K, b3, K)


def acc(a, c):
    return 16 * (4 * a + c)


# Thin streams (r06): the job blocks whose row operand is ONE raw bf16 fragment (the head rows d_head, d_sigma_pre: one live 32-row pair) used
# to run the full stream -- two dump row duties decoded at full price, 36 MFMAs on stale rows -- so a narrow block cost a workgroup what a
# full 256 x 256 block costs.  A thin stream is the same slice loop with only what such a block needs: `n_df` column double fragments
# (PHASE8, read from the activation workspace through duty slots 0 .. n_df - 1), optionally the raw fragment, and either the 5 MFMAs per
# k-step of ONE row pair (A0 x B0..B3 + the aux tile) or none at all (a wave that only decodes for the others).  Same LDS image, same
# publish / consume protocol, same registers.  THIN lists the variants wgrad9.hip compiles: (n_df, raw, mfma) -> file tag.
THIN = {(1, True, "thin"): "t1", (3, False, "none"): "d3", (0, True, "thin"): "t0", (2, True, "none"): "r2", (2, False, "none"): "d2",
        (0, False, "none"): "d0"}


class Stream:
    def __init__(self, col_codec, ablate=(), main=True, thin=None, raw=True):
        assert col_codec in ("phase", "mx")
        self.col_codec = col_codec
        self.main = main               # False: a wave whose quadrant nobody reads (narrow blocks): aux tiles only, same loads / decode / rendezvous
        self.ablate = set(ablate)      # timing experiments (wrong results): nomfma, noload, nodec, noread, nowrite, nobar
        self.thin = thin               # None, or (n_df, raw, mfma) of THIN
        if thin is not None:
            assert col_codec == "phase" and thin in THIN
            self.n_df, self.raw, self.mfma_set = thin
            self.duties = list(range(self.n_df))
            self.ns = 0
        else:
            # raw = False (r06): a wave of a full block whose raw duty is a dump (three of the four waves of most blocks) runs the stream without
            # the raw fragment's load, conversion (21 VALU) and LDS write -- work nobody reads
            self.n_df, self.raw, self.mfma_set = 4, bool(raw), ("full" if main else "aux")
            self.duties = [0, 2, 1, 3]                 # decode order: row, column, row, column
            self.ns = 2 if col_codec == "phase" else 4     # scale-byte loads per tile
        self.nld = self.n_df + self.ns + (1 if self.raw else 0)   # global loads per tile and wave
        # the operands a k-step reads (in read order) and the MFMAs it issues at which of the 18 slots of the k-step
        if self.mfma_set == "full":
            self.rq = list(OPS)
        elif self.mfma_set == "aux":
            self.rq = ["A0", "A1", "X"]
        elif self.mfma_set == "thin":
            self.rq = ["B0", "A0", "B1", "B2", "B3", "X"]
        else:
            self.rq = []
        self.ins = []
        self.lgkm = []
        self.n = {}
        self.in_loop = False
        self.states = []
        self._build()

    # ---- emission ------------------------------------------------------------------------------------------------------------
    def e(self, text, kind="valu"):
        ab = self.ablate
        if self.in_loop and ((kind == "mfma" and "nomfma" in ab) or (kind == "vmem" and "noload" in ab) or (text == "s_barrier" and "nobar" in ab)
                             or (kind == "dec" and "nodec" in ab) or ("noload" in ab and text.startswith("s_waitcnt vmcnt"))):
            return
        self.ins.append(text)
        self.n[kind] = self.n.get(kind, 0) + 1

    def lds(self, tag, text):
        if self.in_loop and (("noread" in self.ablate and text.startswith("ds_read")) or ("nowrite" in self.ablate and text.startswith("ds_write"))):
            return
        if len(self.lgkm) >= 15:                       # the counter has 4 bits: retire the older half in ONE wait (they were issued >= 3 MFMAs ago)
            self.wait_for(self.lgkm[len(self.lgkm) - 8])
        self.lgkm.append(tag)
        self.e(text, "lds")

    def wait_for(self, tag):
        """wait until the NEWEST in-flight operation carrying `tag` (and everything older) has returned"""
        if tag not in self.lgkm:
            return
        newest = len(self.lgkm) - 1 - self.lgkm[::-1].index(tag)
        keep = len(self.lgkm) - 1 - newest
        self.e(f"s_waitcnt lgkmcnt({keep})", "wait")
        self.lgkm = self.lgkm[len(self.lgkm) - keep:] if keep else []

    # ---- pieces --------------------------------------------------------------------------------------------------------------
    def read_operand(self, tile_tag, ks, slot, name):
        """the two transposed reads of one MFMA operand (32 rows x 16 points) of k-step ks from LDS slot `slot`"""
        if name == "X":
            cls, imm = "X", 0
        elif name[0] == "B":
            cls, imm = "B", int(name[1]) * PAIR
        else:
            j = int(name[1])
            cls, imm = ("AL" if j < 2 else "AH"), (j & 1) * PAIR
        base = RD[cls] + 2 * ks + (slot >> 1)
        imm += (slot & 1) * SLOT
        dst = KBUF[ks] + OPREG[name]
        for rd in range(2):   # the second read's points are 4 further: + 64 bytes in both halves of the rotated image
            self.lds((tile_tag, ks, name, rd), f"ds_read_b64_tr_b16 v[{dst + 2 * rd}:{dst + 2 * rd + 1}], v{base} offset:{imm + 64 * rd}")

    def mfma(self, ks, a, c):
        self.wait_for(("cur", ks, f"A{a}", 1))
        self.wait_for(("cur", ks, f"B{c}", 1))
        A, B, d = KBUF[ks] + OPREG[f"A{a}"], KBUF[ks] + OPREG[f"B{c}"], acc(a, c)
        self.e(f"v_mfma_f32_32x32x16_f16 a[{d}:{d + 15}], v[{A}:{A + 3}], v[{B}:{B + 3}], a[{d}:{d + 15}]", "mfma")

    def mfma_aux(self, ks, j):
        self.wait_for(("cur", ks, f"A{j}", 1))
        self.wait_for(("cur", ks, "X", 1))
        A, B, d = KBUF[ks] + OPREG[f"A{j}"], KBUF[ks] + OPREG["X"], XACC[j]
        self.e(f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{A}:{A + 3}], v[{B}:{B + 3}], v[{d}:{d + 15}]", "mfma")

    def load_items(self, sset):
        """L(t): the wave's global loads of one tile into staging set `sset`, then the per-lane offsets step to the next tile (clamped
        at the last tile of the workspace).  Closures, one instruction each."""
        it = []
        V = lambda text: it.append(lambda: self.e(text))                     # noqa: E731
        S = lambda text: it.append(lambda: self.e(text, "salu"))             # noqa: E731
        M = lambda text: it.append(lambda: self.e(text, "vmem"))             # noqa: E731
        for d in range(self.n_df):
            r = STG + 16 * sset + 4 * d
            from_acts = d >= 2 or self.thin is not None   # (thin streams: every double fragment is a column duty)
            M(f"global_load_dwordx4 v[{r}:{r + 3}], v{VOFF_A if from_acts else VOFF_D}, %[b{d}]")
        for d in range(self.ns):
            M(f"global_load_ubyte v{SCS + 4 * sset + d}, v{VOFF_D if d < 2 else VOFF_A}, %[sb{d}]")
        if self.raw:
            r = RAWX + 4 * sset
            M(f"global_load_dwordx4 v[{r}:{r + 3}], v{VOFF_X}, %[bx]")
        S(f"s_cmp_lg_u32 {S_ADV}, 0")
        S(f"s_cselect_b32 {S_T0}, %[strd], 0")
        S(f"s_cselect_b32 {S_T1}, %[stra], 0")
        S(f"s_cselect_b32 {S_T2}, %[strx], 0")
        S(f"s_cselect_b32 {S_T3}, 1, 0")
        S(f"s_sub_u32 {S_ADV}, {S_ADV}, {S_T3}")
        V(f"v_add_u32 v{VOFF_D}, {S_T0}, v{VOFF_D}")
        V(f"v_add_u32 v{VOFF_A}, {S_T1}, v{VOFF_A}")
        V(f"v_add_u32 v{VOFF_X}, {S_T2}, v{VOFF_X}")
        return it

    def decode_items(self, d, sset, slot):
        """D: double fragment d (0, 1 rows: MX8; 2, 3 columns: self.col_codec) of staging set `sset` -> two fp16 fragments in LDS slot
        `slot`.  Value n = byte n & 3 of raw dword n >> 2 -> output dword n >> 1 (codec8.h)."""
        codec = self.col_codec if (d >= 2 or self.thin is not None) else "mx"
        raw = STG + 16 * sset + 4 * d
        it = []
        V = lambda text: it.append(lambda: self.e(text, "dec"))              # noqa: E731
        L = lambda tag, text: it.append(lambda: self.lds(tag, text))          # noqa: E731
        it.append(lambda: self.e(f"v_add_u32 v{WADDR}, {slot * SLOT}, v{WB + d}"))
        if codec == "mx":
            sc = SCS + 4 * sset + d
            # fp16 scale 2^(E - Eref - 15): exponent field E - Eref (Eref = Emax - 20), flushed to zero below fp16's normal range
            V(f"v_subrev_u32 v{SCL}, %[{f'erow{d}' if d < 2 else 'ecol'}], v{sc}")   # the reference exponent of this duty's row pair / of the columns
            V(f"v_max_i32 v{SCL}, 0, v{SCL}")
            V(f"v_lshlrev_b32 v{SCL}, 10, v{SCL}")
            V(f"v_mul_f16 v{BIAS}, 0xe480, v{SCL}")                            # bias = -1152 scale: the 1024 of the magic number + the 128 of the offset
            for q in range(8):                                                # output dword q = values 2 q, 2 q + 1
                V(f"v_perm_b32 v{T0}, v{K64}, v{raw + (q >> 1)}, {S_SEL01 if q & 1 == 0 else S_SEL23}")
                V(f"v_pk_fma_f16 v{OUT + q}, v{T0}, v{SCL}, v{BIAS} op_sel_hi:[1,0,0]")
        else:
            # one v_perm builds the pair (4 + u0 / 256, 4 + u1 / 256) revolutions in fp16, v_sin_f16 takes revolutions (period 1).  The second
            # sine of a dword (SDWA, preserving the other half) reads the first one's result: they are issued two apart
            for q in range(8):
                V(f"v_perm_b32 v{OUT + q}, v{K44}, v{raw + (q >> 1)}, {S_SEL01 if q & 1 == 0 else S_SEL23}")
            for q0 in range(0, 8, 2):
                for half in (0, 1):
                    for q in (q0, q0 + 1):
                        V(f"v_sin_f16_sdwa v{OUT + q}, v{OUT + q} dst_sel:WORD_{half} dst_unused:UNUSED_PRESERVE src0_sel:WORD_{half}")
            V("s_nop 0")
        L(("w", d, 0), f"ds_write_b128 v{WADDR}, v[{OUT}:{OUT + 3}]")
        L(("w", d, 1), f"ds_write_b128 v{WADDR}, v[{OUT + 4}:{OUT + 7}] offset:{FRAG}")
        return it

    def rawcopy_items(self, sset, slot):
        """the raw bf16 fragment (aux columns / a bf16 row fragment) -> fp16, times the duty's scale (1 or the row scale G)"""
        it = []
        V = lambda text: it.append(lambda: self.e(text, "dec"))              # noqa: E731
        r = RAWX + 4 * sset
        it.append(lambda: self.e(f"v_add_u32 v{WADDR}, {slot * SLOT}, v{WBX}"))
        for q in range(4):
            V(f"v_lshlrev_b32 v{T0}, 16, v{r + q}")
            V(f"v_and_b32 v{T0 + 1}, 0xffff0000, v{r + q}")
            V(f"v_mul_f32 v{T0}, v{T0}, v{SX}")       # (two plain multiplies: a v_pk_mul_f32 holds the matrix pipe for 7-10 cycles,
            V(f"v_mul_f32 v{T0 + 1}, v{T0 + 1}, v{SX}")   #  tools/probe_power.hip)
            V(f"v_cvt_pk_f16_f32 v{OUT + q}, v{T0}, v{T0 + 1}")
        it.append(lambda: self.lds(("w", "x", 0), f"ds_write_b128 v{WADDR}, v[{OUT}:{OUT + 3}]"))
        return it

    def decode_tile(self, sset, slot):
        it = []
        if self.nld:
            it.append(lambda: self.e(f"s_waitcnt vmcnt({3 * self.nld})", "wait"))    # only the three newest tiles may still be in flight
        for d in self.duties:
            it += self.decode_items(d, sset, slot)
        if self.raw:
            it += self.rawcopy_items(sset, slot)
        return it

    def publish(self, slot):
        """this wave's fragments of a tile are in LDS slot `slot` (their writes have been waited for): bump the slot's publish counter"""
        # (an LDS atomic is a per-lane operation: one lane adds, or the counter would move by 64)
        if "lane64" in self.ablate:   # debugging aid: every lane adds (the counter moves by 64 per wave), EXEC untouched
            self.lds(("flag", "add"), f"ds_add_u32 v{VFLAG}, v{VONE} offset:{4 * slot}")
            return
        self.e("s_mov_b64 exec, 1", "salu")
        self.lds(("flag", "add"), f"ds_add_u32 v{VFLAG}, v{VONE} offset:{4 * slot}")
        self.e("s_mov_b64 exec, -1", "salu")

    def consume_fetch(self, slot):
        self.lds(("flag", "rd"), f"ds_read_b32 v{VSEEN}, v{VFLAG} offset:{4 * slot}")

    def consume_check(self, slot, extra):
        """all four waves have published the tile in `slot` (counter >= S_EXP + extra) before anyone reads it.  With four slots and a
        decode distance of two this also orders the writes of iteration i behind every wave's reads of iteration i - 2: a wave publishes
        tile i only after its own reads of tile i - 2 (same slot as tile i + 2) have returned.  The counter was read a few MFMAs ago;
        the slow path re-reads it until the last wave has arrived (rare: the tile was written a whole iteration ago)."""
        if "nobar" in self.ablate and self.in_loop:
            self.wait_for(("flag", "rd"))
            return
        self.wait_for(("flag", "rd"))
        lbl = self.next_label = getattr(self, "next_label", 10) + 1
        self.e(f"v_readfirstlane_b32 {S_SEEN}, v{VSEEN}", "salu")
        self.e(f"s_sub_u32 {S_SEEN}, {S_SEEN}, {S_EXP}", "salu")
        if "lane64" in self.ablate:
            extra *= 64
        self.e(f"s_cmp_ge_i32 {S_SEEN}, {extra}", "salu")
        self.e(f"s_cbranch_scc1 {lbl}f", "salu")
        # slow path: drain (so that the bookkeeping below stays a lower bound of what has returned), poll
        self.e(f"{lbl + 100}:", "label")
        self.e(f"ds_read_b32 v{VSEEN}, v{VFLAG} offset:{4 * slot}", "lds")
        self.e("s_waitcnt lgkmcnt(0)", "wait")
        self.e(f"v_readfirstlane_b32 {S_SEEN}, v{VSEEN}", "salu")
        self.e(f"s_sub_u32 {S_SEEN}, {S_SEEN}, {S_EXP}", "salu")
        self.e(f"s_cmp_ge_i32 {S_SEEN}, {extra}", "salu")
        self.e(f"s_cbranch_scc0 {lbl + 100}b", "salu")
        self.e(f"{lbl}:", "label")
        if "bar" in self.ablate:   # debugging aid: a real barrier on top of the counters
            self.e("s_barrier", "salu")

    # ---- the statement -------------------------------------------------------------------------------------------------------
    def body(self, s):
        """one tile: L(i + 5), M(i) on slot s, rendezvous, D(i + 2), K0(i + 1)"""
        # the prefetched k-step-0 operands are now the current tile's
        self.lgkm = [("cur",) + t[1:] if t[0] == "nxt" else t for t in self.lgkm]
        if s == 0 and not self.rq:
            # a stream without operand reads has no wait at the top of the first body: over the back edge the fourth body's late writes
            # are still in flight (the bookkeeping at the loop label is the prologue's: nothing) -- drain them before the publish below
            self.e("s_waitcnt lgkmcnt(0)", "wait")
            self.lgkm = []
        nslot = (s + 1) % 4
        def publish_prev():   # the previous iteration's decode (tile i + 1): its writes are older than anything issued since, long retired
            w = [t for t in self.lgkm if t[0] == "w"]
            if w:
                self.wait_for(w[-1])
            self.publish(nslot)

        fillers = self.load_items((s + 1) % 4) + [publish_prev] + self.decode_tile((s + 2) % 4, (s + 2) % 4)
        order = [(a, c) for c in range(4) for a in range(4)]
        per_gap = -(-len(fillers) // 34)
        fi = 0

        def fill(k):
            nonlocal fi
            for _ in range(k):
                if fi < len(fillers):
                    fillers[fi]()
                    fi += 1

        for ks in range(2):
            rq = list(self.rq)
            for idx in range(18):
                if idx == 0:   # every operand of this k-step has landed (they were read one k-step ago): one wait instead of one per MFMA
                    self.wait_for(("cur", ks, "X", 1))
                    if ks == 1:  # the next tile is complete (all four waves) before its first operand is prefetched; tile i + 1 of the
                        self.consume_check(nslot, 4 if s < 3 else 8)  # fourth body belongs to the next trip
                if idx < 16:
                    if self.mfma_set == "full" or (self.mfma_set == "thin" and order[idx][0] == 0):
                        self.mfma(ks, *order[idx])
                elif self.mfma_set in ("full", "aux") or (self.mfma_set == "thin" and idx == 16):
                    self.mfma_aux(ks, idx - 16)
                if ks == 0 and idx == 12:   # the other waves published tile i + 1 some ten MFMAs ago: read its counter now, compare at the k-step's end
                    self.consume_fetch(nslot)
                if rq and idx < 9:     # one operand (two reads) of the next k-step per gap, all nine under way by gap 8
                    name = rq.pop(0)
                    if ks == 0:
                        self.read_operand("cur", 1, s, name)
                    else:
                        self.read_operand("nxt", 0, (s + 1) % 4, name)
                fill(per_gap)
        while fi < len(fillers):
            fillers[fi]()
            fi += 1
        return list(self.lgkm)

    def _build(self):
        e = self.e
        # ---- prologue ---------------------------------------------------------------------------------------------------------
        for r in range(256):
            e(f"v_accvgpr_write_b32 a{r}, 0")
        for r in range(32):
            e(f"v_mov_b32 v{r}, 0")
        e(f"v_mov_b32 v{K64}, 0x64646464")
        e(f"v_mov_b32 v{K44}, 0x44444444")
        e(f"v_mov_b32 v{SX}, %[sraw]")
        e(f"v_mov_b32 v{SX + 1}, %[sraw]")
        e(f"v_mov_b32 v{VONE}, 1")
        e(f"v_mov_b32 v{VFLAG}, %[flags]")
        e(f"s_mov_b32 {S_EXP}, 0", "salu")
        e(f"s_mov_b32 {S_SEL01}, {SEL01}", "salu")
        e(f"s_mov_b32 {S_SEL23}, {SEL23}", "salu")
        for ks in range(2):
            src = IN_RD0 + ks
            for half in range(2):
                off = 2 * half * SLOT
                for cls, sreg in (("AL", "%[aofl]"), ("AH", "%[aofh]"), ("B", "%[bof]")):
                    r = RD[cls] + 2 * ks + half
                    e(f"v_add_u32 v{r}, {sreg}, v{src}")
                    if off:
                        e(f"v_add_u32 v{r}, {off}, v{r}")
                e(f"v_add_u32 v{RD['X'] + 2 * ks + half}, {AUX_PAIR * PAIR + off}, v{src}")
        for d in range(4):
            e(f"v_add_u32 v{WB + d}, %[w{d}], v{IN_LANE16}")
        e(f"v_add_u32 v{WBX}, %[wx], v{IN_LANE16}")
        e(f"v_mov_b32 v{VOFF_D}, v{IN_VD}")
        e(f"v_mov_b32 v{VOFF_A}, v{IN_VA}")
        e(f"v_mov_b32 v{VOFF_X}, v{IN_VX}")
        e(f"s_mov_b32 {S_ADV}, %[tleft]", "salu")
        e("s_cmp_eq_u32 %[nt], 0", "salu")
        e("s_cbranch_scc1 9f", "salu")
        for t in range(4):
            for f in self.load_items(t):
                f()
        for f in self.decode_tile(0, 0):
            f()
        e("s_waitcnt lgkmcnt(0)", "wait")
        self.lgkm = []
        self.publish(0)
        for f in self.load_items(0):
            f()
        for f in self.decode_tile(1, 1):
            f()
        e("s_waitcnt lgkmcnt(0)", "wait")   # (tile 1 is published by the first iteration, as every iteration publishes its predecessor's decode)
        self.lgkm = []
        e("s_barrier", "salu")
        for name in self.rq:
            self.read_operand("nxt", 0, 0, name)
        # ---- the loop: four tiles per trip -------------------------------------------------------------------------------------
        # LDS-counter bookkeeping at the loop label: the first body is generated from the prologue's state (the 18 reads of K0(0) in
        # flight, issued last); over the back edge it is entered with the fourth body's late writes in flight as well, all NEWER than the
        # prefetched reads.  The first body opens with the wait for the newest of those reads, computed as lgkmcnt(0) here: over the back
        # edge it waits for the late writes too (stricter than needed, never weaker), after which model and machine agree again.
        e("1:", "label")
        self.in_loop = True
        for s in range(4):
            if s == 0 and self.rq:
                assert self.lgkm and self.lgkm[-1][:3] == ("nxt", 0, "X"), self.lgkm[-3:]
            self.states.append(self.body(s))
            e("s_sub_u32 %[nt], %[nt], 1", "salu")
            e("s_cmp_eq_u32 %[nt], 0", "salu")
            e("s_cbranch_scc1 9f", "salu")
        e(f"s_add_u32 {S_EXP}, {S_EXP}, {256 if 'lane64' in self.ablate else 4}", "salu")
        e("s_branch 1b", "salu")
        self.in_loop = False
        e("9:", "label")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait")
        for _ in range(3):
            e("s_nop 15", "salu")    # the last MFMAs' results -> the epilogue's v_accvgpr_read (inline asm is not hazard-padded)

    def inc_file(self):
        kind = ("" if self.raw else "no raw duty; ") if self.thin is None else f"thin stream {self.thin}; "
        head = [f"// GENERATED by csrc/gen/wgrad9_loop.py -- do not edit.  {kind}Column codec: {self.col_codec}; {len(self.ins)} lines: "
                + ", ".join(f"{v} {k}" for k, v in sorted(self.n.items()))]
        return "\n".join(head + ['"' + t + '\\n"' for t in self.ins]) + "\n"


def clobber_file():
    regs = [f'"v{r}"' for r in range(32, N_VGPR) if not IN_RD0 <= r <= IN_VX]
    return ("// GENERATED by csrc/gen/wgrad9_loop.py: clobber list of the slice-loop statement (accumulators are outputs, v240..v245 inputs)\n"
            + ", ".join(regs) + f', "{S_ADV}", "{S_T0}", "{S_T1}", "{S_T2}", "{S_T3}", "{S_SEL01}", "{S_SEL23}", "{S_EXP}", "{S_SEEN}", "memory", "scc"\n')


def main():
    """wgrad9_loop.py [out_dir [suffix [ablation,...]]]"""
    out_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1:
        out_dir = sys.argv[1]
    suffix = sys.argv[2] if len(sys.argv) > 2 else ""
    ablate = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ()
    for codec, tag in (("phase", "p"), ("mx", "m"), ("phase", "px"), ("mx", "mx")):
        s = Stream(codec, ablate=ablate, main=len(tag) == 1)
        with open(os.path.join(out_dir, f"wgrad9_loop_{tag}{suffix}.inc"), "w") as f:
            f.write(s.inc_file())
        print(tag, s.n, len(s.ins), "LDS operations in flight at body ends:", [len(x) for x in s.states])
    for codec, tag in (("phase", "pn"), ("mx", "mn"), ("phase", "pxn"), ("mx", "mxn")):   # the same four without the raw duty
        s = Stream(codec, ablate=ablate, main=len(tag) == 2, raw=False)
        with open(os.path.join(out_dir, f"wgrad9_loop_{tag}{suffix}.inc"), "w") as f:
            f.write(s.inc_file())
        print(tag, s.n, len(s.ins), "LDS operations in flight at body ends:", [len(x) for x in s.states])
    for thin, tag in THIN.items():
        s = Stream("phase", ablate=ablate, thin=thin)
        with open(os.path.join(out_dir, f"wgrad9_loop_{tag}{suffix}.inc"), "w") as f:
            f.write(s.inc_file())
        print(tag, s.n, len(s.ins), "LDS operations in flight at body ends:", [len(x) for x in s.states])
    with open(os.path.join(out_dir, "wgrad9_loop_clobbers.inc"), "w") as f:
        f.write(clobber_file())


if __name__ == "__main__":
    sys.exit(main())
