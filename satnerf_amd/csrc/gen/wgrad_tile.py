#!/usr/bin/env python3
"""Generator of the hand-placed per-tile instruction stream of the fat-wave weight-gradient kernel (csrc/wgrad8f.hip).

One workgroup = one 256 x 256 (+ 32 aux columns) job block, 8 waves of <= 256 VGPRs (two per SIMD).  Per 32-point tile a wave
  * contracts its 4 x 2 grid of 32 x 32 output tiles (+ one aux tile) over the tile's two k-steps: 18 MFMAs, operands read back
    transposed from the point-major LDS slot with ds_read_b64_tr_b16 (8 operands = 16 reads per k-step);
  * decodes, in place, the two 8-bit double fragments it fetched for the NEXT tile: one MX8 row pair (dpre) and one column pair
    (PHASE8 -> sin, or MX8 for the feats columns): ~100 VALU instructions.
r02's kernel (16 thin waves, hipcc's schedule) ran the two halves one after the other -- MFMA busy 30 %, VALU 36 % of the kernel
(profiles/r02_train_pmc.csv) -- although the pipes overlap fully when the stream is placed by hand (profiles/r03_coissue.txt).
This script emits ONE asm statement per tile: address set-up, the 32 transposed reads software-pipelined against the MFMAs with
counted lgkmcnt waits, the decode spread over the MFMA gaps, the decoded fragments written back.  ``python wgrad_tile.py`` writes
csrc/wgrad8f_tile_{p,m}.inc (column codec PHASE8 / MX8) and csrc/wgrad8f_tile_clobbers.inc.

Registers (operands / clobbers of the statement in csrc/wgrad8f.hip):
  v[0:127]   acc[a][c] at 16 (2 a + c), a = 0..3 row tiles, c = 0..1 column tiles        v[128:143] acc_aux
  v[144:175] operands of k-step 0: B0 A0 A1 A2 A3 B1 XA XB (4 registers each)             v[176:207] the same for k-step 1
  v[208:211] raw bytes   v[212:215] decoded values   v[216:223] packed output   v224 scale   v225 bias   v226 write address
  v227.. addresses of A, B, XA, XB for k-step 0 / 1 (8 registers)
  v235, v236 per-lane transposed-read offset of k-step 0 / 1 (operand), v237 lane * 16, v238 rotated-image lane * 16 (operands)
  v[240:246] the column double fragment's raw bytes, scale, bias, address
  s[60:65] scratch scalars
Scalar operands: %[cur] / %[nxt] LDS address of this tile's / the next tile's slot; %[aoff] %[boff] %[xoff] byte offset inside a
slot of the wave's first row pair, first column pair, aux row pair; %[rraw] %[craw] offset of the row / column double fragment's
raw bytes (= the place of its SECOND decoded fragment; the first goes 1088 bytes lower); %[rsc] %[csc] offset of the scale byte
area + byte (MX8).  The aux column pair sits at pair 16 of every slot.
"""
from __future__ import annotations

import os
import sys

FRAG = 1088           # kFragStride8 (csrc/wgrad8.hip)
PAIR = 2 * FRAG
AUX_PAIR_OFF = 16 * PAIR
OPS = ("B0", "A0", "A1", "A2", "A3", "B1", "XA", "XB")   # read order of a k-step
OPREG = {name: 4 * i for i, name in enumerate(OPS)}
KBUF = (144, 176)
RAW, VAL, OUT, SCL, BIAS, WADDR = 208, 212, 216, 224, 225, 226
ADDR = {"A": 227, "B": 229, "XA": 231, "XB": 233}   # + k-step
RDOFF, LANE16, SRC16 = 235, 237, 238
RAWC, SCLC, BIASC, WADDRC = 240, 244, 245, 246   # the column double fragment's raw bytes / scale / bias / address (fetched with the row one's)
N_VGPR = 247
ACC_AUX = 128
IMM = {"A0": 0, "A1": PAIR, "A2": 4 * PAIR, "A3": 5 * PAIR, "B0": 0, "B1": PAIR, "XA": 0, "XB": 0}


def acc(a, c):
    return 16 * (2 * a + c)


class Tile:
    def __init__(self, col_codec, fill=7):
        assert col_codec in ("phase", "mx")
        self.col_codec, self.fill = col_codec, fill
        self.ins = []      # text lines
        self.lgkm = []     # outstanding LDS operations in issue order
        self.n = dict(mfma=0, valu=0, lds=0, wait=0)
        self._build()

    def e(self, text, kind="valu"):
        self.ins.append(text)
        self.n[kind] = self.n.get(kind, 0) + 1

    # ---- LDS bookkeeping: every LDS operation returns in order, waits are counted ---------------------------------------------------
    def lds(self, tag, text):
        if len(self.lgkm) >= 15:  # the counter has 4 bits
            self.wait_for(self.lgkm[0])
        self.lgkm.append(tag)
        self.e(text, "lds")

    def wait_for(self, tag):
        if tag not in self.lgkm:
            return
        keep = len(self.lgkm) - 1 - self.lgkm.index(tag)
        self.e(f"s_waitcnt lgkmcnt({keep})", "wait")
        self.lgkm = self.lgkm[len(self.lgkm) - keep:] if keep else []

    def read_operand(self, ks, name):
        base = ADDR["XA" if name == "XA" else "XB" if name == "XB" else name[0]] + ks
        dst = KBUF[ks] + OPREG[name]
        for rd in range(2):  # the second read's points are 4 further: + 64 bytes in both halves of the rotated image
            self.lds((ks, name, rd), f"ds_read_b64_tr_b16 v[{dst + 2 * rd}:{dst + 2 * rd + 1}], v{base} offset:{IMM[name] + 64 * rd}")

    def mfma(self, ks, a, c):
        self.wait_for((ks, f"A{a}", 1))
        self.wait_for((ks, f"B{c}", 1))
        A, B, d = KBUF[ks] + OPREG[f"A{a}"], KBUF[ks] + OPREG[f"B{c}"], acc(a, c)
        self.e(f"v_mfma_f32_32x32x16_bf16 v[{d}:{d + 15}], v[{A}:{A + 3}], v[{B}:{B + 3}], v[{d}:{d + 15}]", "mfma")

    def mfma_aux(self, ks):
        self.wait_for((ks, "XA", 1))
        self.wait_for((ks, "XB", 1))
        A, B = KBUF[ks] + OPREG["XA"], KBUF[ks] + OPREG["XB"]
        self.e(f"v_mfma_f32_32x32x16_bf16 v[{ACC_AUX}:{ACC_AUX + 15}], v[{A}:{A + 3}], v[{B}:{B + 3}], v[{ACC_AUX}:{ACC_AUX + 15}]", "mfma")

    def decode_items(self, which, part):
        """one double fragment's decode as a list of closures (each emits one instruction or wait): 'r' = MX8 row pair, 'c' = column
        pair in self.col_codec.  part 'fetch': addresses + the LDS reads of the raw bytes and the scale byte (issued early, for both
        fragments, so that nothing waits on them right after issue); part 'compute': everything else."""
        codec = "mx" if which == "r" else self.col_codec
        raw_off = "%[rraw]" if which == "r" else "%[craw]"
        sc_off = "%[rsc]" if which == "r" else "%[csc]"
        raw, scl, bias, waddr = (RAW, SCL, BIAS, WADDR) if which == "r" else (RAWC, SCLC, BIASC, WADDRC)
        it = []
        V = lambda text: it.append(lambda: self.e(text))                       # noqa: E731
        S = lambda text: it.append(lambda: self.e(text, "salu"))               # noqa: E731
        L = lambda tag, text: it.append(lambda: self.lds(tag, text))            # noqa: E731
        W = lambda tag: it.append(lambda: self.wait_for(tag))                   # noqa: E731
        if part == "fetch":
            S(f"s_add_u32 s64, %[nxt], {raw_off}")
            V(f"v_add_u32 v{waddr}, s64, v{LANE16}")
            L(("raw", which), f"ds_read_b128 v[{raw}:{raw + 3}], v{waddr}")
            if codec == "mx":
                S(f"s_add_u32 s65, %[nxt], {sc_off}")
                V(f"v_add_u32 v{scl}, s65, v{SRC16}")
                L(("scale", which), f"ds_read_u8 v{scl}, v{scl}")
            return it
        if codec == "mx":
            W(("scale", which))
            V(f"v_add_u32 v{scl}, -6, v{scl}")                 # scale = 2^(E - 133): bits (E - 6) << 23
            V(f"v_lshlrev_b32 v{scl}, 23, v{scl}")
            V(f"v_mul_f32 v{bias}, 0xc3000000, v{scl}")         # bias = -128 scale
        W(("raw", which))
        # value n = byte n & 3 of raw dword n >> 2; output dword n >> 1.  The pack of a pair is issued after the NEXT pair's first
        # conversion (trans -> VALU use needs one instruction in between; so no s_nop is spent on it)
        pending_pk = None
        for n in range(16):
            t = VAL + (n & 3)
            V(f"v_cvt_f32_ubyte{n & 3} v{t}, v{raw + (n >> 2)}")
            if pending_pk is not None:
                V(pending_pk)
                pending_pk = None
            if codec == "mx":
                V(f"v_fma_f32 v{t}, v{t}, v{scl}, v{bias}")
            else:
                V(f"v_mul_f32 v{t}, 0x3b800000, v{t}")         # / 256: revolutions
                V(f"v_sin_f32 v{t}, v{t}")
            if n & 1:
                pending_pk = f"v_cvt_pk_bf16_f32 v{OUT + (n >> 1)}, v{t - 1}, v{t}"
        V(f"v_subrev_u32 v{waddr}, {FRAG}, v{waddr}")           # (also the instruction between the last sin and its pack)
        V(pending_pk)
        L(("w0", which), f"ds_write_b128 v{waddr}, v[{OUT}:{OUT + 3}]")
        L(("w1", which), f"ds_write_b128 v{waddr}, v[{OUT + 4}:{OUT + 7}] offset:{FRAG}")
        return it

    def _build(self):
        # ---- addresses of this tile's operands: per-lane transposed-read offset (k-step 0 / 1) + slot + the wave's pair offsets
        self.e("s_add_u32 s60, %[cur], %[aoff]", "salu")
        self.e("s_add_u32 s61, %[cur], %[boff]", "salu")
        self.e("s_add_u32 s62, %[cur], %[xoff]", "salu")
        self.e(f"s_add_u32 s63, %[cur], {AUX_PAIR_OFF}", "salu")
        for ks in range(2):
            for name, sreg in (("B", "s61"), ("A", "s60"), ("XA", "s62"), ("XB", "s63")):
                self.e(f"v_add_u32 v{ADDR[name] + ks}, {sreg}, v{RDOFF + ks}")
            if ks == 0:
                for name in OPS[:5]:  # the first reads go out as soon as their addresses exist
                    self.read_operand(0, name)
        for name in OPS[5:]:
            self.read_operand(0, name)
        # ---- fillers: the two decodes, in order; the k-step 1 reads are placed explicitly
        fillers = (self.decode_items("r", "fetch") + self.decode_items("c", "fetch") + self.decode_items("r", "compute")
                   + self.decode_items("c", "compute"))
        fi = 0

        def fill(k):
            nonlocal fi
            for _ in range(k):
                if fi < len(fillers):
                    fillers[fi]()
                    fi += 1

        order = [(a, c) for c in range(2) for a in range(4)]
        k1_reads = list(OPS)
        for ks in range(2):
            for idx, (a, c) in enumerate(order):
                self.mfma(ks, a, c)
                if ks == 0 and k1_reads:          # one operand of k-step 1 per gap: all eight are under way before k-step 0 ends
                    self.read_operand(1, k1_reads.pop(0))
                fill(self.fill)
            self.mfma_aux(ks)
            fill(self.fill)
        while fi < len(fillers):
            fillers[fi]()
            fi += 1
        self.e("s_waitcnt lgkmcnt(0)", "wait")  # the decoded fragments are written (the caller's vmcnt wait and s_barrier follow)
        self.lgkm = []

    def inc_file(self):
        head = [f"// GENERATED by csrc/gen/wgrad_tile.py -- do not edit.  Column codec: {self.col_codec}; {self.n['mfma']} MFMAs, "
                f"{self.n['valu']} VALU, {self.n['lds']} LDS operations, {self.n['wait']} waits per tile."]
        return "\n".join(head + ['"' + t + '\\n"' for t in self.ins]) + "\n"


def clobber_file():
    regs = [r for r in range(144, N_VGPR) if r not in (RDOFF, RDOFF + 1, LANE16, SRC16)]
    return ("// GENERATED by csrc/gen/wgrad_tile.py: clobber list of the per-tile statement (the accumulators v[0:143] are operands)\n"
            + ", ".join(f'"v{r}"' for r in regs) + ', "s60", "s61", "s62", "s63", "s64", "s65", "memory", "scc"\n')


def main():
    out_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1:
        out_dir = sys.argv[1]
    for codec, tag in (("phase", "p"), ("mx", "m")):
        t = Tile(codec)
        with open(os.path.join(out_dir, f"wgrad8f_tile_{tag}.inc"), "w") as f:
            f.write(t.inc_file())
        print(codec, t.n, len(t.ins))
    with open(os.path.join(out_dir, "wgrad8f_tile_clobbers.inc"), "w") as f:
        f.write(clobber_file())


if __name__ == "__main__":
    sys.exit(main())
