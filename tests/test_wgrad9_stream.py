"""The generated slice loop of the 4-wave weight-gradient kernel (csrc/gen/wgrad9_loop.py -> csrc/wgrad9_loop_{p,m,px,mx}.inc): the committed
files are current, every iteration has the shape the kernel's header promises, and -- replaying the instruction text through an
independent model of the in-order LDS counter, across the loop's back edge -- no MFMA issues while a transposed read into one of its
operand registers is still in flight, and no decoded fragment is published before its LDS writes have been waited for; and a functional
model (numpy register files, LDS, workspaces; four waves as cooperating generators) executes the stream on synthetic 8-bit workspaces and
compares every accumulator with the float64 contraction of the decoded fragments.  (End-to-end numerics against the reference's gradients
are held on the GPU, tests/test_hip_backward.py; this file runs without one.)"""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "satnerf_amd", "csrc", "gen", "wgrad9_loop.py")
VARIANTS = [("phase", True, "p"), ("mx", True, "m"), ("phase", False, "px"), ("mx", False, "mx")]
NORAW = [("phase", True, "pn"), ("mx", True, "mn"), ("phase", False, "pxn"), ("mx", False, "mxn")]   # r06: the same four without the raw duty


def _gen():
    spec = importlib.util.spec_from_file_location("wgrad9_loop_gen", GEN)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("codec,main,tag", VARIANTS)
def test_generated_files_are_current(codec, main, tag):
    g = _gen()
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", f"wgrad9_loop_{tag}.inc")) as f:
        assert f.read() == g.Stream(codec, main=main).inc_file(), "re-run satnerf_amd/csrc/gen/wgrad9_loop.py"
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", "wgrad9_loop_clobbers.inc")) as f:
        assert f.read() == g.clobber_file()


@pytest.mark.parametrize("codec,main,tag", NORAW)
def test_noraw_files_are_current_and_one_duty_lighter(codec, main, tag):
    g = _gen()
    s = g.Stream(codec, main=main, raw=False)
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", f"wgrad9_loop_{tag}.inc")) as f:
        assert f.read() == s.inc_file(), "re-run satnerf_amd/csrc/gen/wgrad9_loop.py"
    _, bodies = _split(s.ins)
    for b in bodies:
        n = lambda pat: sum(bool(re.match(pat, t)) for t in b)  # noqa: E731
        assert n(r"v_mfma_f32_32x32x16_f16") == (36 if main else 4) and n(r"ds_write_b128") == 8 and n(r"global_load_dwordx4") == 4
        assert n(r"v_cvt_pk_f16_f32") == 0   # (the raw fragment's conversion is gone)


def _thin_variants():
    return list(_gen().THIN.items())


def test_thin_stream_files_are_current_and_numbered_as_the_kernel_numbers_them():
    """r06: the thin streams of one-row blocks.  packing.WG9_THIN (the ids of the per-wave variant table) lists them in the order
    csrc/wgrad9.hip dispatches on."""
    g = _gen()
    from satnerf_amd import packing

    assert list(packing.WG9_THIN) == list(g.THIN)
    with open(os.path.join(ROOT, "satnerf_amd", "csrc", "wgrad9.hip")) as f:
        src = f.read()
    for k, (thin, tag) in enumerate(g.THIN.items()):
        with open(os.path.join(ROOT, "satnerf_amd", "csrc", f"wgrad9_loop_{tag}.inc")) as f:
            assert f.read() == g.Stream("phase", thin=thin).inc_file(), "re-run satnerf_amd/csrc/gen/wgrad9_loop.py"
        assert re.search(rf'variant == {k + 1}\) {{\s+asm volatile\(\s+#include "wgrad9_loop_{tag}.inc"', src), (k + 1, tag)


@pytest.mark.parametrize("thin,tag", [(t, tag) for t, tag in [((1, True, "thin"), "t1"), ((3, False, "none"), "d3"), ((0, True, "thin"), "t0"),
                                                                ((2, True, "none"), "r2"), ((2, False, "none"), "d2"), ((0, False, "none"), "d0")]])
def test_thin_iteration_shape(thin, tag):
    g = _gen()
    assert g.THIN[thin] == tag
    n_df, raw, mfma = thin
    s = g.Stream("phase", thin=thin)
    prologue, bodies = _split(s.ins)
    assert not any(t == "s_barrier" for b in bodies for t in b) and sum(t == "s_barrier" for t in prologue) == 1
    for b in bodies:
        n = lambda pat: sum(bool(re.match(pat, t)) for t in b)  # noqa: E731
        assert n(r"v_mfma_f32_32x32x16_f16") == (10 if mfma == "thin" else 0)      # 2 k-steps x (row pair 0 x 4 column pairs + its aux tile)
        assert n(r"ds_read_b64_tr_b16") == (24 if mfma == "thin" else 0)           # A0, B0..B3, X: two reads each, two k-steps
        assert n(r"ds_write_b128") == 2 * n_df + (1 if raw else 0) and n(r"ds_add_u32") == 1
        assert n(r"global_load_dwordx4") == n_df + (1 if raw else 0) and n(r"global_load_ubyte") == 0
        assert n(r"v_sin_f16_sdwa") == 16 * n_df and n(r"v_pk_fma_f16") == 0


def _regs(tok):
    """registers of an operand like v[32:35], a[0:15], v246"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def _split(ins):
    loop, exit_ = ins.index("1:"), ins.index("9:")
    body = ins[loop + 1:exit_]
    ends = [i for i, t in enumerate(body) if t.startswith("s_cbranch_scc1 9f")]
    assert len(ends) == 4
    bodies, start = [], 0
    for e in ends:
        bodies.append(body[start:e + 1])
        start = e + 1
    return ins[:loop], bodies


@pytest.mark.parametrize("codec,main,tag", VARIANTS)
def test_iteration_shape(codec, main, tag):
    g = _gen()
    s = g.Stream(codec, main=main)
    prologue, bodies = _split(s.ins)
    assert not any(t == "s_barrier" for b in bodies for t in b), "no s_barrier inside the loop"
    assert sum(t == "s_barrier" for t in prologue) == 1
    for b in bodies:
        n = lambda pat: sum(bool(re.match(pat, t)) for t in b)  # noqa: E731
        assert n(r"v_mfma_f32_32x32x16_f16") == (36 if main else 4)
        assert n(r"ds_read_b64_tr_b16") == (36 if main else 12)
        assert n(r"ds_write_b128") == 9 and n(r"ds_add_u32") == 1
        assert n(r"global_load_dwordx4") == 5 and n(r"global_load_ubyte") == (2 if codec == "phase" else 4)
        assert n(r"v_sin_f16_sdwa") == (32 if codec == "phase" else 0)
        assert n(r"v_pk_fma_f16") == (16 if codec == "phase" else 32)


@pytest.mark.parametrize("codec,main,tag", VARIANTS + [(c, ("noraw", m), t) for c, m, t in NORAW] + [("phase", t, tag) for t, tag in (((1, True, "thin"), "t1"), ((3, False, "none"), "d3"),
                                                                                          ((0, True, "thin"), "t0"), ((2, True, "none"), "r2"),
                                                                                          ((2, False, "none"), "d2"), ((0, False, "none"), "d0"))])
def test_lds_counter_discipline_across_the_back_edge(codec, main, tag):
    g = _gen()
    if isinstance(main, bool):
        s = g.Stream(codec, main=main)
    elif main[0] == "noraw":
        s = g.Stream(codec, main=main[1], raw=False)
    else:
        s = g.Stream(codec, thin=main)
    prologue, bodies = _split(s.ins)
    inflight = []  # LDS operations in issue order: (kind, destination registers)
    exec_one = False

    def run(seq):
        nonlocal inflight, exec_one
        for t in seq:
            op, _, rest = t.partition(" ")
            args = [a.strip() for a in rest.split(",")] if rest else []
            if op.startswith(("v_", "ds_", "global_")) and op != "ds_add_u32":
                assert not exec_one, f"{t} issued with EXEC = 1"
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    inflight = inflight[len(inflight) - keep:] if keep else []
            elif op == "ds_read_b64_tr_b16" or op == "ds_read_b32":
                inflight.append(("read", _regs(args[0])))
                assert len(inflight) <= 15, "the LDS counter has four bits"
            elif op == "ds_write_b128":
                inflight.append(("write", set()))
                assert len(inflight) <= 15
            elif op == "s_mov_b64" and args[0] == "exec":
                exec_one = args[1] == "1"
            elif op == "ds_add_u32":
                # publish: every fragment write issued before it has been waited for; ONE lane adds (an LDS atomic is per lane)
                assert not any(k == "write" for k, _ in inflight), "a tile is published before its LDS writes have retired"
                assert exec_one, "ds_add_u32 with all 64 lanes active moves the counter by 64"
                inflight.append(("add", set()))
            elif op.startswith("v_mfma"):
                pending = set().union(*[r for k, r in inflight if k == "read"]) if inflight else set()
                for a in args[1:3]:
                    assert not (_regs(a) & pending), f"MFMA reads {a} while a transposed read into it is in flight: {t}"
            elif op == "v_readfirstlane_b32":
                pending = set().union(*[r for k, r in inflight if k == "read"]) if inflight else set()
                assert not (_regs(args[1]) & pending), t
            elif re.match(r"\d+:$", t) and int(t[:-1]) >= 100:
                # the poll loop of a consume check is entered with everything drained (model: only ever FEWER operations in flight)
                pass

    run(prologue)
    for trip in range(3):  # the second and third trips enter the first body over the back edge, with the fourth body's operations in flight
        for b in bodies:
            # (the slow path of a consume check drains the counter; the fast path is what is modelled: skip the poll loop's instructions)
            seq, skip = [], False
            for t in b:
                if re.match(r"1\d\d:$", t):
                    skip = True
                elif re.match(r"1\d:$", t) or re.match(r"\d\d:$", t) and not re.match(r"1\d\d:$", t):
                    skip = False
                    continue
                if not skip:
                    seq.append(t)
            run(seq)


# ---- functional model: the stream executed on numpy register files / LDS / workspaces -------------------------------------------------
# Four waves (generators that yield at the prologue's barrier and in the poll loop of a consume check) run one workgroup of wgrad9.hip over a
# short slice of synthetic 8-bit workspaces; the accumulators must equal the contraction of the decoded fragments computed in float64.
# What this holds without a GPU: duty addressing, the packed-fp16 decode (selectors, magic numbers, exponent arithmetic), the LDS image
# (write addresses, rotated lanes, transposed reads -- semantics of ds_read_b64_tr_b16 as printed by tools/probe_tr.hip), the operand ->
# accumulator mapping of every wave, the aux tiles, the publish / consume counters, the clamped loads and the loop's exits.  The wrapper
# arithmetic of wgrad9.hip (bases, offsets, scales) is restated here in Python.
import numpy as np  # noqa: E402

from satnerf_amd import packing  # noqa: E402

F16 = np.float16


def _f16(bits):
    return np.asarray(bits, np.uint16).view(F16).astype(np.float64)


def _to_f16_bits(x):
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float64).astype(F16).view(np.uint16).astype(np.uint32)


class _Wave:
    def __init__(self, ins, ops, lds, mem, shared):
        self.ins, self.ops, self.lds, self.mem, self.shared = ins, ops, lds, mem, shared
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.float32)
        self.s = {}
        self.scc, self.exec1 = 0, False
        self.labels = {}

    def sval(self, tok):  # scalar source: sNN, %[name], integer / hex literal
        tok = tok.strip()
        if tok.startswith("%["):
            return self.ops[tok[2:-1]]
        if re.fullmatch(r"s\d+", tok):
            return self.s.get(tok, 0)
        return int(tok, 0) & 0xffffffffffffffff if tok.startswith("0x") else int(tok)

    def src(self, tok):   # vector source: vNN or a scalar broadcast
        tok = tok.strip()
        if re.fullmatch(r"v\d+", tok):
            return self.v[int(tok[1:])]
        return np.full(64, self.sval(tok) & 0xffffffff, np.uint32)

    def run(self):
        ins, pc = self.ins, 0
        fwd = lambda name, i: next(k for k in range(i, len(ins)) if ins[k] == name + ":")            # noqa: E731
        bwd = lambda name, i: next(k for k in range(i, -1, -1) if ins[k] == name + ":")              # noqa: E731
        while pc < len(ins):
            t = ins[pc]
            pc += 1
            if t.endswith(":"):
                continue
            op, _, rest = t.partition(" ")
            # split operands at top-level commas (modifiers follow after a space)
            body, *mods = rest.split(" op_sel_hi") if " op_sel_hi" in rest else (rest.split(" dst_sel") if " dst_sel" in rest else [rest])
            args = [x.strip() for x in body.split(",")]
            if op in ("s_waitcnt", "s_nop"):
                continue
            if op == "s_barrier":
                self.shared["barrier"] += 1
                while self.shared["barrier"] % 4:
                    yield
                continue
            if op == "s_mov_b32":
                self.s[args[0]] = self.sval(args[1]) & 0xffffffff
            elif op == "s_mov_b64":
                self.exec1 = args[1] == "1"
            elif op in ("s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_ge_i32"):
                x, y = self.sval(args[0]) & 0xffffffff, self.sval(args[1]) & 0xffffffff
                if op == "s_cmp_ge_i32":
                    x, y = np.int32(np.uint32(x)), np.int32(np.uint32(y))
                self.scc = int({"s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y, "s_cmp_ge_i32": x >= y}[op])
            elif op == "s_cselect_b32":
                self.s[args[0]] = (self.sval(args[1]) if self.scc else self.sval(args[2])) & 0xffffffff
            elif op in ("s_sub_u32", "s_add_u32"):
                x, y = self.sval(args[1]) & 0xffffffff, self.sval(args[2]) & 0xffffffff
                r = x - y if op == "s_sub_u32" else x + y
                self.scc = int(r < 0 or r > 0xffffffff)
                if args[0].startswith("%["):
                    self.ops[args[0][2:-1]] = r & 0xffffffff
                else:
                    self.s[args[0]] = r & 0xffffffff
            elif op in ("s_cbranch_scc1", "s_cbranch_scc0", "s_branch"):
                take = op == "s_branch" or self.scc == (1 if op == "s_cbranch_scc1" else 0)
                if take:
                    name, direction = args[0][:-1], args[0][-1]
                    if direction == "b" and len(name) == 3:   # the poll loop of a consume check: let the other waves run
                        yield
                    pc = fwd(name, pc) if direction == "f" else bwd(name, pc)
            elif op == "v_accvgpr_write_b32":
                self.a[int(args[0][1:])] = 0.0
            elif op == "v_mov_b32":
                self.v[int(args[0][1:])] = self.src(args[1])
            elif op == "v_add_u32":
                self.v[int(args[0][1:])] = (self.src(args[1]).astype(np.uint64) + self.src(args[2])).astype(np.uint32)
            elif op == "v_subrev_u32":   # D = S1 - S0
                self.v[int(args[0][1:])] = (self.src(args[2]).astype(np.int64) - self.src(args[1]).astype(np.int64)).astype(np.uint32)
            elif op == "v_max_i32":
                self.v[int(args[0][1:])] = np.maximum(self.src(args[1]).view(np.int32), self.src(args[2]).view(np.int32)).view(np.uint32)
            elif op == "v_lshlrev_b32":  # D = S1 << S0
                self.v[int(args[0][1:])] = (self.src(args[2]).astype(np.uint64) << (self.sval(args[1]) & 31)).astype(np.uint32)
            elif op == "v_and_b32":
                self.v[int(args[0][1:])] = self.src(args[1]) & self.src(args[2])
            elif op == "v_mul_f16":
                r = _f16(self.src(args[1]) & 0xffff) * _f16(self.src(args[2]) & 0xffff)
                self.v[int(args[0][1:])] = _to_f16_bits(r)
            elif op == "v_perm_b32":
                s0, s1, sel = self.src(args[1]), self.src(args[2]), self.sval(args[3])
                pool = np.stack([(s1 >> (8 * k)) & 0xff for k in range(4)] + [(s0 >> (8 * k)) & 0xff for k in range(4)])
                out = np.zeros(64, np.uint32)
                for k in range(4):
                    b = (sel >> (8 * k)) & 0xff
                    assert b < 8, "only plain byte selectors are modelled"
                    out |= pool[b] << (8 * k)
                self.v[int(args[0][1:])] = out
            elif op == "v_pk_fma_f16":   # op_sel_hi:[1,0,0]: both halves use the LOW halves of the second and third operand
                s0, s1, s2 = self.src(args[1]), self.src(args[2]), self.src(args[3])
                lo = _f16(s0 & 0xffff) * _f16(s1 & 0xffff) + _f16(s2 & 0xffff)
                hi = _f16(s0 >> 16) * _f16(s1 & 0xffff) + _f16(s2 & 0xffff)
                self.v[int(args[0][1:])] = _to_f16_bits(lo) | (_to_f16_bits(hi) << 16)
            elif op == "v_sin_f16_sdwa":
                half = int(re.search(r"WORD_(\d)", t).group(1))
                d = int(args[0][1:])
                x = _f16((self.v[int(args[1][1:])] >> (16 * half)) & 0xffff)
                r = _to_f16_bits(np.sin(2 * np.pi * x))
                self.v[d] = (self.v[d] & (0xffff0000 if half == 0 else 0x0000ffff)) | (r << (16 * half))
            elif op == "v_mul_f32":
                self.v[int(args[0][1:])] = (self.src(args[1]).view(np.float32) * self.src(args[2]).view(np.float32)).astype(np.float32).view(np.uint32)
            elif op == "v_cvt_pk_f16_f32":
                lo, hi = self.src(args[1]).view(np.float32), self.src(args[2]).view(np.float32)
                self.v[int(args[0][1:])] = _to_f16_bits(lo) | (_to_f16_bits(hi) << 16)
            elif op == "v_readfirstlane_b32":
                self.s[args[0]] = int(self.v[int(args[1][1:])][0])
            elif op in ("global_load_dwordx4", "global_load_ubyte"):
                base = self.sval(args[2])
                addr = base + self.src(args[1]).astype(np.int64)
                if op == "global_load_ubyte":
                    self.v[int(args[0][1:])] = self.mem[addr].astype(np.uint32)
                else:
                    d = int(re.match(r"v\[(\d+):", args[0]).group(1))
                    for k in range(4):
                        w = np.zeros(64, np.uint32)
                        for b in range(4):
                            w |= self.mem[addr + 4 * k + b].astype(np.uint32) << (8 * b)
                        self.v[d + k] = w
            elif op == "ds_write_b128":
                m = re.search(r"offset:(\d+)", t)
                off = int(m.group(1)) if m else 0
                addr = self.src(args[0]).astype(np.int64) + off
                d = int(re.match(r"v\[(\d+):", args[1].split(" ")[0]).group(1))
                for k in range(4):
                    for b in range(4):
                        self.lds[addr + 4 * k + b] = ((self.v[d + k] >> (8 * b)) & 0xff).astype(np.uint8)
            elif op == "ds_read_b64_tr_b16":
                m = re.search(r"offset:(\d+)", t)
                off = int(m.group(1)) if m else 0
                vaddr = self.src(args[1].split(" ")[0]).astype(np.int64) + off
                d = int(re.match(r"v\[(\d+):", args[0]).group(1))
                chunk = np.zeros((64, 4), np.uint32)   # the four halfwords at every lane's address
                for j in range(4):
                    chunk[:, j] = self.lds[vaddr + 2 * j].astype(np.uint32) | (self.lds[vaddr + 2 * j + 1].astype(np.uint32) << 8)
                out = np.zeros((64, 4), np.uint32)
                for lane in range(64):
                    g0, i = lane & ~15, lane & 15
                    for j in range(4):      # tools/probe_tr.hip: lane i of a 16-lane group gets element i & 3 of lanes 4 j + (i >> 2)
                        out[lane, j] = chunk[g0 + 4 * j + (i >> 2), i & 3]
                self.v[d] = out[:, 0] | (out[:, 1] << 16)
                self.v[d + 1] = out[:, 2] | (out[:, 3] << 16)
            elif op == "ds_add_u32":
                assert self.exec1, "a 64-lane LDS atomic would move the counter by 64"
                m = re.search(r"offset:(\d+)", t)
                addr = int(self.v[int(args[0][1:])][0]) + (int(m.group(1)) if m else 0)
                cur = int.from_bytes(bytes(self.lds[addr:addr + 4]), "little") + int(self.v[int(args[1].split(" ")[0][1:])][0])
                self.lds[addr:addr + 4] = np.frombuffer(cur.to_bytes(4, "little"), np.uint8)
            elif op == "ds_read_b32":
                m = re.search(r"offset:(\d+)", t)
                addr = int(self.v[int(args[1].split(" ")[0][1:])][0]) + (int(m.group(1)) if m else 0)
                self.v[int(args[0][1:])] = int.from_bytes(bytes(self.lds[addr:addr + 4]), "little")
            elif op == "v_mfma_f32_32x32x16_f16":
                dst, A, B = args[0], int(re.match(r"v\[(\d+):", args[1]).group(1)), int(re.match(r"v\[(\d+):", args[2]).group(1))
                def operand(r0):   # (32, 16) matrix: row / column index = lane & 31, k = 8 (lane >> 5) + element
                    M = np.zeros((32, 16))
                    for q in range(4):
                        w = self.v[r0 + q]
                        for half in range(2):
                            vals = _f16((w >> (16 * half)) & 0xffff)
                            for lane in range(64):
                                M[lane & 31, 8 * (lane >> 5) + 2 * q + half] = vals[lane]
                    return M
                D = operand(A) @ operand(B).T   # D[row][col]
                d0 = int(re.match(r"[av]\[(\d+):", dst).group(1))
                for g in range(16):
                    for lane in range(64):
                        row, col = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5), lane & 31
                        if dst[0] == "a":
                            self.a[d0 + g, lane] += D[row, col]
                        else:
                            self.v[d0 + g, lane] = (self.v[d0 + g, lane:lane + 1].view(np.float32) + np.float32(D[row, col])).view(np.uint32)[0]
            else:
                raise AssertionError("instruction not modelled: " + t)


def _decoded(mem, base, unit_off, codec, scale_addr, g_bits):
    """a double fragment of 1 KiB -> (2, 16 slots, 32 points) in float64, as the stream decodes it (fp16 rounding of the scale product only)"""
    raw = mem[base + unit_off: base + unit_off + 1024].reshape(64, 16).astype(np.float64)   # [source lane][value n]
    if codec == "phase":
        val = np.sin(2 * np.pi * (raw / 256.0))
    else:
        e = mem[scale_addr + 16 * np.arange(64)].astype(np.int64)
        field = np.maximum(e - g_bits, 0)                      # fp16 exponent field of the lane's scale; 0 = flushed
        scale = np.where(field > 0, 2.0 ** (field - 15), 0.0)
        val = (raw - 128.0) * scale[:, None]
    out = np.zeros((2, 16, 32))
    for lane in range(64):
        p, h = lane & 31, lane >> 5
        for n in range(16):
            out[n >> 3, 8 * h + (n & 7), p] = val[lane, n]
    return out


@pytest.mark.parametrize("blk,nt,spread", [(2, 6, 0), (1, 5, 0), (8, 3, 0), (1, 4, 44), (9, 3, 30), (12, 5, 0), (13, 6, 0), (12, 2, 0)])
def test_stream_computes_the_block_contraction(blk, nt, spread):
    """``spread`` > 0: the exponent bytes of the block's SECOND row group lie that many binades below the first one's (two layers of very
    different gradient scale sharing a block): with one Emax per block those rows would flush to zero; the range is fitted
    per 32-row pair, and every accumulator row tile is checked against ITS OWN largest entry."""
    g = _gen()
    rng = np.random.default_rng(10 * blk + nt)
    loads = packing.wgrad8_loads(256, 4)
    bm = packing.backward_maps(256, 4)
    col_mx = bm["blocks"][blk, 8] == packing.KIND_BF16
    duties = loads[blk, 20:100].reshape(4, 5, 4)
    quad = int(loads[blk, 108])
    variants = [int(x) for x in loads[blk, 109:113]]   # r06: 0 = the full stream, k > 0 = thin stream packing.WG9_THIN[k - 1]
    assert (blk in packing.wgrad9_thin_blocks(256, 4)) == any(variants)
    def n_df_of(w):   # double-fragment duty slots wave w's stream executes
        return 4 if variants[w] == 0 else packing.WG9_THIN[variants[w] - 1][0]
    def raw_of(w):   # full streams: the wave runs the raw duty unless it is a dump (r06: the no-raw variants); thin streams: per variant
        return int(duties[w, 4, 2]) != packing.WG9_DUMP_FRAG if variants[w] == 0 else packing.WG9_THIN[variants[w] - 1][1]
    ak, dk = packing.act8_units(1, 256), packing.dpre8_units(256)
    n_tiles = nt + 2                                   # two tiles beyond the slice: what the clamp logic may touch
    D0, A0 = 4096, 4096 + n_tiles * dk * 1024 + 4096   # byte addresses of the two workspaces in `mem`
    mem = rng.integers(0, 256, A0 + n_tiles * ak * 1024 + 4096, dtype=np.uint8)
    # sane exponent bytes (every MX8 scale unit) and bf16 fragments (aux, d_sigma_pre, d_head)
    g8 = packing.fmt8_geometry(256)
    for t in range(n_tiles):
        for u in range(g8["D8_SCALE"], g8["D8_UNITS"]):
            mem[D0 + (t * dk + u) * 1024: D0 + (t * dk + u + 1) * 1024] = rng.integers(96, 112, 1024, dtype=np.uint8)
        if spread:   # the scale group of the block's last MX8 row pair: `spread` binades down (its 8 bytes of every lane's 16)
            pg = packing.wgrad9_pair_groups(256, 4)[blk]
            low = [int(x) for x in pg if 0 <= x < 14][-1]
            assert low != int(pg[0])
            u, b0 = g8["D8_SCALE"] + low // g8["GROUPS_PER_UNIT"], (low % g8["GROUPS_PER_UNIT"]) * g8["MT"]
            unit = mem[D0 + (t * dk + u) * 1024: D0 + (t * dk + u + 1) * 1024].reshape(64, 16)
            unit[:, b0:b0 + g8["MT"]] = rng.integers(96 - spread, 112 - spread, (64, g8["MT"]), dtype=np.uint8)
        u = 1 + g8["A8_SCALE"]
        mem[A0 + (t * ak + u) * 1024: A0 + (t * ak + u + 1) * 1024] = rng.integers(118, 130, 1024, dtype=np.uint8)
        def put_bf16(addr, lo, hi):
            vals = rng.uniform(lo, hi, 512).astype(np.float32) * rng.choice([-1.0, 1.0], 512).astype(np.float32)
            mem[addr:addr + 1024] = (vals.view(np.uint32) >> 16).astype(np.uint16).view(np.uint8)
        put_bf16(A0 + t * ak * 1024, 0.1, 1.0)
        put_bf16(D0 + (t * dk + g8["D8_SIGMA"]) * 1024, 2.0 ** -18, 2.0 ** -16)
        put_bf16(D0 + (t * dk + g8["D8_HEAD"]) * 1024, 2.0 ** -18, 2.0 ** -16)
    # ---- the wrapper's arithmetic (csrc/wgrad9.hip), restated --------------------------------------------------------------------
    ring, slot_b = 0, g.SLOT
    lds = np.zeros(4 * slot_b + 16, np.uint8)
    lds[:] = rng.integers(0, 256, lds.size, dtype=np.uint8)   # stale LDS content must not matter ...
    lds[4 * slot_b:] = 0                                       # ... except the publish counters, which the kernel zeroes
    # fp16 range per 32-row pair: the largest exponent of the pair's source over the slice (what the dX kernel's table of exponent maxima
    # delivers per scale group; here taken from the bytes themselves), columns likewise when they are MX8
    pair_e, ec = {}, 0
    for w in range(4):
        for k in range(n_df_of(w)):
            src, unit, dst, sc = (int(x) for x in duties[w, k])
            if dst == packing.WG9_DUMP_FRAG or ((k >= 2 or variants[w]) and not col_mx):
                continue
            base = D0 if src == 1 else A0
            stride = (dk if src == 1 else ak) * 1024
            e = max(int(mem[base + t * stride + (sc >> 4) * 1024 + (sc & 15) + 16 * np.arange(64)].max()) for t in range(nt))
            if k < 2:
                assert dst % 2 == 0
                pair_e[dst // 2] = max(pair_e.get(dst // 2, 0), e)
            else:
                ec = max(ec, e)
    src0, unit0, dst0, _ = (int(x) for x in duties[0, 4])
    if src0 == 1 and dst0 != packing.WG9_DUMP_FRAG:   # a bf16 row fragment, alone in its pair: its largest exponent
        assert dst0 % 2 == 0 and dst0 // 2 not in pair_e
        e = 0
        for t in range(nt):
            h = mem[D0 + (t * dk + unit0) * 1024: D0 + (t * dk + unit0 + 1) * 1024].view(np.uint16)
            e = max(e, int((h & 0x7fff).max()) >> 7)
        pair_e[dst0 // 2] = e
    clamp = lambda e: min(max(e, 32), 254)  # noqa: E731
    pair_e = {p: clamp(e) for p, e in pair_e.items()}
    ec = clamp(ec)
    er_of = lambda frag: pair_e.get(frag // 2, 32)  # noqa: E731  (a dump duty: any reference)
    g_row_of = lambda frag: 2.0 ** (138 - er_of(frag))  # noqa: E731
    g_col = 2.0 ** (138 - ec) if col_mx else 1.0
    shared = {"barrier": 0}
    s = g.Stream("mx" if col_mx else "phase", main=True)
    sx = g.Stream("mx" if col_mx else "phase", main=False)
    waves = []
    for w in range(4):
        wr, wc = w >> 1, w & 1
        ops = {"nt": nt, "tleft": n_tiles - 1, "erow0": er_of(int(duties[w, 0, 2])) - 20, "erow1": er_of(int(duties[w, 1, 2])) - 20,
               "ecol": ec - 20, "flags": ring + 4 * slot_b,
               "strd": dk * 1024, "stra": ak * 1024, "aofl": 0 if variants[w] else (4 * wr + 2 * wc) * g.PAIR,
               "aofh": (4 * wr + ((2 * wc + 2) & 3)) * g.PAIR, "bof": (8 + 4 * wc) * g.PAIR}
        raw_src = int(duties[w, 4, 0])
        ops["strx"] = ops["strd"] if raw_src == 1 else ops["stra"]
        ops["sraw"] = int(np.float32(g_row_of(int(duties[w, 4, 2])) if raw_src == 1 else 1.0).view(np.uint32))
        for k in range(5):
            src, unit, dst, sc = (int(x) for x in duties[w, k])
            base = D0 if src == 1 else A0
            ops["bx" if k == 4 else f"b{k}"] = base + unit * 1024
            if k < 4:
                ops[f"sb{k}"] = base + (sc >> 4) * 1024 + (sc & 15)
            ops["wx" if k == 4 else f"w{k}"] = ring + dst * g.FRAG
        if variants[w]:
            stream = g.Stream("phase", thin=packing.WG9_THIN[variants[w] - 1])
        else:   # wgrad9.hip's dispatch: column codec, quadrant on / off, raw duty or not
            stream = g.Stream("mx" if col_mx else "phase", main=bool((quad >> w) & 1), raw=raw_of(w))
        wave = _Wave(stream.ins, ops, lds, mem, shared)
        lane = np.arange(64)
        src_unit = np.where(lane < 32, lane, 32 + ((lane - 8) & 31))
        hh, rh, m, q = lane >> 5, (lane >> 4) & 1, (lane >> 2) & 3, lane & 3
        for ks in range(2):
            point = 16 * ks + 8 * hh + m
            wave.v[g.IN_RD0 + ks] = ring + rh * g.FRAG + np.where(q >> 1, 512 + ((point + 8) & 31) * 16, point * 16) + (q & 1) * 8
        wave.v[g.IN_LANE16] = lane * 16
        wave.v[g.IN_VD] = wave.v[g.IN_VA] = wave.v[g.IN_VX] = src_unit * 16
        waves.append(wave)
    runs = [w.run() for w in waves]
    live = list(range(4))
    for _ in range(100000):
        for i in list(live):
            try:
                next(runs[i])
            except StopIteration:
                live.remove(i)
        if not live:
            break
    assert not live, "a wave never finished: the publish / consume protocol dead-locked"
    # ---- reference: the contraction of the decoded fragments, float64 ----------------------------------------------------------------
    rows, cols = np.zeros((256, 32 * nt)), np.zeros((256, 32 * nt))
    aux = np.zeros((32, 32 * nt))
    for t in range(nt):
        sl = slice(32 * t, 32 * t + 32)
        for w in range(4):
            for k in list(range(n_df_of(w))) + ([4] if raw_of(w) else []):
                src, unit, dst, sc = (int(x) for x in duties[w, k])
                if dst == packing.WG9_DUMP_FRAG:
                    continue
                base = (D0 if src == 1 else A0) + t * (dk if src == 1 else ak) * 1024
                if k == 4:   # raw bf16 fragment: 8 values per lane = slots 8 h + j of point p
                    halves = mem[base + unit * 1024: base + (unit + 1) * 1024].view(np.uint16).astype(np.uint32) << 16
                    vals = halves.view(np.float32).astype(np.float64).reshape(64, 8) * (g_row_of(dst) if src == 1 else 1.0)
                    frag = np.zeros((16, 32))
                    for lane in range(64):
                        frag[8 * (lane >> 5): 8 * (lane >> 5) + 8, lane & 31] = vals[lane]
                    frag = _f16(_to_f16_bits(frag))
                    if dst >= 32:
                        aux[16 * (dst - 32): 16 * (dst - 32) + 16, sl] = frag
                    else:
                        rows[16 * dst: 16 * dst + 16, sl] = frag
                    continue
                is_row = k < 2 and not variants[w]   # (thin streams: every double-fragment duty is a column duty)
                codec = "mx" if is_row or col_mx else "phase"
                dec = _decoded(mem, base, unit * 1024, codec, base + (sc >> 4) * 1024 + (sc & 15), (er_of(dst) if is_row else ec) - 20)
                dec = _f16(_to_f16_bits(dec))
                tgt, f0 = (rows, dst) if dst < 16 else (cols, dst - 16)
                tgt[16 * f0: 16 * f0 + 16, sl], tgt[16 * f0 + 16: 16 * f0 + 32, sl] = dec[0], dec[1]
    want, want_aux = rows @ cols.T, rows @ aux.T
    scale = np.abs(want).max()
    n_rows, n_cols = 16 * len(bm["block_rows"][blk]), 16 * len(bm["block_cols"][blk])   # beyond them: stale LDS, results nobody reads
    for w, wave in enumerate(waves):
        wr, wc = w >> 1, w & 1
        thin_mfma = variants[w] and packing.WG9_THIN[variants[w] - 1][2] == "thin"
        for a in range(4):
            row0 = 32 * (4 * wr + ((a + 2 * wc) & 3))
            if variants[w]:   # thin streams contract row pair 0 in operand slot 0 (wgrad9.hip's epilogue reads accumulator row tile 0 only)
                if a != 0 or not thin_mfma:
                    continue
                row0 = 0
            got_rows = np.zeros((32, 128))
            for c in range(4):
                for gg in range(16):
                    for lane in range(64):
                        got_rows[(gg & 3) + 8 * (gg >> 2) + 4 * (lane >> 5), 32 * c + (lane & 31)] = wave.a[16 * (4 * a + c) + gg, lane]
            nr_, nc_ = max(min(n_rows - row0, 32), 0), max(min(n_cols - 128 * wc, 128), 0)
            if (quad >> w) & 1 and nr_ and nc_:
                ref = want[row0: row0 + nr_, 128 * wc: 128 * wc + nc_]
                err = np.abs(got_rows[:nr_, :nc_] - ref).max()
                assert err < 2e-3 * np.abs(ref).max(), (w, a, err, np.abs(ref).max(), scale)   # relative to the TILE's own largest entry
            if a < 2 and nr_:   # the aux tiles of operand slots 0, 1
                got_aux = np.zeros((32, 32))
                for gg in range(16):
                    for lane in range(64):
                        got_aux[(gg & 3) + 8 * (gg >> 2) + 4 * (lane >> 5), lane & 31] = wave.v[16 * a + gg, lane: lane + 1].view(np.float32)[0]
                n_real = 16  # aux fragment 0 only (tau = 4): slots 0..15
                err = np.abs(got_aux[:nr_, :n_real] - want_aux[row0: row0 + nr_, :n_real]).max()
                assert err < 2e-3 * max(np.abs(want_aux).max(), 1e-30), (w, a, "aux", err)
