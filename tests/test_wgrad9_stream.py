"""The generated slice loop of the 4-wave weight-gradient kernel (csrc/gen/wgrad9_loop.py -> csrc/wgrad9_loop_{p,m,px,mx}.inc): the committed
files are current, every iteration has the shape the kernel's header promises, and -- replaying the instruction text through an
independent model of the in-order LDS counter, across the loop's back edge -- no MFMA issues while a transposed read into one of its
operand registers is still in flight, and no decoded fragment is published before its LDS writes have been waited for.  (Numerics are
held by the gradient goldens on the GPU, tests/test_hip_backward.py; this runs without one.)"""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "satnerf_amd", "csrc", "gen", "wgrad9_loop.py")
VARIANTS = [("phase", True, "p"), ("mx", True, "m"), ("phase", False, "px"), ("mx", False, "mx")]


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


@pytest.mark.parametrize("codec,main,tag", VARIANTS)
def test_lds_counter_discipline_across_the_back_edge(codec, main, tag):
    g = _gen()
    s = g.Stream(codec, main=main)
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
