"""Time the 8-bit weight-gradient kernel (wgrad8) alone for one library build (SATRENDER_LIB); prints one line.  Inputs are random bit
patterns of the training workspace sizes (1024 rays x 64 samples, tau 4): timing does not depend on the values."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, packing, _lib
dev = 'cuda:0'
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
maps = packing.backward_maps(256, 4)
blocks = torch.from_numpy(maps["blocks"]).to(dev).contiguous()
if os.environ.get("AB_BLOCKS"):  # e.g. AB_BLOCKS=14 or 1,2,3: time a sub-table
    sel = [int(b) for b in os.environ["AB_BLOCKS"].split(",")]; blocks = blocks[sel].contiguous()
tiles = (n_points + 31) // 32
act_e = _lib.lib().sr_act_elems_per_tile(256, 8); dp_e = _lib.lib().sr_dpre_elems_per_tile(256, 8)
loads = torch.from_numpy(packing.wgrad8_loads(256, 4)).to(dev).contiguous()
acts = torch.randint(0, 30000, (tiles * act_e,), dtype=torch.int16, device=dev)
# the whole dpre workspace INCLUDING the table of exponent maxima the 4-wave kernel reads behind the last tile (sr_dpre_workspace_elems;
# tiles * sr_dpre_elems_per_tile is too short and the scales were read out of bounds); the table gets plausible exponent bytes
ws_tiles = _lib.lib().sr_workspace_tiles(n_points)
dpre = torch.randint(0, 30000, (_lib.lib().sr_dpre_workspace_elems(n_points, 256, 8),), dtype=torch.int16, device=dev)
dpre[ws_tiles * dp_e:].view(torch.uint8).fill_(120)
n_wgs = [int(a) for a in sys.argv[2:]] or [0]
dbg = None
if os.environ.get("AB_TIMING"):  # kernel built with -DSR_W8_TIMING: per-wave s_memtime stamps of workgroup 0
    dbg = torch.zeros(16 * 32 * 8, dtype=torch.int64, device=dev)
    os.environ["SR_W8_DBG"] = str(dbg.data_ptr())
dbg9 = None
if os.environ.get("AB_TIMING9"):  # wgrad9.hip built with -DSR_W9_TIMING: per workgroup (shader cycles, 100-MHz ticks, tiles)
    dbg9 = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
    os.environ["SR_W9_DBG"] = str(dbg9.data_ptr())
def make(n_wg):
    plan, n_slices, span = ops.wgrad_plan(blocks, n_points, n_wg)
    if os.environ.get("AB_SPLITS"):  # override: equal slices per block
        k = int(os.environ["AB_SPLITS"]); plan = plan.clone(); plan[:, 9] = k; plan[:, 10] = torch.arange(plan.shape[0], device=plan.device, dtype=plan.dtype) * k; n_slices = k * plan.shape[0]
    partial = torch.empty(n_slices * (256 * 256 + 256 * 32), dtype=torch.float32, device=dev)
    ld = loads[sel].contiguous() if os.environ.get("AB_BLOCKS") else loads
    return n_slices, lambda: _lib.call("sr_satnerf_wgrad8", 256, 4, n_points, dpre.data_ptr(), dpre.numel(), acts.data_ptr(), plan.data_ptr(), ld.data_ptr(), plan.shape[0],
                                       n_slices, span, partial.data_ptr(), torch.cuda.current_stream().cuda_stream)
for n_wg in n_wgs:
    n_slices, run = make(n_wg)
    for _ in range(5): run()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{os.path.basename(os.environ.get('SATRENDER_LIB', 'default')):28s} points={n_points} n_wg={n_wg} slices={n_slices} {best*1e3:8.1f} us")

if dbg is not None:
    torch.cuda.synchronize()
    d = dbg.cpu().view(16, 32, 8)
    names = ["start", "issued", "tile", "vmwait", "barrier"]
    for w in (0, 1, 3, 4, 5, 7):
        print(f"wave {w}: per-iteration deltas (shader cycles) iterations 8..15")
        for i in range(8, 16):
            row = d[w, i]
            deltas = [int(row[k] - row[k - 1]) for k in range(1, 5)]
            nxt = int(d[w, i + 1, 0] - row[0])
            print(f"  it {i:2d}: " + " ".join(f"{n}={v:5d}" for n, v in zip(names[1:], deltas)) + f"  | iter={nxt}")

if dbg9 is not None:
    torch.cuda.synchronize()
    d = dbg9.cpu()[:3 * 1024].view(-1, 3)[:n_slices].double()
    cyc, ticks, nt = d[:, 0], d[:, 1], d[:, 2]
    print(f"  wgrad9 stamps over {n_slices} workgroups: cycles/tile median {float((cyc / nt).median()):.0f} (min {float((cyc / nt).min()):.0f}, max {float((cyc / nt).max()):.0f}), "
          f"tiles {int(nt.min())}..{int(nt.max())}, slice loop {float(ticks.median()) / 100:.1f} us median / {float(ticks.max()) / 100:.1f} us max, "
          f"clock {float((cyc / ticks).median()) * 0.1:.2f} GHz")
