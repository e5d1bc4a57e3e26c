"""Time the weight-gradient kernel alone for one library build (SATRENDER_LIB); prints one line.  Inputs are random bit
patterns of the training workspace sizes (1024 rays x 64 samples, tau 4): timing does not depend on the values."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, packing, _lib
dev = 'cuda:0'
n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
maps = packing.backward_maps(256, 4)
blocks = torch.from_numpy(maps["blocks"]).to(dev).contiguous()
if os.environ.get("AB_BLOCKS"):  # e.g. AB_BLOCKS=14 or 1,2,3: time a sub-table
    blocks = blocks[[int(b) for b in os.environ["AB_BLOCKS"].split(",")]].contiguous()
tiles = (n_points + 31) // 32
act_e = _lib.lib().sr_act_elems_per_tile(256, 16); dp_e = _lib.lib().sr_dpre_elems_per_tile(256, 16)
acts = torch.randint(0, 30000, (tiles * act_e,), dtype=torch.int16, device=dev)
dpre = torch.randint(0, 30000, (_lib.lib().sr_dpre_workspace_elems(n_points, 256, 16),), dtype=torch.int16, device=dev)  # (the 16-bit kernel reads no table behind it)
n_wgs = [int(a) for a in sys.argv[2:]] or [0]
def make(n_wg):
    plan, n_slices, span = ops.wgrad_plan(blocks, n_points, n_wg)
    if os.environ.get("AB_SPLITS"):  # override: equal slices per block
        k = int(os.environ["AB_SPLITS"]); plan = plan.clone(); plan[:, 9] = k; plan[:, 10] = torch.arange(plan.shape[0], device=plan.device, dtype=plan.dtype) * k; n_slices = k * plan.shape[0]
    partial = torch.empty(n_slices * (256 * 256 + 256 * 32), dtype=torch.float32, device=dev)
    return n_slices, lambda: _lib.call("sr_satnerf_wgrad", 256, 4, n_points, dpre.data_ptr(), acts.data_ptr(), plan.data_ptr(), plan.shape[0], n_slices,
                                       partial.data_ptr(), torch.cuda.current_stream().cuda_stream)
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
