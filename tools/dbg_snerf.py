import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import satnerf_oracle as O
from satnerf_amd import ops
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
DEV = torch.device("cuda:0")
orig = ops.adam_step_graph
def spy(params, grads, m, v, state, **kw):
    if not torch.cuda.is_current_stream_capturing():
        print("   adam: grads max", float(grads.abs().max()), "state", state.tolist(), "params ptr", params.data_ptr(), "n", params.numel())
    return orig(params, grads, m, v, state, **kw)
ops.adam_step_graph = spy
args = O.default_args(model="s-nerf", sc_lambda=0.0, mlp_mode="bf16x3")
m = load_model(args)
m.load_state_dict(O.procedural_snerf_params(256, seed=3))
models = {"coarse": m.to(DEV)}
n = 128
rays, ts = O.synthetic_rays(n, seed=41)
rays, ts = rays.to(DEV), ts.to(DEV)
target = (torch.rand(n, 3, generator=torch.Generator().manual_seed(42)) * 0.3 + 0.3).to(DEV)
trg = Trainer(models, args)
print("flat ptr", trg.state.params.data_ptr(), trg.state.params.numel(), "model flat", m._flat.data_ptr(), m._flat.numel(), "grads", trg.state.grads.data_ptr(), m.flat_grads().data_ptr())
trg.step(rays, ts, target)
print("after step: model flat ptr", m._flat.data_ptr(), "flat_grads ptr", m._flat_grad.data_ptr(), "state grads ptr", trg.state.grads.data_ptr())
# eager comparison
tr2 = Trainer(models, args, use_graph=False)
parts = tr2._forward_backward(rays, tr2._zero_ts(ts), target)
print("eager grads max", float(tr2.state.grads.abs().max()), "ptrs", tr2.state.grads.data_ptr(), m._flat_grad.data_ptr())
