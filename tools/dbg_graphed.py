"""Reproducer: GraphedTrainStep replays, one eager evaluation (W-distance) in between, more replays --
prints the captured iteration's status words after each group of replays.  usage: dbg_graphed.py rk4|dopri5 [keep]"""
import sys, torch
sys.path.insert(0, '.')
from gan_flow_amd import _native, gu_ffjord
from gan_flow_amd.gu import make_gu
dev = torch.device("cuda:0")
data = make_gu(8, device="cpu", seed=1, extend_dim=True)
torch.manual_seed(1)
model = gu_ffjord.build_model(2, solver=sys.argv[1]).to(dev).train()
for b in model.chain:
    if sys.argv[1] != "dopri5": b.solver_options["step_size"] = 0.05
    b.spread_adjoint = True
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
x = data.get_sample(2048).to(dev)
g = gu_ffjord.GraphedTrainStep(model, opt, x, warmup_batches=[x])
keep = dict(model.chain[0].last_stats) if len(sys.argv) > 2 else None
for it in range(3):
    for _ in range(20):
        loss = g(data.get_sample(2048).to(dev))
    torch.cuda.synchronize()
    for k in ("forward", "backward"):
        w = model.chain[0].last_stats[k]
        print(it, k, w.tolist() if torch.is_tensor(w) else w, w.data_ptr() if torch.is_tensor(w) else None)
    print("loss", loss.item())
    if keep:
        print("   captured blocks now:", {k: (v.tolist(), v.data_ptr()) for k, v in keep.items()})
    if it == 0:
        wd = gu_ffjord.evaluate_w_distance(model, data, 2000, device=dev)
        print("wd", wd, {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in model.chain[0].last_stats.items()})
