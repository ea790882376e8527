import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_flow_amd import gu_ffjord, _native
from gan_flow_amd.gu import make_gu
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "F0"
nwarm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
data = make_gu(8, device="cpu", seed=1, extend_dim=True)
xs = [data.get_sample(2048).to(dev) for _ in range(12)]
torch.manual_seed(1)
model = gu_ffjord.build_model(2).to(dev).train()
def report(step):
    torch.cuda.synchronize()
    w = step.stats.cpu()
    print(mode, "fwd", _native.Stats(w[0, 0]), "bwd", _native.Stats(w[0, 1]))
    print("params finite", bool(torch.isfinite(step.flats[0]).all()), "adam state", step.adam_state[0].tolist())
if mode == "F0":
    try:
        step = gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=True, warmup_batches=xs[:nwarm])
        print("constructed fine", step.check())
    except Exception as e:
        print("FAILED", str(e)[:300])
else:
    step = gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=False)
    if mode == "F1":
        torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for x in xs[:nwarm]:
            step.static_x.copy_(x)
            step._iteration()
            if mode == "F3":
                torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    report(step)
