import sys, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle
from _helpers import oracle_fixture_model
from gan_flow_amd import _native as nat
DEV='cuda:0'
m = oracle_fixture_model("multimodal")
g = torch.Generator().manual_seed(9)
z0 = torch.randn(16384, 1, generator=g).clamp(-3.5, 3.5)
net = nat.net_desc(1, 64, 3, False)
params = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).float().to(DEV)
T = 0.5
qs = torch.tensor([0.5, 0.9, 0.99, 0.999])
def pct(v): return [float(x) for x in torch.quantile(v.double().flatten().cpu(), qs.double())] + [float(v.max())]
for tile in (0, 16):
    o = nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=tile)
    x, l, s1 = nat.solve_fwd(net, params, z0.to(DEV), None, T, 0.0, o)
    zb, lb, s2 = nat.solve_fwd(net, params, x, l, 0.0, T, o)
    torch.cuda.synchronize()
    print("hip tile=%d dz pct" % tile, ["%.2e" % v for v in pct((zb.cpu()-z0).abs())], "dlogp", ["%.2e" % v for v in pct(lb.abs())], flush=True)
# oracle per tile on the first 8 tiles, and batch-global on the first 256 rows
t0=time.time()
dz, dl = [], []
with torch.no_grad():
    for lo in range(0, 256, 16):
        zt = z0[lo:lo+16]
        xo, lo_ = m(zt.clone(), torch.zeros(16,1), reverse=True)
        zb, lb = m(xo, lo_)
        dz.append((zb-zt).abs()); dl.append(lb.abs())
print("oracle per-tile 256 rows dz", ["%.2e" % v for v in pct(torch.cat(dz))], "dlogp", ["%.2e" % v for v in pct(torch.cat(dl))], "%.0fs" % (time.time()-t0), flush=True)
t0=time.time()
with torch.no_grad():
    zt = z0[:1024]
    xo, lo_ = m(zt.clone(), torch.zeros(1024,1), reverse=True)
    zb, lb = m(xo, lo_)
print("oracle batch-global 1024 rows dz", ["%.2e" % v for v in pct((zb-zt).abs())], "dlogp", ["%.2e" % v for v in pct(lb.abs())], "%.0fs" % (time.time()-t0), flush=True)
