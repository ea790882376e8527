import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle
from _helpers import oracle_fixture_model
from gan_flow_amd import _native as nat
DEV='cuda:0'
m = oracle_fixture_model("multimodal")
flat = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).float().to(DEV)
g = torch.Generator().manual_seed(9)
z0 = torch.randn(16384, 1, generator=g).clamp(-3.5, 3.5)
net = nat.net_desc(1, 64, 3, False)
T = float(m.chain[0].sqrt_end_time.detach() ** 2)
o = nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=16)
x, l, s1 = nat.solve_fwd(net, flat, z0.to(DEV), None, T, 0.0, o)
zb, lb, s2 = nat.solve_fwd(net, flat, x, l, 0.0, T, o)
torch.cuda.synchronize()
i = int(lb.abs().argmax()); t0 = (i // 16) * 16
print("worst row", i, "z0", z0[i].item(), "x", x[i].item(), "l", l[i].item(), "zb", zb[i].item(), "lb", lb[i].item())
zt = z0[t0:t0+16]
m64 = oracle_fixture_model("multimodal", torch.float64)
with torch.no_grad():
    xs64, ls64 = oracle.odeint(m64.chain[0].odefunc, (zt.double(), torch.zeros(16,1,dtype=torch.float64)), torch.tensor([T,0.0],dtype=torch.float64), rtol=1e-10, atol=1e-10, method='dopri5')
    st={}
    xs32, ls32 = oracle.odeint(m.chain[0].odefunc, (zt.clone(), torch.zeros(16,1)), torch.tensor([T,0.0]), rtol=[1e-5]*2, atol=[1e-5]*2, method='dopri5', options=dict(stats=st))
print("tile fwd: hip-vs-f64 x %.3e l %.3e | oracle32-vs-f64 x %.3e l %.3e | oracle steps %d+%d" % ((x[t0:t0+16].cpu().double()-xs64[1]).abs().max(), (l[t0:t0+16].cpu().double()-ls64[1]).abs().max(), (xs32[1].double()-xs64[1]).abs().max(), (ls32[1].double()-ls64[1]).abs().max(), st['n_accepted'], st['n_rejected']))
# return leg on the f64 x
xt = x[t0:t0+16].cpu(); lt = l[t0:t0+16].cpu()
with torch.no_grad():
    zr64, lr64 = oracle.odeint(m64.chain[0].odefunc, (xt.double(), lt.double()), torch.tensor([0.0,T],dtype=torch.float64), rtol=1e-10, atol=1e-10, method='dopri5')
    zr32, lr32 = oracle.odeint(m.chain[0].odefunc, (xt.clone(), lt.clone()), torch.tensor([0.0,T]), rtol=[1e-5]*2, atol=[1e-5]*2, method='dopri5')
print("tile back: hip-vs-f64 z %.3e l %.3e | oracle32-vs-f64 z %.3e l %.3e" % ((zb[t0:t0+16].cpu().double()-zr64[1]).abs().max(), (lb[t0:t0+16].cpu().double()-lr64[1]).abs().max(), (zr32[1].double()-zr64[1]).abs().max(), (lr32[1].double()-lr64[1]).abs().max()))
print("f64 round trip logp residual", lr64[1].abs().max().item())
print("---- same 16 rows, back leg, three ways")
st={}
with torch.no_grad():
    zr32, lr32 = oracle.odeint(m.chain[0].odefunc, (xt.clone(), lt.clone()), torch.tensor([0.0,T]), rtol=[1e-5]*2, atol=[1e-5]*2, method='dopri5', options=dict(stats=st))
print("oracle steps %d+%d" % (st['n_accepted'], st['n_rejected']), "dts head", ["%.4g"%v for v in st['dts'][:8]])
for tile in (16, 0):
    oo = nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=tile)
    zh, lh, sh = nat.solve_fwd(net, flat, xt.to(DEV), lt.to(DEV), 0.0, T, oo)
    torch.cuda.synchronize()
    sh = nat.Stats(sh)
    print("hip tile=%d" % tile, sh, "vs f64: z %.3e l %.3e ; vs oracle32 z %.3e l %.3e" % ((zh.cpu().double()-zr64[1]).abs().max(), (lh.cpu().double()-lr64[1]).abs().max(), (zh.cpu()-zr32[1]).abs().max(), (lh.cpu()-lr32[1]).abs().max()))
