"""cnf_solve_fwd_static (include/cnf_hip.h): the forward solve on the library's own static device scratch -- no workspace
argument, nothing allocated by the call -- against cnf_solve_fwd, through the C ABI (ctypes, raw pointers).

Bar: bit-identical z, logp and stats words (same kernels, same arithmetic).  Every case makes TWO calls in a row on the
shared scratch, the second with a smaller input, so anything the first call leaves behind (granules of a 128-workgroup
grid, epochs, the H = 256 weight pieces of another model) would show.  Collected last (file name) so that a failure here
hides no other test under `pytest -x`."""
import ctypes

import pytest
import torch

from _helpers import oracle_fixture_model, oracle_random_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _flat(m):
    return torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).float()


def _raw_static(nat, lib, net, params, x, opts, t0, t1, z, lp, st):
    """The raw call: plain pointers and sizes, outputs preallocated by the caller."""
    rc = lib.cnf_solve_fwd_static(ctypes.byref(net), ctypes.c_void_p(params.data_ptr()), ctypes.c_void_p(x.data_ptr()), None, None,
                                  x.shape[0], t0, t1, ctypes.byref(opts), ctypes.c_void_p(z.data_ptr()),
                                  ctypes.c_void_p(lp.data_ptr()), ctypes.c_void_p(st.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.cnf_last_error().decode()


@pytest.mark.parametrize("case", ["h64_global", "h64_tile_local", "h64_stiff_reverse", "h256_global", "h64_rk4"])
def test_static_scratch_solve_matches_workspace_solve(case):
    from gan_flow_amd import _native as nat
    lib = nat.load_library()
    assert lib.cnf_static_scratch_bytes() == 16384 + 64 + 786432
    g = torch.Generator().manual_seed(77)
    hidden = 256 if case.startswith("h256") else 64
    if case == "h64_stiff_reverse":
        m, dim = oracle_fixture_model("multimodal"), 1
    else:
        dim = 2
        m = oracle_random_model(dim, seed=3, scale=1.5, hidden_dims=(hidden,) * 3)
    net = nat.net_desc(dim, hidden, 3, False)
    params = _flat(m).to(DEV)
    tile = 16 if case == "h64_tile_local" else 0
    opts = nat.make_opts("rk4", step_size=0.05) if case == "h64_rk4" else nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=tile)
    t0, t1 = (0.5, 0.0) if case == "h64_stiff_reverse" else (0.0, 0.5)
    sizes = {"h64_global": (2048, 100), "h64_tile_local": (5000, 333), "h64_stiff_reverse": (1024, 48), "h256_global": (512, 40),
             "h64_rk4": (3000, 17)}[case]
    # the first call on a device makes the HIP runtime load the library's code object (which is when the static array
    # itself comes to exist on that device): done here, so that the calls below can be held to "took no device memory"
    xw = torch.randn(16, dim, generator=g).to(DEV)
    nat.solve_fwd_static(net, params, xw, None, t0, t1, opts)
    torch.cuda.synchronize()
    for rows in sizes:                      # two calls in a row on the shared scratch, the second smaller
        x = torch.randn(rows, dim, generator=g).to(DEV)
        z_ref, lp_ref, st_ref = nat.solve_fwd(net, params, x, None, t0, t1, opts)
        z = torch.full_like(x, float("nan"))
        lp = torch.full((rows, 1), float("nan"), device=DEV)
        st = torch.full((8,), -1, dtype=torch.int32, device=DEV)
        free_before = torch.cuda.mem_get_info()[0]
        _raw_static(nat, lib, net, params, x, opts, t0, t1, z, lp, st)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before            # the call took no device memory
        s = nat.Stats(st)
        s.raise_for_status()
        assert torch.equal(z, z_ref) and torch.equal(lp, lp_ref), case
        keep = [0, 1, 2, 3, 5, 6, 7] if tile else list(range(8))   # tile-local control: last_dt is whichever tile wrote last
        assert torch.equal(st.cpu()[keep], st_ref.cpu()[keep]), (case, s, nat.Stats(st_ref))
        assert s.nfe > 0


def test_static_scratch_rejects_what_it_cannot_hold():
    from gan_flow_amd import _native as nat
    lib = nat.load_library()
    net = nat.net_desc(2, 64, 3, False)
    m = oracle_random_model(2, seed=3)
    params = _flat(m).to(DEV)
    x = torch.randn(20000, 2).to(DEV)       # batch-global control needs every workgroup resident: at most 16 384 rows
    with pytest.raises(nat.CnfError, match="at most"):
        nat.solve_fwd_static(net, params, x, None, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5))
    bad = nat.net_desc(2, 128, 3, False)    # an unsupported net never reaches the scratch
    with pytest.raises(nat.CnfError, match="unsupported net"):
        nat.solve_fwd_static(bad, params, x[:64], None, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5))
    # the wrapper with caller-owned outputs, twice, interleaved with an ordinary solve on another workspace
    z, lp = torch.empty(64, 2, device=DEV), torch.empty(64, 1, device=DEV)
    o = nat.make_opts("dopri5", 1e-5, 1e-5)
    a = nat.solve_fwd_static(net, params, x[:64], None, 0.0, 0.5, o, out=(z, lp))
    r = nat.solve_fwd(net, params, x[:64], None, 0.0, 0.5, o)
    b = nat.solve_fwd_static(net, params, x[:64], None, 0.0, 0.5, o)
    torch.cuda.synchronize()
    assert a[0].data_ptr() == z.data_ptr()
    assert torch.equal(a[0], r[0]) and torch.equal(b[0], r[0]) and torch.equal(b[1], r[1])
