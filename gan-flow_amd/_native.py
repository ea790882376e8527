"""ctypes binding of libcnf_hip.so (C ABI: include/cnf_hip.h).  PyTorch is plumbing only: it owns the
device memory and the stream; every number is produced by the HIP kernels.  There is NO fallback: if the
library is missing or the tensors are not on a HIP device the calls raise."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
ACTIVATIONS = {"tanh": 0, "softplus": 1, "elu": 2, "relu": 3}   # cnf_activation() of the library built for each (csrc/Makefile)


class CnfError(RuntimeError):
    """Raised for rejected calls and for integrator failures (wording follows torchdiffeq's assertions)."""


class CnfNetDesc(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("n_hidden", ctypes.c_int32),
                ("divergence", ctypes.c_int32)]


class CnfSolverOpts(ctypes.Structure):
    _fields_ = [("solver", ctypes.c_int32), ("rtol", ctypes.c_float), ("atol", ctypes.c_float),
                ("step_size", ctypes.c_float), ("tile_rows", ctypes.c_int32), ("max_steps", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("after", ctypes.c_void_p)]


MAX_REG = 2                      # CNF_MAX_REG
REG_KINDS = {"l1": 1, "l2": 2}   # CNF_REG_L1, CNF_REG_L2


class CnfRegDesc(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("kind", ctypes.c_int32 * MAX_REG)]


def reg_desc(kinds):
    """kinds: a sequence of 'l1' / 'l2' in the order of the reference's regularization_fns."""
    kinds = list(kinds)
    if not 1 <= len(kinds) <= MAX_REG or any(k not in REG_KINDS for k in kinds):
        raise CnfError("regularisers built into the HIP path: up to %d of %s (got %r)" % (MAX_REG, sorted(REG_KINDS), kinds))
    d = CnfRegDesc()
    d.n = len(kinds)
    for i, k in enumerate(kinds):
        d.kind[i] = REG_KINDS[k]
    return d


GU_MAX_MIXTURE = 16


class CnfGuDesc(ctypes.Structure):
    _fields_ = [("n_mixture", ctypes.c_int32), ("extend_dim", ctypes.c_int32), ("mean", ctypes.c_float * GU_MAX_MIXTURE),
                ("sigma", ctypes.c_float * GU_MAX_MIXTURE), ("unif_intsect", ctypes.c_float * GU_MAX_MIXTURE),
                ("unif_p", ctypes.c_float * GU_MAX_MIXTURE)]


class CnfAdamOpts(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double)]


SOLVER_IDS = {"dopri5": 0, "rk4": 1, "euler": 2, "midpoint": 3}
DIV_EXACT, DIV_HUTCHINSON = 0, 1
DIV_RESIDUAL = 256   # CNF_DIV_RESIDUAL, OR-ed into CnfNetDesc.divergence
STATUS_MESSAGES = {
    1: "underflow in dt {last_dt}",
    2: "non-finite values in state `y`",
    3: "max_num_steps exceeded",
    4: "grid-wide reduction timed out inside the solver kernel",
    5: "debug build (CNF_DEBUG_BOUNDS): a kernel formed a workspace range outside what cnf_workspace_bytes() names for the call",
}
EXPORTS = ("cnf_param_count", "cnf_workspace_bytes", "cnf_last_error", "cnf_eval_field", "cnf_solve_fwd",
           "cnf_eval_aug", "cnf_solve_bwd", "cnf_solve_bwd_train", "cnf_nll_loss", "cnf_adam_step", "cnf_debug_fill_lds",
           "cnf_debug_kernel_timing", "cnf_debug_last_kernel_ms", "cnf_activation", "cnf_debug_eval_field_b16",
           "cnf_solve_fwd_static", "cnf_static_scratch_bytes", "cnf_gu_sample", "cnf_gu_sample_rows", "cnf_solve_fwd_times",
           "cnf_solve_fwd_reg", "cnf_solve_bwd_reg")


def library_path(act="tanh"):
    """One shared object per ODEnet nonlinearity: libcnf_hip.so (tanh), libcnf_hip_softplus.so, libcnf_hip_elu.so, libcnf_hip_relu.so."""
    if act not in ACTIVATIONS:
        raise CnfError("nonlinearity %r is not built into the HIP path (have: %s)" % (act, ", ".join(ACTIVATIONS)))
    # CNF_HIP_LIB: another build of the tanh library (kernel timing experiments, tools/build_variants.sh)
    # CNF_HIP_LIBDIR: another build of all four libraries (tools/build_wide_ilp.sh)
    here = os.environ.get("CNF_HIP_LIBDIR") or _HERE
    if act == "tanh":
        return os.environ.get("CNF_HIP_LIB") or os.path.join(here, "libcnf_hip.so")
    return os.path.join(here, "libcnf_hip_%s.so" % act)


def load_library(act="tanh"):
    if act in _LIBS:
        return _LIBS[act]
    path = library_path(act)
    if not os.path.exists(path):
        raise CnfError("%s is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C gan-flow_amd/csrc`.  There is no CPU fallback." % (os.path.basename(path), path))
    lib = ctypes.CDLL(path)
    P, F, I64, SZ, VP = ctypes.c_void_p, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p
    ND, SO = ctypes.POINTER(CnfNetDesc), ctypes.POINTER(CnfSolverOpts)
    lib.cnf_param_count.restype = I64
    lib.cnf_param_count.argtypes = [ND]
    lib.cnf_workspace_bytes.restype = SZ
    lib.cnf_workspace_bytes.argtypes = [ND, I64]
    lib.cnf_last_error.restype = ctypes.c_char_p
    lib.cnf_last_error.argtypes = []
    lib.cnf_eval_field.restype = ctypes.c_int
    lib.cnf_eval_field.argtypes = [ND, P, F, P, P, I64, P, P, VP]
    lib.cnf_debug_eval_field_b16.restype = ctypes.c_int
    lib.cnf_debug_eval_field_b16.argtypes = [ND, P, F, P, P, I64, P, P, VP]
    lib.cnf_solve_fwd.restype = ctypes.c_int
    lib.cnf_solve_fwd.argtypes = [ND, P, P, P, P, I64, F, F, SO, P, P, VP, SZ, P, VP]
    RD = ctypes.POINTER(CnfRegDesc)
    lib.cnf_solve_fwd_reg.restype = ctypes.c_int
    lib.cnf_solve_fwd_reg.argtypes = [ND, RD, P, P, P, P, I64, F, F, SO, P, P, P, P, SZ, P, VP]
    lib.cnf_solve_bwd_reg.restype = ctypes.c_int
    lib.cnf_solve_bwd_reg.argtypes = [ND, RD, P, P, P, P, P, P, P, P, I64, F, F, SO, P, P, P, P, SZ, P, VP]
    lib.cnf_solve_fwd_times.restype = ctypes.c_int
    lib.cnf_solve_fwd_times.argtypes = [ND, P, P, P, P, I64, ctypes.POINTER(ctypes.c_float), ctypes.c_int, SO, P, P, P, SZ, P, VP]
    lib.cnf_solve_fwd_static.restype = ctypes.c_int
    lib.cnf_solve_fwd_static.argtypes = [ND, P, P, P, P, I64, F, F, SO, P, P, P, VP]
    lib.cnf_static_scratch_bytes.restype = SZ
    lib.cnf_static_scratch_bytes.argtypes = []
    lib.cnf_gu_sample.restype = ctypes.c_int
    lib.cnf_gu_sample.argtypes = [ctypes.POINTER(CnfGuDesc), ctypes.c_uint64, P, I64, P, VP]
    lib.cnf_gu_sample_rows.restype = ctypes.c_int
    lib.cnf_gu_sample_rows.argtypes = [ctypes.POINTER(CnfGuDesc), ctypes.c_uint64, ctypes.c_uint32, I64, I64, P, VP]
    lib.cnf_eval_aug.restype = ctypes.c_int
    lib.cnf_eval_aug.argtypes = [ND, P, F, P, P, P, P, I64, P, P, P, P, VP, SZ, VP]
    lib.cnf_solve_bwd.restype = ctypes.c_int
    lib.cnf_solve_bwd.argtypes = [ND, P, P, P, P, P, P, I64, F, F, SO, P, P, P, VP, SZ, P, VP]
    lib.cnf_solve_bwd_train.restype = ctypes.c_int
    lib.cnf_solve_bwd_train.argtypes = [ND, P, P, P, P, P, P, I64, F, F, SO, P, P, P, P, P, P, P, ctypes.POINTER(CnfAdamOpts),
                                        VP, SZ, P, VP]
    lib.cnf_nll_loss.restype = ctypes.c_int
    lib.cnf_nll_loss.argtypes = [P, P, I64, ctypes.c_int32, P, P, P, VP]
    lib.cnf_adam_step.restype = ctypes.c_int
    lib.cnf_adam_step.argtypes = [P, P, P, P, P, I64, ctypes.POINTER(CnfAdamOpts), P, VP]
    lib.cnf_debug_kernel_timing.restype = ctypes.c_int
    lib.cnf_debug_kernel_timing.argtypes = [ctypes.c_int]
    lib.cnf_debug_last_kernel_ms.restype = ctypes.c_int
    lib.cnf_debug_last_kernel_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
    lib.cnf_debug_fill_lds.restype = ctypes.c_int
    lib.cnf_debug_fill_lds.argtypes = [ctypes.c_uint32, P, VP]
    lib.cnf_activation.restype = ctypes.c_int
    lib.cnf_activation.argtypes = []
    if lib.cnf_activation() != ACTIVATIONS[act]:
        raise CnfError("%s was built for another nonlinearity (cnf_activation() = %d, expected %d)"
                       % (path, lib.cnf_activation(), ACTIVATIONS[act]))
    _LIBS[act] = lib
    return lib


def _lib_for(net):
    """The library built for the net's nonlinearity (net_desc() tags the description; plain CnfNetDesc = tanh)."""
    return load_library(getattr(net, "act", "tanh"))


def _check(rc, lib):
    if rc != 0:
        raise CnfError(lib.cnf_last_error().decode())


def _dev(t, name, shape=None):
    if t is None:
        return None
    if not torch.is_tensor(t) or not t.is_cuda:
        raise CnfError("%s must be a tensor on a HIP device (got %s); the CNF hot path has no CPU fallback"
                       % (name, getattr(t, "device", type(t))))
    if t.dtype != torch.float32:
        raise CnfError("%s must be float32 (got %s)" % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise CnfError("%s must have shape %s (got %s)" % (name, tuple(shape), tuple(t.shape)))
    return t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_WORKSPACES = {}


def _workspace(lib, net, rows, device):
    need = int(lib.cnf_workspace_bytes(ctypes.byref(net), rows))
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def workspace_bytes(net, rows):
    return int(_lib_for(net).cnf_workspace_bytes(ctypes.byref(net), rows))


def _own_workspace(lib, net, rows, ws):
    """A caller-owned workspace (uint8 tensor on the device), e.g. the zero-initialised one FLAG_CLEAN calls share."""
    need = int(lib.cnf_workspace_bytes(ctypes.byref(net), rows))
    if not (torch.is_tensor(ws) and ws.is_cuda and ws.dtype == torch.uint8 and ws.is_contiguous() and ws.numel() >= need):
        raise CnfError("workspace must be a contiguous uint8 tensor of at least %d bytes on the HIP device" % need)
    return ws


def net_desc(dim, hidden, n_hidden, hutchinson, residual=False, act="tanh"):
    """residual: ODEfunc(residual=True), dynamics f(t, y) - y (CNF_DIV_RESIDUAL).  act: the ODEnet's nonlinearity; it is not
    part of the C struct -- it selects WHICH library the calls with this description go to."""
    if act not in ACTIVATIONS:
        raise CnfError("nonlinearity %r is not built into the HIP path (have: %s)" % (act, ", ".join(ACTIVATIONS)))
    net = CnfNetDesc(int(dim), int(hidden), int(n_hidden),
                     (DIV_HUTCHINSON if hutchinson else DIV_EXACT) | (DIV_RESIDUAL if residual else 0))
    net.act = act
    return net


def param_count(net):
    n = _lib_for(net).cnf_param_count(ctypes.byref(net))
    if n < 0:
        raise CnfError("unsupported net: dim=%d hidden=%d n_hidden=%d divergence=%d"
                       % (net.dim, net.hidden, net.n_hidden, net.divergence))
    return int(n)


class Stats:
    """Host view of a CnfStats block (8 int32 words on the device)."""

    def __init__(self, words):
        w = words.tolist()
        self.status, self.nfe, self.n_accepted, self.n_rejected = w[0], w[1], w[2], w[3]
        self.last_dt = torch.tensor([w[4]], dtype=torch.int32).view(torch.float32).item()
        self.evals_run = w[5]
        self.row_evals = (w[6] & 0xFFFFFFFF) | ((w[7] & 0xFFFFFFFF) << 32)

    def raise_for_status(self):
        if self.status:
            raise CnfError(STATUS_MESSAGES.get(self.status, "solver status %d" % self.status).format(last_dt=self.last_dt))

    def __repr__(self):
        return ("Stats(status=%d, nfe=%d, accepted=%d, rejected=%d, last_dt=%g, evals_run=%d, row_evals=%d)"
                % (self.status, self.nfe, self.n_accepted, self.n_rejected, self.last_dt, self.evals_run, self.row_evals))


FLAG_CLEAN = 1   # CNF_FLAG_CLEAN, include/cnf_hip.h
FLAG_SPREAD = 2  # CNF_FLAG_SPREAD
FLAG_FWD_FOUR_WAVES = 8  # CNF_FLAG_FWD_FOUR_WAVES: diagnostics, the four-wave forward kernel (same bits)
FLAG_ADJ_CONTROL_ALL = 4  # CNF_FLAG_ADJ_CONTROL_ALL: a_z and a_logp under the adjoint solve's step control as well


def make_opts(solver="dopri5", rtol=1e-5, atol=1e-5, step_size=None, tile_rows=0, max_steps=0, flags=0, after=None):
    """after: the stats words (8 int32 on the device) of the solve whose output this one consumes; a non-zero status there
    makes this solve exit at once with the same status (CnfSolverOpts.after, include/cnf_hip.h)."""
    if solver not in SOLVER_IDS:
        raise CnfError("solver %r is not built into the HIP path (have: %s)" % (solver, ", ".join(SOLVER_IDS)))
    return CnfSolverOpts(SOLVER_IDS[solver], float(rtol), float(atol), float(step_size or 0.0), int(tile_rows), int(max_steps),
                         int(flags), _stats_ptr(after))


def _stats_ptr(words):
    if words is None:
        return None
    if not (torch.is_tensor(words) and words.is_cuda and words.dtype == torch.int32 and words.numel() == 8 and words.is_contiguous()):
        raise CnfError("a stats block must be 8 contiguous int32 words on the HIP device")
    return words.data_ptr()


def with_after(opts, words):
    """A copy of `opts` whose `after` names the stats block `words` (or nothing)."""
    return CnfSolverOpts(opts.solver, opts.rtol, opts.atol, opts.step_size, opts.tile_rows, opts.max_steps, opts.flags,
                         _stats_ptr(words))


def _stats_block(stats, device):
    """The 32-byte CnfStats block a solve writes: a fresh one, or the caller's (8 contiguous int32 on `device`)."""
    if stats is None:
        return torch.empty(8, dtype=torch.int32, device=device)   # the call zeroes it (no FLAG_CLEAN without a caller's block)
    if not (stats.is_cuda and stats.dtype == torch.int32 and stats.numel() == 8 and stats.is_contiguous()):
        raise CnfError("stats must be 8 contiguous int32 words on the HIP device")
    return stats


def eval_field(net, params, t, z, e=None, b16=False):
    """(f, -div) = ODEfunc.forward(t, (z, .)) for a (rows, dim) batch.  b16: computed the way the forward dopri5 solves at
    hidden width 64, dim <= 2 compute it (hidden layers on the bf16 pipe with three-piece operands; diagnostics)."""
    lib = _lib_for(net)
    z = _dev(z, "z")
    rows = z.shape[0]
    params = _dev(params, "params", (param_count(net),))
    e = _dev(e, "e", z.shape) if e is not None else None
    f = torch.empty_like(z)
    nd = torch.empty(rows, 1, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        fn = lib.cnf_debug_eval_field_b16 if b16 else lib.cnf_eval_field
        _check(fn(ctypes.byref(net), _ptr(params), float(t), _ptr(z), _ptr(e), rows, _ptr(f), _ptr(nd), _stream()), lib)
    return f, nd


def solve_fwd(net, params, x, logp0, t0, t1, opts, e=None, stats=None, workspace=None):
    """State (z, logp) at t1 of the CNF ODE started from (x, logp0) at t0.  Returns (z, logp, stats_words)."""
    lib = _lib_for(net)
    x = _dev(x, "x")
    rows = x.shape[0]
    params = _dev(params, "params", (param_count(net),))
    logp0 = _dev(logp0, "logp0", (rows, 1)) if logp0 is not None else None
    e = _dev(e, "e", x.shape) if e is not None else None
    z = torch.empty_like(x)
    logp = torch.empty(rows, 1, dtype=torch.float32, device=x.device)
    stats = _stats_block(stats, x.device)
    with torch.cuda.device(x.device):
        ws = _workspace(lib, net, rows, x.device) if workspace is None else _own_workspace(lib, net, rows, workspace)
        _check(lib.cnf_solve_fwd(ctypes.byref(net), _ptr(params), _ptr(x), _ptr(logp0), _ptr(e), rows, float(t0), float(t1),
                                 ctypes.byref(opts), _ptr(z), _ptr(logp), _ptr(ws), ws.numel(), _ptr(stats), _stream()), lib)
    return z, logp, stats


def solve_fwd_reg(net, reg, params, x, logp0, t0, t1, opts, e=None, stats=None):
    """cnf_solve_fwd_reg: solve_fwd of the regularised dynamics.  Returns (z, logp, reg_states[reg.n], stats_words)."""
    lib = _lib_for(net)
    x = _dev(x, "x")
    rows = x.shape[0]
    params = _dev(params, "params", (param_count(net),))
    logp0 = _dev(logp0, "logp0", (rows, 1)) if logp0 is not None else None
    e = _dev(e, "e", x.shape) if e is not None else None
    z = torch.empty_like(x)
    logp = torch.empty(rows, 1, dtype=torch.float32, device=x.device)
    regs = torch.empty(reg.n, dtype=torch.float32, device=x.device)
    stats = _stats_block(stats, x.device)
    with torch.cuda.device(x.device):
        ws = _workspace(lib, net, rows, x.device)
        _check(lib.cnf_solve_fwd_reg(ctypes.byref(net), ctypes.byref(reg), _ptr(params), _ptr(x), _ptr(logp0), _ptr(e), rows,
                                     float(t0), float(t1), ctypes.byref(opts), _ptr(z), _ptr(logp), _ptr(regs), _ptr(ws),
                                     ws.numel(), _ptr(stats), _stream()), lib)
    return z, logp, regs, stats


def solve_bwd_reg(net, reg, params, z1, logp1, reg_t1, grad_z1, grad_logp1, grad_reg, t0, t1, opts, e=None, stats=None):
    """cnf_solve_bwd_reg: the adjoint pass of the regularised dynamics.  reg_t1: the regularisation states at t1 (reg.n
    floats on the device), grad_reg: dL/d(those states).  Returns (grad_x, grad_logp0, grad_params, stats_words)."""
    lib = _lib_for(net)
    z1 = _dev(z1, "z1")
    rows = z1.shape[0]
    params = _dev(params, "params", (param_count(net),))
    logp1 = _dev(logp1, "logp1", (rows, 1))
    grad_z1 = _dev(grad_z1, "grad_z1", z1.shape)
    grad_logp1 = _dev(grad_logp1, "grad_logp1", (rows, 1))
    reg_t1 = _dev(reg_t1, "reg_t1", (reg.n,))
    grad_reg = _dev(grad_reg, "grad_reg", (reg.n,))
    e = _dev(e, "e", z1.shape) if e is not None else None
    gx = torch.empty_like(z1)
    gl = torch.empty_like(logp1)
    gp = torch.empty_like(params)
    stats = _stats_block(stats, z1.device)
    with torch.cuda.device(z1.device):
        ws = _workspace(lib, net, rows, z1.device)
        _check(lib.cnf_solve_bwd_reg(ctypes.byref(net), ctypes.byref(reg), _ptr(params), _ptr(z1), _ptr(logp1), _ptr(e),
                                     _ptr(reg_t1), _ptr(grad_z1), _ptr(grad_logp1), _ptr(grad_reg), rows, float(t0), float(t1),
                                     ctypes.byref(opts), _ptr(gx), _ptr(gl), _ptr(gp), _ptr(ws), ws.numel(), _ptr(stats),
                                     _stream()), lib)
    return gx, gl, gp, stats


MAX_TIMES = 32   # CNF_MAX_TIMES, include/cnf_hip.h


def solve_fwd_times(net, params, x, logp0, times, opts, e=None, stats=None):
    """cnf_solve_fwd_times: the state at every time of `times` (a sequence of floats, strictly monotonic), one adaptive
    integration.  Returns (z_t [n, rows, dim], logp_t [n, rows, 1], stats_words); slab 0 is the input."""
    lib = _lib_for(net)
    x = _dev(x, "x")
    rows = x.shape[0]
    times = [float(t) for t in times]
    n = len(times)
    if not 2 <= n <= MAX_TIMES:
        raise CnfError("between 2 and %d output times (got %d)" % (MAX_TIMES, n))
    params = _dev(params, "params", (param_count(net),))
    logp0 = _dev(logp0, "logp0", (rows, 1)) if logp0 is not None else None
    e = _dev(e, "e", x.shape) if e is not None else None
    z = torch.empty((n,) + tuple(x.shape), dtype=torch.float32, device=x.device)
    logp = torch.empty(n, rows, 1, dtype=torch.float32, device=x.device)
    stats = _stats_block(stats, x.device)
    host_times = (ctypes.c_float * n)(*times)
    with torch.cuda.device(x.device):
        ws = _workspace(lib, net, rows, x.device)
        _check(lib.cnf_solve_fwd_times(ctypes.byref(net), _ptr(params), _ptr(x), _ptr(logp0), _ptr(e), rows, host_times, n,
                                       ctypes.byref(opts), _ptr(z), _ptr(logp), _ptr(ws), ws.numel(), _ptr(stats), _stream()), lib)
    if n == 2:   # the library ran the ordinary solve: its one output slab is the state at times[1]
        z = torch.stack([x, z[0]])
        logp = torch.stack([logp0 if logp0 is not None else torch.zeros_like(logp[0]), logp[0]])
    return z, logp, stats


def solve_fwd_static(net, params, x, logp0, t0, t1, opts, e=None, stats=None, out=None):
    """cnf_solve_fwd_static: solve_fwd on the library's own static scratch (no workspace, nothing allocated by the call).
    out: (z, logp) tensors to write into (a caller that must not allocate passes them); else fresh ones."""
    lib = _lib_for(net)
    x = _dev(x, "x")
    rows = x.shape[0]
    params = _dev(params, "params", (param_count(net),))
    logp0 = _dev(logp0, "logp0", (rows, 1)) if logp0 is not None else None
    e = _dev(e, "e", x.shape) if e is not None else None
    z, logp = out if out is not None else (torch.empty_like(x), torch.empty(rows, 1, dtype=torch.float32, device=x.device))
    stats = _stats_block(stats, x.device)
    with torch.cuda.device(x.device):
        _check(lib.cnf_solve_fwd_static(ctypes.byref(net), _ptr(params), _ptr(x), _ptr(logp0), _ptr(e), rows, float(t0),
                                        float(t1), ctypes.byref(opts), _ptr(z), _ptr(logp), _ptr(stats), _stream()), lib)
    return z, logp, stats


def eval_aug(net, params, t, z, a_z, a_logp, e=None):
    """One evaluation of the adjoint's augmented dynamics: (f, -div, vjp_z, vjp_params)."""
    lib = _lib_for(net)
    z = _dev(z, "z")
    rows = z.shape[0]
    params = _dev(params, "params", (param_count(net),))
    a_z = _dev(a_z, "a_z", z.shape)
    a_logp = _dev(a_logp, "a_logp", (rows, 1))
    e = _dev(e, "e", z.shape) if e is not None else None
    f = torch.empty_like(z)
    nd = torch.empty(rows, 1, dtype=torch.float32, device=z.device)
    vz = torch.empty_like(z)
    vp = torch.empty_like(params)
    with torch.cuda.device(z.device):
        ws = _workspace(lib, net, rows, z.device)
        _check(lib.cnf_eval_aug(ctypes.byref(net), _ptr(params), float(t), _ptr(z), _ptr(e), _ptr(a_z), _ptr(a_logp), rows,
                                _ptr(f), _ptr(nd), _ptr(vz), _ptr(vp), _ptr(ws), ws.numel(), _stream()), lib)
    return f, nd, vz, vp


def solve_bwd(net, params, z1, logp1, grad_z1, grad_logp1, t0, t1, opts, e=None, stats=None, workspace=None):
    """Adjoint pass for the interval [t0, t1]: returns (grad_x, grad_logp0, grad_params, stats_words)."""
    lib = _lib_for(net)
    z1 = _dev(z1, "z1")
    rows = z1.shape[0]
    params = _dev(params, "params", (param_count(net),))
    logp1 = _dev(logp1, "logp1", (rows, 1))
    grad_z1 = _dev(grad_z1, "grad_z1", z1.shape)
    grad_logp1 = _dev(grad_logp1, "grad_logp1", (rows, 1))
    e = _dev(e, "e", z1.shape) if e is not None else None
    gx = torch.empty_like(z1)
    gl = torch.empty_like(logp1)
    gp = torch.empty_like(params)
    stats = _stats_block(stats, z1.device)
    with torch.cuda.device(z1.device):
        ws = _workspace(lib, net, rows, z1.device) if workspace is None else _own_workspace(lib, net, rows, workspace)
        _check(lib.cnf_solve_bwd(ctypes.byref(net), _ptr(params), _ptr(z1), _ptr(logp1), _ptr(e), _ptr(grad_z1),
                                 _ptr(grad_logp1), rows, float(t0), float(t1), ctypes.byref(opts), _ptr(gx), _ptr(gl),
                                 _ptr(gp), _ptr(ws), ws.numel(), _ptr(stats), _stream()), lib)
    return gx, gl, gp, stats


def solve_bwd_train(net, params, z1, logp1, grad_z1, grad_logp1, t0, t1, opts, exp_avg, exp_avg_sq, state, e=None, stats=None,
                    workspace=None, want_grad_params=False, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """cnf_solve_bwd_train: adjoint pass + Adam update of `params` (in place) in two kernels.  grad_z1 = grad_logp1 = None:
    the cotangents of the flow's NLL, formed in the kernel; the loss is returned.
    Returns (grad_x, grad_logp0, grad_params or None, loss[1] or None, stats_words)."""
    lib = _lib_for(net)
    z1 = _dev(z1, "z1")
    rows = z1.shape[0]
    n = param_count(net)
    for name, t in (("params", params), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if _dev(t, name, (n,)).data_ptr() != t.data_ptr():
            raise CnfError("%s must be contiguous" % name)
    if not (torch.is_tensor(state) and state.is_cuda and state.dtype == torch.int32 and state.numel() == 2):
        raise CnfError("state must be 2 int32 words on the HIP device")
    logp1 = _dev(logp1, "logp1", (rows, 1))
    nll = grad_z1 is None
    if nll and grad_logp1 is not None:
        raise CnfError("grad_z1=None selects the NLL cotangents: grad_logp1 must be None too")
    grad_z1 = None if nll else _dev(grad_z1, "grad_z1", z1.shape)
    grad_logp1 = None if nll else _dev(grad_logp1, "grad_logp1", (rows, 1))
    e = _dev(e, "e", z1.shape) if e is not None else None
    gx = torch.empty_like(z1)
    gl = torch.empty_like(logp1)
    gp = torch.empty_like(params) if want_grad_params else None
    loss = torch.empty(1, dtype=torch.float32, device=z1.device) if nll else None
    stats = _stats_block(stats, z1.device)
    o = CnfAdamOpts(float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay))
    with torch.cuda.device(z1.device):
        ws = _workspace(lib, net, rows, z1.device) if workspace is None else _own_workspace(lib, net, rows, workspace)
        _check(lib.cnf_solve_bwd_train(ctypes.byref(net), _ptr(params), _ptr(z1), _ptr(logp1), _ptr(e), _ptr(grad_z1),
                                       _ptr(grad_logp1), rows, float(t0), float(t1), ctypes.byref(opts), _ptr(gx), _ptr(gl),
                                       _ptr(gp), _ptr(loss), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(state), ctypes.byref(o),
                                       _ptr(ws), ws.numel(), _ptr(stats), _stream()), lib)
    return gx, gl, gp, loss, stats


def nll_loss(z, delta_logp, want_grads=True):
    """-mean(sum_d log N(z_d; 0, 1) - delta_logp) of a batch, with d loss / d z and d loss / d delta_logp.
    Returns (loss[1], grad_z or None, grad_delta_logp or None)."""
    lib = load_library()
    z = _dev(z, "z")
    rows, dim = z.shape
    delta_logp = _dev(delta_logp, "delta_logp", (rows, 1))
    loss = torch.empty(1, dtype=torch.float32, device=z.device)
    gz = torch.empty_like(z) if want_grads else None
    gl = torch.empty_like(delta_logp) if want_grads else None
    with torch.cuda.device(z.device):
        _check(lib.cnf_nll_loss(_ptr(z), _ptr(delta_logp), rows, dim, _ptr(loss), _ptr(gz), _ptr(gl), _stream()), lib)
    return loss, gz, gl


def adam_step(params, grad, exp_avg, exp_avg_sq, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, guard=None):
    """In-place torch.optim.Adam update of a flat fp32 vector; `state` = 2 int32 on the device ({step, internal}).
    guard: stats words of the solve that produced `grad`; a non-zero status there turns the update into a no-op."""
    lib = load_library()
    n = params.numel()
    for name, t in (("params", params), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if _dev(t, name, (n,)).data_ptr() != t.data_ptr():
            raise CnfError("%s must be contiguous" % name)
    if not (torch.is_tensor(state) and state.is_cuda and state.dtype == torch.int32 and state.numel() == 2):
        raise CnfError("state must be 2 int32 words on the HIP device")
    o = CnfAdamOpts(float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay))
    with torch.cuda.device(params.device):
        _check(lib.cnf_adam_step(_ptr(params), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(state), n,
                                 ctypes.byref(o), ctypes.c_void_p(_stats_ptr(guard)), _stream()), lib)


def gu_desc(mean, sigma, unif_intsect, unif_p, extend_dim):
    n = len(mean)
    if not 1 <= n <= GU_MAX_MIXTURE:
        raise CnfError("the device sampler holds 1..%d mixture components (got %d)" % (GU_MAX_MIXTURE, n))
    d = CnfGuDesc(n, 1 if extend_dim else 0)
    for i in range(n):
        d.mean[i], d.sigma[i], d.unif_intsect[i], d.unif_p[i] = float(mean[i]), float(sigma[i]), float(unif_intsect[i]), float(unif_p[i])
    return d


def gu_sample(desc, seed, state, out):
    """cnf_gu_sample: fill `out` (rows x (2 if extend_dim else 1), fp32, on the device) with a fresh batch of the mixture;
    `state` = 2 int32 on the device ({call number, internal}), zeroed before the first call."""
    lib = load_library()
    cols = 2 if desc.extend_dim else 1
    if not (torch.is_tensor(out) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 2
            and out.shape[1] == cols):
        raise CnfError("out must be a contiguous fp32 (rows, %d) tensor on the HIP device" % cols)
    if not (torch.is_tensor(state) and state.is_cuda and state.dtype == torch.int32 and state.numel() == 2):
        raise CnfError("state must be 2 int32 words on the HIP device")
    with torch.cuda.device(out.device):
        _check(lib.cnf_gu_sample(ctypes.byref(desc), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), _ptr(state), out.shape[0], _ptr(out),
                                 _stream()), lib)
    return out


def gu_sample_rows(desc, seed, call, row0, out):
    """cnf_gu_sample_rows: rows row0 .. row0 + len(out) - 1 of draw number `call`, addressed by global row index (stateless:
    a shard of a data set holds the same values for every number of shards)."""
    lib = load_library()
    cols = 2 if desc.extend_dim else 1
    if not (torch.is_tensor(out) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 2
            and out.shape[1] == cols):
        raise CnfError("out must be a contiguous fp32 (rows, %d) tensor on the HIP device" % cols)
    if out.shape[0] == 0:
        return out
    with torch.cuda.device(out.device):
        _check(lib.cnf_gu_sample_rows(ctypes.byref(desc), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), ctypes.c_uint32(int(call) & 0xFFFFFFFF),
                                      int(row0), out.shape[0], _ptr(out), _stream()), lib)
    return out


def debug_fill_lds(pattern=0x7FC00000, device=None):
    """Diagnostics: overwrite the LDS of every CU with `pattern` (default: a quiet NaN)."""
    lib = load_library()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    word = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _check(lib.cnf_debug_fill_lds(int(pattern), _ptr(word), _stream()), lib)
    return word


def kernel_timing(enable):
    """Diagnostics: bracket the solve kernel of every following solve call with HIP events (see last_kernel_ms)."""
    lib = load_library()
    _check(lib.cnf_debug_kernel_timing(1 if enable else 0), lib)


def last_kernel_ms():
    """Duration of the most recent solve kernel launched while kernel_timing(True) was on (waits for it)."""
    lib = load_library()
    ms = ctypes.c_float(0.0)
    _check(lib.cnf_debug_last_kernel_ms(ctypes.byref(ms)), lib)
    return float(ms.value)
