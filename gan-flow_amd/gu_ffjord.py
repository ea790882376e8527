"""FFJORD on the Gaussian-uniform mixtures: model builder, eval transforms, loss and the training driver.

Stands in for the reference's `ffjord.gu_ffjord` / `ffjord.gu_maf` (imported at /root/reference/model_plot.py:2, run as
`python3 -m ffjord.gu_maf --gu_num {1,8}` per /root/reference/README.md:41,57; defaults 50 000 iterations, batch 2 048,
README.md:28).  Run:  python -m gan_flow_amd.gu_maf --gu_num 8 --niters 50000 --batch_size 2048
"""
import argparse
import math
import os
import time

import torch

from .gu import make_gu
from .lib.layers import CNF, MovingBatchNorm1d, ODEfunc, ODEnet, SequentialFlow
from .util import count_parameters, get_logger, makedirs, w_distance


def build_model(dim=2, dims=(64, 64, 64), num_blocks=1, time_length=0.5, train_T=True, divergence_fn="brute_force",
                solver="dopri5", atol=1e-5, rtol=1e-5, rademacher=False, layer_type="concatsquash", residual=False,
                batch_norm=False, bn_lag=0.0, nonlinearity="tanh", regularization_fns=()):
    """`num_blocks` CNF blocks over a dim -> dims -> dim concatsquash/tanh ODEnet: the architecture stored in the
    reference's checkpoints.  Other `layer_type`s: see ODEnet."""
    blocks = []
    for _ in range(num_blocks):
        net = ODEnet(tuple(dims), (dim,), strides=None, conv=False, layer_type=layer_type, nonlinearity=nonlinearity)
        fn = ODEfunc(net, divergence_fn=divergence_fn, residual=residual, rademacher=rademacher)
        blocks.append(CNF(fn, T=time_length, train_T=train_T, regularization_fns=tuple(regularization_fns), solver=solver, atol=atol,
                          rtol=rtol))
    if batch_norm:   # build_model_tabular's arrangement: a norm layer first, then one after every CNF block
        chain = [MovingBatchNorm1d(dim, bn_lag=bn_lag)]
        for blk in blocks:
            chain += [blk, MovingBatchNorm1d(dim, bn_lag=bn_lag)]
        blocks = chain
    return SequentialFlow(blocks)


def get_transforms(model):
    """-> (sample_fn, density_fn) as used at /root/reference/model_plot.py:94-98."""

    def sample_fn(z, logpz=None):
        return model(z, reverse=True) if logpz is None else model(z, logpz, reverse=True)

    def density_fn(x, logpx=None):
        return model(x, reverse=False) if logpx is None else model(x, logpx, reverse=False)

    return sample_fn, density_fn


def standard_normal_logprob(z):
    return -0.5 * math.log(2 * math.pi) - z.pow(2) / 2


def compute_loss(model, x):
    """Negative mean log-likelihood of a batch under the flow."""
    zero = torch.zeros(x.shape[0], 1, device=x.device, dtype=x.dtype)
    z, delta_logp = model(x, zero)
    logpz = standard_normal_logprob(z).view(z.shape[0], -1).sum(1, keepdim=True)
    return -torch.mean(logpz - delta_logp)


def get_regularization(model, coeffs):
    """Sum over the CNF blocks of their regularisation states, weighted: what ffjord/train_misc.py adds to the loss
    (`--l1int` / `--l2int`).  Reads (and clears) the states the last training-mode forward left in the blocks."""
    total = None
    for block in model.chain:
        if isinstance(block, CNF) and block.nreg:
            states = block.get_regularization_states()
            if len(states) != len(coeffs):
                raise ValueError("%d regularisation states, %d coefficients" % (len(states), len(coeffs)))
            part = sum(c * s for c, s in zip(coeffs, states))
            total = part if total is None else total + part
    return total


def count_nfe(model):
    return sum(block.num_evals() for block in model.chain if hasattr(block, "num_evals"))


class checked_solves:
    """Within the block every CNF block reads its solver status back after each solve and raises on failure (sync_stats),
    whatever a captured training step had set: evaluation must never turn a failed solve (NaN, max steps, sync timeout)
    into a logged number."""

    def __init__(self, model):
        self.blocks = [b for b in model.chain if hasattr(b, "sync_stats")]

    def __enter__(self):
        self.saved = [b.sync_stats for b in self.blocks]
        for b in self.blocks:
            b.sync_stats = True

    def __exit__(self, *exc):
        for b, v in zip(self.blocks, self.saved):
            b.sync_stats = v
        return False


def evaluate_w_distance(model, data, eval_size=100000, chunk=10000, device=None):
    """The reference's sampling protocol (model_plot.py:93-101): eval_size prior draws pushed through sample_fn in
    10 000-row chunks, 1-D Wasserstein distance of the first coordinate against fresh data.  `data`: anything with
    get_sample(n) -- the host mixture, or its DeviceSampler (then nothing leaves the GPU but the final number)."""
    device = device or next(model.parameters()).device
    dim = model.chain[0].odefunc.odefunc.diffeq.dims[0]
    was_training = model.training
    model.eval()
    sample_fn, _ = get_transforms(model)
    with torch.no_grad(), checked_solves(model):
        z = torch.randn(eval_size, dim, device=device)
        fake = torch.cat([sample_fn(zc) for zc in torch.split(z, chunk)], 0)
        real = data.get_sample(eval_size).to(device)
    model.train(was_training)
    return w_distance(real[:, 0], fake[:, 0])


class GraphedTrainStep:
    """One training iteration (forward solve, loss, adjoint solve, Adam) captured into a single hipGraph.

    The reference's loop costs a host round trip per Runge-Kutta step; here a whole solve is already one kernel, and
    what is left of an iteration on the host (a few dozen small PyTorch launches, two status read-backs) is as long
    as the kernels themselves early in training.  Capturing the iteration removes it: replay = one graph launch.
    Requirements: fixed batch shape, an optimizer built with capturable=True, no host sync inside (the solver status
    is left in device memory and checked by .check()).  `warmup` eager iterations run first on a side stream (they
    are real training steps)."""

    def __init__(self, model, optimizer, example_batch, warmup_batches=()):
        self.model, self.optimizer = model, optimizer
        self.static_x = example_batch.clone()
        for blk in model.chain:
            blk.sync_stats = False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for xb in warmup_batches:
                self.static_x.copy_(xb)
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.check()
        self.nfe = torch.zeros(2, dtype=torch.int64, device=example_batch.device)   # running sums: forward, backward
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may call HIP
            self.static_loss = self._eager()
            st = [model.chain[0].last_stats["forward"], model.chain[0].last_stats["backward"]]
            self._captured_stats = [(blk, dict(blk.last_stats)) for blk in model.chain]
            self.nfe += torch.stack([st[0][1], st[1][1]]).to(torch.int64)

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = compute_loss(self.model, self.static_x)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, x):
        self.static_x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_loss

    def check(self):
        """Read the solver status words back (one sync) and raise what the eager path would have raised."""
        from . import _native
        nfe = [0, 0]
        # the captured iteration's own blocks when there are any: eager solves in between (evaluation) replace blk.last_stats
        for blk, stats in getattr(self, "_captured_stats", None) or [(b, b.last_stats) for b in self.model.chain]:
            for i, key in enumerate(("forward", "backward")):
                words = stats.get(key)
                if torch.is_tensor(words):
                    st = _native.Stats(words)
                    st.raise_for_status()
                    nfe[i] += st.nfe
        self.last_nfe = tuple(nfe)


class FusedTrainStep:
    """One training iteration with nothing but the library's kernels in it:

        forward solve(s) -> cnf_nll_loss (loss + its gradient) -> adjoint solve(s) -> cnf_adam_step

    i.e. 4 kernels (+ the slab reduction) per CNF block instead of the ~100 launches that autograd, the loss
    expression and torch.optim.Adam add around the two solves; at hidden width 64 cnf_solve_bwd_train folds the loss
    into the adjoint kernel and the Adam update into the slab reduction: 3 kernels per block (fuse_tail).  The arithmetic is that of
    `loss = compute_loss(model, x); loss.backward(); Adam(lr, betas, eps, weight_decay).step()`; the parameters
    are updated in place through ODEnet.flatten_parameters(), so `model` (state_dict, evaluation) always sees the
    trained weights.  As in the reference, `sqrt_end_time` receives no gradient (T enters the solve detached).
    With graph=True the iteration is captured once and replayed as one hipGraph; the solver status words stay on the
    device until .check()."""

    def __init__(self, model, example_batch, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, graph=True,
                 warmup_batches=(), spread=True, fuse_tail=True, sampler=None, warmup=0, reg_coeffs=None):
        # sampler: a gu.DeviceSampler (or anything with .fill(tensor)): every iteration starts by drawing its batch on the
        # device straight into the static input (one more kernel in the captured graph, no host sampler, no copy) and
        # step() takes no argument; `warmup` such iterations run before the capture (warmup_batches is for fed batches)
        self.sampler = sampler
        if sampler is not None and len(warmup_batches):
            raise ValueError("with a device sampler the warm-up iterations draw their own batches: pass warmup=N")
        if sampler is not None:
            warmup_batches = [None] * int(warmup)
        # fuse_tail: use cnf_solve_bwd_train (same arithmetic, fewer kernels) where it is built
        self.fuse_tail = bool(fuse_tail)
        # spread: flag the adjoint solves CNF_FLAG_SPREAD (include/cnf_hip.h): two workgroups per 16-row tile when the
        # batch would leave half the compute units idle.  Same results; must be False for trials that share a GPU
        # (ConcurrentTrials), whose solves have to be resident side by side.
        self.spread = bool(spread)
        from . import _native
        from .lib.layers.cnf import _inner
        self._native, self.model = _native, model
        self.hyper = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.blocks = list(model.chain)
        # reg_coeffs: the weights of the blocks' regularisation states in the loss (`--l1int` / `--l2int`, in the order of the
        # blocks' regularization_fns): loss = nll + sum over blocks and k of c_k r_k.  Regularised blocks run cnf_solve_fwd_reg /
        # cnf_solve_bwd_reg (the adjoint of a state is the constant c_k) and cnf_adam_step: library kernels only, capturable
        for blk in self.blocks:
            if not isinstance(blk, CNF):
                raise NotImplementedError("FusedTrainStep handles a SequentialFlow of CNF blocks")
            if blk.nreg and (reg_coeffs is None or len(reg_coeffs) != blk.nreg):
                raise ValueError("regularised blocks need reg_coeffs, one coefficient per regularisation function (%d)" % blk.nreg)
        self.reg_coeffs = None if reg_coeffs is None else torch.tensor([float(c) for c in reg_coeffs], dtype=torch.float32,
                                                                       device=example_batch.device)
        self.nets = [_inner(blk.odefunc).diffeq for blk in self.blocks]
        self.flats = [net.flatten_parameters() for net in self.nets]
        # entries of the flat buffers that are not parameters (zero padding of narrower nets, the constants of the 'ignore' /
        # 'concat_v2' / 'squash' layer types): restored after every update, one index_copy_ per block in the captured graph
        self.consts = [net.flat_constants for net in self.nets]
        self.exp_avg = [torch.zeros_like(f) for f in self.flats]
        self.exp_avg_sq = [torch.zeros_like(f) for f in self.flats]
        dev = example_batch.device
        self.adam_state = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in self.flats]
        self.stats = torch.zeros(len(self.blocks), 2, 8, dtype=torch.int32, device=dev)   # [block][fwd, bwd][CnfStats words]
        # a workspace of its own, zero-initialised and only ever used by CNF_FLAG_CLEAN calls: the solves enqueue no memsets
        need = max(_native.workspace_bytes(_inner(blk.odefunc).net_desc(), example_batch.shape[0]) for blk in self.blocks)
        self.workspace = torch.zeros(need, dtype=torch.uint8, device=dev)
        self.static_x = example_batch.clone()
        self.static_loss = None
        self.graph = None
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            # The LAST warm-up iteration runs as the graph's first replay (same kernels, same order, same bits as an eager
            # one): a process's first hipGraphLaunch pays ~0.3 ms of one-time work (the graph's upload), which belongs to
            # the warm-up, not to the caller's first real iteration.
            eager, replayed = list(warmup_batches[:-1]), list(warmup_batches[-1:])
            with torch.cuda.stream(side):
                for xb in eager:
                    if xb is not None:
                        self.static_x.copy_(xb)
                    self._iteration()
                if not len(eager):                   # capture needs the library loaded and the workspace allocated
                    self._dry_run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.check()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may call HIP
                self.static_loss = self._iteration()
            for xb in replayed:
                if xb is not None:
                    self.static_x.copy_(xb)
                self.graph.replay()
            if len(replayed):
                torch.cuda.synchronize()
                self.check()
        else:
            for xb in warmup_batches:
                if xb is not None:
                    self.static_x.copy_(xb)
                self._iteration()

    def _dry_run(self):
        """A forward + adjoint solve that changes nothing (sizes the cached workspace before capture)."""
        saved = [f.clone() for f in self.flats], [s.clone() for s in self.adam_state]
        m, v = [t.clone() for t in self.exp_avg], [t.clone() for t in self.exp_avg_sq]
        self._iteration()
        for dst, src in zip(self.flats + self.adam_state + self.exp_avg + self.exp_avg_sq, saved[0] + saved[1] + m + v):
            dst.copy_(src)

    def _iteration(self):
        nat = self._native
        # every solve names the stats block of the solve before it (CnfSolverOpts.after): a failure anywhere makes all
        # later solves of the iteration exit at once with that status, and the Adam updates skip themselves -- the
        # parameters are never updated from the garbage a failed solve leaves behind
        if self.sampler is not None:
            self.sampler.fill(self.static_x)
        z, lp, tape, prev, reg_terms = self.static_x, None, [], None, []
        for i, blk in enumerate(self.blocks):
            net, t0, t1, opts, e = blk.solve_plan(z, reset_counter=False)
            if opts.solver == 0 and opts.tile_rows == 0:
                opts.flags |= nat.FLAG_CLEAN
            if self.spread:
                opts.flags |= nat.FLAG_SPREAD
            if blk.nreg and blk.training:
                reg = nat.reg_desc(blk.odefunc.reg_kinds)
                z, lp, regs, prev = nat.solve_fwd_reg(net, reg, self.flats[i], z, lp, t0, t1, nat.with_after(opts, prev), e,
                                                      stats=self.stats[i, 0])
                reg_terms.append((self.reg_coeffs * regs).sum())
                tape.append((net, t0, t1, opts, e, z, lp, (reg, regs)))
                continue
            z, lp, prev = nat.solve_fwd(net, self.flats[i], z, lp, t0, t1, nat.with_after(opts, prev), e, stats=self.stats[i, 0],
                                        workspace=self.workspace)
            tape.append((net, t0, t1, opts, e, z, lp, None))
        # the tail: cnf_solve_bwd_train (adjoint + slab sum + Adam in two kernels; for the last block also the loss and its
        # cotangents) where it is built -- hidden width 64, <= 4096 rows -- else cnf_nll_loss, cnf_solve_bwd, cnf_adam_step
        last = len(self.blocks) - 1
        # (fixed-grid solvers: cnf_solve_bwd_train takes the loss kernel's cotangents, at dim <= 2)
        fixed = [tape[i][3].solver != 0 for i in range(last + 1)]
        fused = [self.fuse_tail and tape[i][0].hidden == 64 and z.shape[0] <= 4096 and not (fixed[i] and tape[i][0].dim > 2)
                 and tape[i][7] is None for i in range(last + 1)]
        loss = gz = gl = None
        if not fused[last] or fixed[last]:
            loss, gz, gl = nat.nll_loss(z, lp)
        for i in range(last, -1, -1):
            net, t0, t1, opts, e, z1, lp1, regd = tape[i]
            opts = nat.with_after(opts, prev)
            if regd is not None:
                gz, gl, gp, prev = nat.solve_bwd_reg(net, regd[0], self.flats[i], z1, lp1, regd[1], gz, gl, self.reg_coeffs, t0, t1, opts, e,
                                                     stats=self.stats[i, 1])
                nat.adam_step(self.flats[i], gp, self.exp_avg[i], self.exp_avg_sq[i], self.adam_state[i], guard=prev, **self.hyper)
            elif fused[i]:
                gz, gl, _, l_, prev = nat.solve_bwd_train(net, self.flats[i], z1, lp1, gz, gl, t0, t1, opts, self.exp_avg[i],
                                                          self.exp_avg_sq[i], self.adam_state[i], e, stats=self.stats[i, 1],
                                                          workspace=self.workspace, **self.hyper)
                loss = l_ if l_ is not None else loss
            else:
                gz, gl, gp, prev = nat.solve_bwd(net, self.flats[i], z1, lp1, gz, gl, t0, t1, opts, e, stats=self.stats[i, 1],
                                                 workspace=self.workspace)
                nat.adam_step(self.flats[i], gp, self.exp_avg[i], self.exp_avg_sq[i], self.adam_state[i], guard=prev, **self.hyper)
            if self.consts[i] is not None:
                self.flats[i].index_copy_(0, *self.consts[i])
        for term in reg_terms:
            loss = loss + term
        return loss

    def __call__(self, x=None):
        if (x is None) != (self.sampler is not None):
            raise ValueError("pass a batch, or build the step with sampler= and call it without one")
        if x is not None:
            self.static_x.copy_(x, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
            return self.static_loss
        return self._iteration()

    def check(self):
        """Read the solver status words of the last iteration back (one sync), raise what the eager path would have
        raised, refresh the modules' evaluation counters, and make sure the parameters still alias the flat buffers.
        Returns (nfe_forward, nfe_backward) of the last iteration, summed over blocks."""
        from .lib.layers.cnf import _inner
        words = self.stats.cpu()
        nfe = [0, 0]
        for i, blk in enumerate(self.blocks):
            for k in (0, 1):
                st = self._native.Stats(words[i, k])
                try:
                    st.raise_for_status()
                except self._native.CnfError as err:
                    raise self._native.CnfError("block %d, %s solve: %s (%r)" % (i, ("forward", "adjoint")[k], err, st)) from None
                nfe[k] += st.nfe
            _inner(blk.odefunc)._num_evals.fill_(float(words[i, 0, 1] + words[i, 1, 1] + 1))
            if not self.nets[i].flat_is_current():
                raise RuntimeError("the ODEnet's parameters no longer alias the flat buffer FusedTrainStep updates "
                                   "(module moved or re-assigned?): build a new FusedTrainStep")
        return tuple(nfe)

    @property
    def step_count(self):
        return int(self.adam_state[0][0].item())

    def state_dict(self):
        """Optimizer state (Adam moments and step counts per block) for checkpoint / resume."""
        return {"exp_avg": [t.clone() for t in self.exp_avg], "exp_avg_sq": [t.clone() for t in self.exp_avg_sq],
                "step": [int(s[0].item()) for s in self.adam_state], "hyper": dict(self.hyper)}

    def load_state_dict(self, state):
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(state["exp_avg"]) + list(state["exp_avg_sq"])):
            dst.copy_(src)
        for s, n in zip(self.adam_state, state["step"]):
            s.copy_(torch.tensor([int(n), 0], dtype=torch.int32))


class ConcurrentTrials:
    """Several independent trainings (hyper-parameter sweeps, seeds: how the reference's Ray Tune driver uses gu_maf) on
    ONE GPU at the same time.  A batch-global solve at B=2048 is 128 workgroups, i.e. half of an MI355X's 256 compute
    units, and its time is a dependent chain, not throughput: two trials on two streams finish in the time of one.

    `trials`: FusedTrainStep objects built with graph=True.  Every trial's iteration is replayed on a stream of its own.
    The solves synchronise their workgroups across the grid, so ALL workgroups of ALL concurrently running solves must
    be resident: the constructor checks sum(ceil(B_i / 16)) <= compute units.  (Should residency fail anyway, e.g. with
    other work on the GPU, the spins are bounded and .check() raises a sync-timeout CnfError: nothing hangs.)

    Best effort: ROCm multiplexes a process's streams onto a few hardware queues and work on streams that share one is
    serialized.  The constructor looks for streams whose graph replays overlap (`.replay`); in a process that has
    already used many streams and graphs none may be left, and the trials then simply run one after the other
    (correct, no faster).  One PROCESS per trial, two processes per GPU, always overlaps (separate queues)."""

    def __init__(self, trials):
        self.trials = list(trials)
        if not self.trials or any(t.graph is None for t in self.trials):
            raise ValueError("ConcurrentTrials needs FusedTrainStep objects built with graph=True")
        if any(t.spread for t in self.trials):
            raise ValueError("ConcurrentTrials needs FusedTrainStep objects built with spread=False: a spread adjoint "
                             "occupies every compute unit and cannot be resident next to another trial's solve")
        dev = self.trials[0].static_x.device
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        need = sum((t.static_x.shape[0] + 15) // 16 for t in self.trials)
        if need > cus:
            raise ValueError("%d concurrently resident workgroups needed, the device has %d compute units" % (need, cus))
        # graph replays when a set of streams is found on which replays overlap, else eager launches of the same kernels
        # (measured to overlap where replays do not; costs ~0.1 ms of host time per trial and iteration)
        self.streams = self._concurrent_streams(dev, len(self.trials))
        self.replay = self.streams is not None
        if not self.replay:
            self.streams = [torch.cuda.Stream(device=dev) for _ in self.trials]

    @staticmethod
    def _concurrent_streams(dev, n, candidates=12, spin_cycles=400_000, chain=4):
        """HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and graph replays on
        two streams that share one run strictly one after the other (measured: 49 ms instead of 28 ms per pair of
        iterations) although single kernels on the same two streams still overlap.  Which queue a stream gets is not
        visible, so measure with the real mechanism: a stream joins the set only if the replay of a small graph (a
        chain of spin kernels) on it overlaps in time with a replay on every stream already chosen."""
        graphs = []
        for _ in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(chain):
                    torch.cuda._sleep(spin_cycles)
            graphs.append(g)

        def replay_ms(streams):
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream(dev)
            a.record()
            for st, g in zip(streams, graphs):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    g.replay()
            for st in streams:
                cur.wait_stream(st)
            b.record()
            torch.cuda.synchronize(dev)
            return a.elapsed_time(b)

        cands = [torch.cuda.Stream(device=dev) for _ in range(candidates)]
        replay_ms(cands[:1])
        one = min(replay_ms(cands[:1]) for _ in range(2))
        chosen = [cands[0]]
        import os
        import sys
        verbose = bool(os.environ.get("CNF_DEBUG_STREAMS"))
        if verbose:
            print("[ConcurrentTrials] one replay %.3f ms" % one, file=sys.stderr)
        for c in cands[1:]:
            if len(chosen) == n:
                break
            times = [min(replay_ms([s, c]) for _ in range(2)) for s in chosen]
            if verbose:
                print("[ConcurrentTrials] candidate with chosen: %s ms" % ["%.3f" % t for t in times], file=sys.stderr)
            if all(t < 1.35 * one for t in times):
                chosen.append(c)
        return chosen if len(chosen) == n else None

    def __call__(self, batches):
        """One iteration of every trial; `batches[i]` feeds trial i.  Returns the trials' (device) loss tensors."""
        cur = torch.cuda.current_stream()
        losses = []
        for t, st, x in zip(self.trials, self.streams, batches):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                if self.replay:
                    losses.append(t(x))
                else:
                    t.static_x.copy_(x, non_blocking=True)
                    losses.append(t._iteration())
        for st in self.streams:
            cur.wait_stream(st)
        return losses

    def check(self):
        return [t.check() for t in self.trials]


def parse_args(argv=None):
    p = argparse.ArgumentParser("FFJORD CNF on the Gaussian-uniform mixture (MI355X HIP path)")
    p.add_argument("--gu_num", type=int, default=8, help="1 = unimodal, 8 = multimodal")
    p.add_argument("--cuda", type=int, default=0, help="device index")
    p.add_argument("--niters", type=int, default=50000)
    p.add_argument("--batch_size", type=int, default=2048)
    p.add_argument("--log_interval", type=int, default=1000)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--dims", type=str, default="64-64-64")
    p.add_argument("--num_blocks", type=int, default=1)
    p.add_argument("--layer_type", type=str, default="concatsquash",
                   choices=["ignore", "concat", "concatcoord", "concat_v2", "squash", "concatsquash"])
    p.add_argument("--nonlinearity", type=str, default="tanh", choices=["tanh", "softplus", "elu", "relu"])
    p.add_argument("--residual", type=eval, default=False, choices=[True, False])
    p.add_argument("--batch_norm", type=eval, default=False, choices=[True, False])
    p.add_argument("--bn_lag", type=float, default=0.0)
    p.add_argument("--time_length", type=float, default=0.5)
    p.add_argument("--train_T", type=eval, default=True, choices=[True, False])
    p.add_argument("--divergence_fn", type=str, default="brute_force", choices=["brute_force", "approximate"])
    p.add_argument("--solver", type=str, default="dopri5", choices=["dopri5", "rk4", "euler", "midpoint"])
    p.add_argument("--atol", type=float, default=1e-5)
    p.add_argument("--rtol", type=float, default=1e-5)
    p.add_argument("--step_size", type=float, default=None)
    # ffjord/train_misc.py::set_cnf_options: what evaluation-mode solves use (default: the training solver / tolerances)
    p.add_argument("--test_solver", type=str, default=None, choices=["dopri5", "rk4", "euler", "midpoint"])
    p.add_argument("--test_atol", type=float, default=None)
    p.add_argument("--test_rtol", type=float, default=None)
    p.add_argument("--rademacher", type=eval, default=False, choices=[True, False], help="Hutchinson probe: Rademacher instead of Gaussian")
    p.add_argument("--adjoint_error_control", type=str, default="state_only", choices=["state_only", "all"],
                   help="which tensors of the adjoint state steer the backward solve: (z, logp) as the reference's length-2 "
                        "tolerance lists do (default), or the per-row adjoints as well (CNF_FLAG_ADJ_CONTROL_ALL; not in the reference)")
    p.add_argument("--l1int", type=float, default=None, help="coefficient of the l1 kinetic regulariser (ffjord/train_misc.py's "
                   "flag: integral of the batch mean of mean_i |dz_i/dt|)")
    p.add_argument("--l2int", type=float, default=None, help="coefficient of the l2 kinetic regulariser (||dz/dt|| / sqrt(d))")
    p.add_argument("--eval_size", type=int, default=100000)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--save", type=str, default="experiments/gu_ffjord")
    p.add_argument("--resume", type=str, default=None, help="checkpoint written by a previous run (ffjord_checkpoint.pth)")
    p.add_argument("--max_seconds", type=float, default=0.0, help="stop (with a checkpoint) after this much wall time; 0 = no limit")
    p.add_argument("--eval_logp_size", type=int, default=20000, help="fresh data points for the E[log p] estimate at log points")
    p.add_argument("--fused", type=int, default=1, help="1: FusedTrainStep (library kernels only, one hipGraph replay per "
                   "iteration); 0: autograd + torch.optim.Adam loop")
    p.add_argument("--graphed", type=int, default=0, help="with --fused 0 (or a configuration FusedTrainStep does not cover, e.g. a "
                   "fixed-grid --solver): 1 = capture the autograd + Adam iteration as one hipGraph (GraphedTrainStep) instead of "
                   "launching it eagerly; needs a model without host-side state in its forward (no --batch_norm) and no --resume")
    p.add_argument("--device_sampler", type=int, default=1, help="1: draw the training and evaluation data on the GPU "
                   "(cnf_gu_sample, same distribution as gu.py's host sampler, another random stream); 0: the reference's "
                   "numpy sampler on the host + a copy per iteration")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("gu_ffjord needs a HIP device: the CNF integrator is a gfx950 kernel and has no CPU fallback")
    device = torch.device("cuda:%d" % args.cuda)
    torch.cuda.set_device(device)
    makedirs(args.save)
    logger = get_logger(os.path.join(args.save, "logs"))
    logger.info(str(args))
    data = make_gu(args.gu_num, device="cpu", seed=args.seed, extend_dim=True)
    torch.manual_seed(args.seed)
    from .lib.layers.wrappers.cnf_regularization import l1_regularzation_fn, l2_regularzation_fn
    reg_fns, reg_coeffs = [], []   # in the order of the original's REGULARIZATION_FNS table
    for coeff, fn in ((args.l1int, l1_regularzation_fn), (args.l2int, l2_regularzation_fn)):
        if coeff is not None:
            reg_fns.append(fn)
            reg_coeffs.append(coeff)
    model = build_model(2, tuple(int(d) for d in args.dims.split("-")), args.num_blocks, args.time_length, args.train_T,
                        args.divergence_fn, args.solver, args.atol, args.rtol, layer_type=args.layer_type,
                        residual=args.residual, batch_norm=args.batch_norm, bn_lag=args.bn_lag, rademacher=args.rademacher,
                        nonlinearity=args.nonlinearity, regularization_fns=tuple(reg_fns)).to(device)
    odenets = [blk.odefunc.odefunc.diffeq for blk in model.chain if isinstance(blk, CNF)]
    if reg_fns:
        if args.solver != "dopri5" or args.dims != "64-64-64" or args.adjoint_error_control != "state_only":
            raise SystemExit("--l1int / --l2int: built for dopri5, --dims 64-64-64 and the default adjoint error control")
        args.graphed = 0   # (the regularisers' states are read by the loss between the solves: FusedTrainStep captures that, the
        #                     autograd loop runs eagerly)
    if args.fused and (args.layer_type in ("concat", "concatcoord") or args.batch_norm):
        logger.info("autograd + torch.optim.Adam loop (FusedTrainStep updates chains of CNF blocks whose parameters have places "
                    "of their own in the kernels' layout: not 'concat' layers, no batch norm between blocks)")
        args.fused = 0
    for block in model.chain:
        if isinstance(block, CNF):
            block.adjoint_error_control = args.adjoint_error_control
            block.test_solver = args.test_solver if args.test_solver else args.solver
            block.test_atol = args.test_atol if args.test_atol else args.atol
            block.test_rtol = args.test_rtol if args.test_rtol else args.rtol
            if args.step_size:
                block.solver_options["step_size"] = args.step_size
    logger.info("parameters: %d" % count_parameters(model))
    model.train()
    fused = optimizer = graphed = None
    on_device = data.device_sampler(device) if args.device_sampler else None
    train_sampler = on_device if (args.fused and on_device is not None) else None
    if args.fused:
        x = data.get_sample(args.batch_size).to(device)
        fused = FusedTrainStep(model, x, lr=args.lr, weight_decay=args.weight_decay, sampler=train_sampler, reg_coeffs=reg_coeffs or None)
    else:
        graphed_ok = bool(args.graphed) and not args.batch_norm and not args.resume
        optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay,
                                     **(dict(capturable=True) if graphed_ok else dict(fused=True)))
        for block in model.chain:          # one training per GPU: the adjoint may use both halves of the chip
            if isinstance(block, CNF):
                block.spread_adjoint = True
        if graphed_ok:
            x = (on_device or data).get_sample(args.batch_size).to(device)
            graphed = GraphedTrainStep(model, optimizer, x, warmup_batches=[x])
    start_itr = 0
    if args.resume:
        ck = torch.load(args.resume, map_location=device, weights_only=True)
        model.load_state_dict(ck["model"])
        start_itr = int(ck["itr"])
        # a fresh sample stream for the rest of the run -- BEFORE the fused step is rebuilt: the captured graph holds the
        # sampler's seed as a by-value kernel argument, so a seed changed afterwards would never reach training (the resumed
        # run would replay the batches the original run had already trained on)
        data = make_gu(args.gu_num, device="cpu", seed=args.seed + start_itr, extend_dim=True)
        if on_device is not None:
            on_device = data.device_sampler(device)
            train_sampler = on_device if args.fused else None
        if fused is not None:
            fused = FusedTrainStep(model, x, lr=args.lr, weight_decay=args.weight_decay, sampler=train_sampler,
                                   reg_coeffs=reg_coeffs or None)   # parameters were re-assigned: re-alias
            fused.load_state_dict(ck["optimizer"])
        else:
            optimizer.load_state_dict(ck["optimizer"])
        logger.info("resumed from %s at iteration %d" % (args.resume, start_itr))

    def checkpoint(itr):
        torch.save({"model": model.state_dict(), "itr": itr,
                    "optimizer": fused.state_dict() if fused is not None else optimizer.state_dict()},
                   os.path.join(args.save, "ffjord_checkpoint.pth"))
        torch.save(model.state_dict(), os.path.join(args.save, "ffjord_state_dict.pth"))

    def mean_log_likelihood():
        from .parallel_eval import model_log_prob
        was_training = model.training
        model.eval()
        for b in model.chain:
            b.tile_rows = 16          # evaluation: every 16-row tile controls its own step
        with torch.no_grad(), checked_solves(model):
            lp = model_log_prob(model, (on_device or data).get_sample(args.eval_logp_size).to(device)).mean().item()
        for b in model.chain:
            b.tile_rows = 0
        model.train(was_training)
        return lp

    t_start = t_last = time.time()
    itr_last = start_itr
    for itr in range(start_itr + 1, args.niters + 1):
        x = None if train_sampler is not None else (on_device or data).get_sample(args.batch_size).to(device)
        log_now = itr % args.log_interval == 0 or itr == start_itr + 1 or itr == args.niters
        if fused is not None:
            loss = fused(x)
            if log_now:
                nfe_f, nfe_b = fused.check()
        elif graphed is not None:
            loss = graphed(x)
            if log_now:
                graphed.check()
                nfe_f, nfe_b = graphed.last_nfe
        else:
            optimizer.zero_grad(set_to_none=True)
            loss = compute_loss(model, x)
            if reg_coeffs:
                loss = loss + get_regularization(model, reg_coeffs)
            nfe_f = count_nfe(model)
            loss.backward()
            nfe_b = count_nfe(model) - nfe_f
            optimizer.step()
        if log_now:
            torch.cuda.synchronize()
            now = time.time()
            per_iter = (now - t_last) / max(1, itr - itr_last)
            wd = evaluate_w_distance(model, on_device or data, args.eval_size, device=device)
            lp = mean_log_likelihood()
            logger.info("iter %06d | loss %.6f | NFE fwd %d bwd %d | %.3f ms/iter = %.1f iters/s | W-distance %.4f | E[log p] %.4f | wall %.0f s"
                        % (itr, loss.item(), nfe_f, nfe_b, 1e3 * per_iter, 1.0 / per_iter, wd, lp, now - t_start))
            checkpoint(itr)
            t_last, itr_last = time.time(), itr
        if args.max_seconds and time.time() - t_start > args.max_seconds:
            torch.cuda.synchronize()
            if fused is not None:
                fused.check()          # never overwrite the last good checkpoint after a failed solve
            if graphed is not None:
                graphed.check()
            checkpoint(itr)
            logger.info("time limit reached at iteration %d: checkpoint written" % itr)
            break
    return model


if __name__ == "__main__":
    main()
