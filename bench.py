#!/usr/bin/env python3
"""bench.py -- FFJORD CNF training throughput on MI355X (BASELINE.json metric: CNF train iters/s, B=2048, d=2).

One "step" = one full training iteration of `gu_maf --gu_num 8` (BASELINE configs[1]): forward dopri5 solve of
the CNF on a (2048, 2) batch, negative log-likelihood, adjoint backward solve, Adam update.  Inputs are
synthetic batches of the gu_num=8 Gaussian-uniform mixture, generated before the timed region and resident in HBM;
weights are nn.Linear default init under torch.manual_seed(1 + rank).

N > 1 (launched by torch.distributed.run, one rank per GPU): training does not shard (batch-global step control),
so every rank runs an independent trial -- the reference's own multi-GPU use (one gu_num trial per cuda:i) -- and
`value` is the total iterations/s over ranks ("weak").  The part of the path that does shard, large-N
log-likelihood evaluation with one RCCL all-reduce, is timed next to it and reported under "eval".

Extra objects on the JSON line: "roofline" (dominant kernel = the adjoint solve, MFMA fp32 peak), "cpu_baseline"
(the oracle, i.e. the PyTorch-CPU restatement of the reference path, timed on the host cores on the same batches),
"late_training" (same step on the stiff trained gu_num=8 field, where NFE is ~100x higher) and "eval".
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_EVAL_D2 = 2 * 8448 + 2 * 2 * (8448 - 2 * 64)   # 50 176 FLOP per sample-eval (d=2, H=64^3, exact div)
EVAL_LAUNCH_ROWS = 1 << 22   # rows per tile-local launch (the tail of a launch idles CUs: 2^18 -> 2^22 rows is +4 %)
F_EVAL_D1 = 2 * 8320 + 2 * 1 * (8320 - 1 * 64)   # 33 152 (d=1 fixtures)
F_EVAL_D2_H256 = 2 * 132096 + 2 * 2 * (132096 - 2 * 256)   # 790 528 per block (d=2, H=256^3, exact divergence)
PEAK_FP32_MATRIX_TFLOPS = 157.3                  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=2048)
    p.add_argument("--eval-rows", type=int, default=1 << 24,
                   help="rows of the sharded log-likelihood evaluation and of the sharded sampling (BASELINE config 3: 2^24)")
    p.add_argument("--late-steps", type=int, default=10)
    p.add_argument("--cpu-iters", type=int, default=30, help="iterations per pass of the cpu_baseline window (rank 0, N=1)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU time to spend on the cpu_baseline sample")
    p.add_argument("--mode", type=str, default="fused", choices=["fused", "graph", "eager"],
                   help="fused: gu_ffjord.FusedTrainStep (solves + cnf_nll_loss + cnf_adam_step, one hipGraph replay per "
                        "iteration); graph: autograd + torch.optim.Adam loop captured as one hipGraph; eager: that loop, launched eagerly")
    p.add_argument("--extras", type=str, default="roofline,late,eval,d16,wide,rk4,pair,cpu,config1,repeats",
                   help="comma list of the additional measurements to run: roofline, late, eval, d16, wide, pair, cpu, config1, "
                        "rk4, repeats ('' = none)")
    p.add_argument("--repeats", type=int, default=5, help="extra repeats of the timed loop (fresh seed-1 model each) for the spread keys")
    p.add_argument("--cpu-late-iters", type=int, default=2,
                   help="CPU-oracle iterations on the late_training field (tens of seconds each; 0 = skip)")
    p.add_argument("--config1-cpu-iters", type=int, default=100, help="CPU-oracle iterations of BASELINE config 1 (gu_num=1)")
    p.add_argument("--dry-run", action="store_true",
                   help="rehearse launch, rendezvous and the JSON contract without a GPU (CPU test of the --gpus N spawn path)")
    return p.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh processes of this script, one rank per GPU, the way
    the reference runs one trial per cuda:i (/root/reference/hyperopt_wgan.py:642-643,739).  Runs BEFORE anything in this
    process has touched HIP (no torch.cuda call so far) and never execs: the ranks are children, rank 0's stdout (the JSON
    line) is this process's stdout, and the exit code is the first non-zero one of the children."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:          # a dead rank leaves the others waiting in a collective
                    other.terminate()
        time.sleep(0.05)
    return rc


def embed_trained_multimodal(model):
    """Late-training stand-in: the reference's trained gu_num=8 weights (d=1 checkpoint fixture) placed in the d=2
    net with the nuisance coordinate decoupled (zero column / zero row): same stiff field, NFE ~1.3-1.7k."""
    import numpy as np
    a = np.load(os.path.join(ROOT, "tests", "golden", "ffjord_multimodal.npz"))
    sd = model.state_dict()
    for k in sd:
        if k not in a.files:
            continue
        src = torch.from_numpy(a[k])
        if k.endswith("layers.0._layer.weight"):
            sd[k].zero_()
            sd[k][:, :1] = src
        elif ".layers.3." in k:
            sd[k].zero_()
            if src.dim() == 2 and src.shape[1] == 64:
                sd[k][:1, :] = src
            elif src.dim() == 2:
                sd[k][:1, :] = src
            else:
                sd[k][:1] = src
        else:
            sd[k].copy_(src)
    model.load_state_dict(sd)
    return model


def train_step(model, opt, x, gu_ffjord):
    opt.zero_grad(set_to_none=True)
    loss = gu_ffjord.compute_loss(model, x)
    loss.backward()
    opt.step()
    blk = model.chain[0]
    return loss, blk.last_stats["forward"], blk.last_stats["backward"]


def dry_run(args, world, rank):
    """No GPU: the launch / rendezvous / one-JSON-line contract only (tests/test_bench_launch.py)."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        assert t.item() == world * (world + 1) / 2
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "CNF train iters/s (B=2048, d=2)", "value": None, "unit": "iters/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True}), flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the CNF path is a HIP kernel with no CPU fallback")
    # BENCH_DIST_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks then share
    # devices; RCCL refuses that).  The driver's runs use the default: nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local if backend == "nccl" else local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    # ranks that share a GPU (the gloo rehearsal only) must not launch adjoints that occupy every compute unit
    spread = not (backend != "nccl" and world > torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gan_flow_amd import _native, gu_ffjord, parallel_eval
    from gan_flow_amd.gu import make_gu
    _native.load_library()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def fused_run(batches_):
        """W warm-up iterations (real steps, then the graph capture), then exactly K timed iterations of a fresh seeded model:
        (model, seconds [max over ranks], (nfe_forward, nfe_backward) of the last iteration)."""
        torch.manual_seed(1 + rank)
        model_ = gu_ffjord.build_model(2).to(dev).train()
        # library kernels only: forward solve, adjoint solve (loss + cotangents inside), slab sum + Adam
        step_ = gu_ffjord.FusedTrainStep(model_, batches_[0], lr=1e-3, weight_decay=1e-5, graph=True, spread=spread,
                                         warmup_batches=batches_[:max(1, args.warmup)])
        barrier()
        t0_ = time.perf_counter()
        for x_ in batches_[args.warmup:]:
            step_(x_)
        barrier()
        el_ = max_over_ranks(time.perf_counter() - t0_)
        return model_, el_, step_.check()

    # ---------------------------------------------------------------- headline: training iterations/s
    data = make_gu(8, device="cpu", seed=1 + rank, extend_dim=True)
    batches = [data.get_sample(args.batch).to(dev) for _ in range(args.warmup + args.steps)]
    graphed = None
    if args.mode == "fused":
        model, elapsed, last_nfe = fused_run(batches)
        nfe_f, nfe_b = [args.steps * v for v in last_nfe]     # NFE of the last iteration (device-side words)
    elif args.mode == "graph":
        torch.manual_seed(1 + rank)
        model = gu_ffjord.build_model(2).to(dev).train()
        # the autograd loop as one hipGraph replay (gu_ffjord.GraphedTrainStep); warm-up iterations run eagerly
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
        graphed = gu_ffjord.GraphedTrainStep(model, opt, batches[0], batches[:max(1, args.warmup)])
        barrier()
        t0 = time.perf_counter()
        for x in batches[args.warmup:]:
            graphed(x)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        graphed.check()
        nfe_f, nfe_b = [int(v) for v in graphed.nfe.tolist()]
        for blk in model.chain:
            blk.sync_stats = True
    else:
        torch.manual_seed(1 + rank)
        model = gu_ffjord.build_model(2).to(dev).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        for x in batches[:args.warmup]:
            train_step(model, opt, x, gu_ffjord)
        nfe_f = nfe_b = 0
        barrier()
        t0 = time.perf_counter()
        for x in batches[args.warmup:]:
            _, sf, sb = train_step(model, opt, x, gu_ffjord)
            nfe_f += sf.nfe
            nfe_b += sb.nfe
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
    value = world * args.steps / elapsed
    out = {
        "metric": "CNF train iters/s (B=2048, d=2)", "value": round(value, 3), "unit": "iters/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "FFJORD CNF gu_maf --gu_num 8, B=%d, d=2, ODEnet 2-64-64-64-2 concatsquash/tanh, dopri5 "
                               "rtol=atol=1e-5, T=0.5, exact divergence, adjoint backward, Adam; weights from seed-1 "
                               "init (first %d iterations of training)" % (args.batch, args.warmup + args.steps),
                   "global_batch": args.batch * world, "parallelism": "independent trial per GPU (replicas only)",
                   "launch": {"fused": "3 library kernels (fwd solve; adjoint solve with the NLL cotangents and loss formed inside, two workgroups per tile; gradient-slab sum + Adam), one hipGraph replay per iteration",
                              "graph": "autograd + torch.optim.Adam loop, one hipGraph replay per iteration",
                              "eager": "autograd + torch.optim.Adam loop, eager"}[args.mode],
                   "mean_nfe_forward": round(nfe_f / args.steps, 1), "mean_nfe_backward": round(nfe_b / args.steps, 1)},
    }

    extras = set(e for e in args.extras.split(",") if e)
    ctx = dict(args=args, world=world, rank=rank, dev=dev, batches=batches, model=model, barrier=barrier, spread=spread,
               max_over_ranks=max_over_ranks, fused_run=fused_run)
    if "pair" in extras and not spread:
        # ranks share a GPU (gloo rehearsal on a box with fewer GPUs than ranks): two more trials per rank would need more
        # co-resident workgroups than the device has compute units -- the solves' grid reductions would time out
        out["two_trials_per_gpu"] = {"skipped": "ranks share a GPU in this run (BENCH_DIST_BACKEND rehearsal)"}
    elif "pair" in extras:
        # first: whether two streams of one process overlap depends on how many streams / graphs the process has used
        # so far (ROCm multiplexes streams onto 4 hardware queues); after the measurements below they no longer do
        out["two_trials_per_gpu"] = measure_pair(ctx)
    if "repeats" in extras and args.mode == "fused" and args.repeats > 0:
        # the timed region is K x ~0.2 ms: one scheduler hiccup away from a wrong number.  The same W + K loop again, from
        # the same seed and batches (identical work), `repeats` more times; `value` above stays the first run's.
        ms = sorted(1e3 * fused_run(batches)[1] / args.steps for _ in range(args.repeats))
        out["repeats"] = {"n": len(ms), "ms_per_step_min": round(ms[0], 4), "ms_per_step_median": round(ms[len(ms) // 2], 4),
                          "ms_per_step_max": round(ms[-1], 4), "value_at_median": round(world * 1e3 / ms[len(ms) // 2], 1),
                          "note": "same W warm-up + K timed iterations repeated from the same seed and batches; `value` is the first run"}
    if "config1" in extras:
        out["config1"] = measure_config1(ctx)
    if "roofline" in extras and rank == 0:
        out["roofline"] = measure_roofline(ctx)
        out["roofline_forward"] = forward_roofline(ctx["model"], ctx["batches"][-1])
    if "late" in extras:
        out["late_training"] = measure_late(ctx)
    if "eval" in extras:
        out["eval"] = measure_eval(ctx)
    if "d16" in extras:
        out["d16_hutchinson"] = measure_d16(ctx)
    if "wide" in extras:
        out["h256_4blocks"] = measure_wide(ctx)
    if "rk4" in extras:
        out["rk4_training"] = measure_rk4(ctx)
    if "cpu" in extras and rank == 0 and world == 1 and args.cpu_iters > 0:
        out["cpu_baseline"] = measure_cpu(ctx)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measure_roofline(ctx):
    """Dominant kernel = the adjoint solve of the timed step (seed-1 init, NFE 14).  `traffic`: HBM bytes per launch from the
    separate rocprofv3 --pmc passes of this round (profiles/), not measurable in-process."""
    args = ctx["args"]
    res = adjoint_roofline(ctx["model"], ctx["batches"][-1], ctx["spread"])
    traffic = None
    import glob
    tpath = (sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bwd_traffic.json"))) or [""])[-1]   # the latest round's passes
    if os.path.exists(tpath) and args.batch == 2048:
        with open(tpath) as fh:
            traffic = json.load(fh)["hbm_bytes_per_launch"]
    res["traffic"] = traffic
    return res


def measure_late(ctx):
    """The same training step on the stiff trained gu_num=8 field (NFE ~1.7k per solve)."""
    from gan_flow_amd import gu_ffjord
    args, world, rank, dev, batches = ctx["args"], ctx["world"], ctx["rank"], ctx["dev"], ctx["batches"]
    torch.manual_seed(1 + rank)
    late = embed_trained_multimodal(gu_ffjord.build_model(2)).to(dev).train()
    step = gu_ffjord.FusedTrainStep(late, batches[0], lr=1e-5, weight_decay=1e-5, graph=False, spread=ctx["spread"])
    step(batches[0])
    ctx["barrier"]()
    t0 = time.perf_counter()
    lf = lb = 0
    per_step = []
    for i in range(args.late_steps):
        t1 = time.perf_counter()
        step(batches[1 + i % (len(batches) - 1)])
        f, b = step.check()          # one host sync per step: it is ~16 ms long, the sync costs ~10 us
        per_step.append(1e3 * (time.perf_counter() - t1))
        lf += f
        lb += b
    ctx["barrier"]()
    lel = ctx["max_over_ranks"](time.perf_counter() - t0)
    per_step.sort()
    res = {"value": round(world * args.late_steps / lel, 4), "unit": "iters/s", "ms_per_step": round(1e3 * lel / args.late_steps, 2),
           "steps": args.late_steps, "rank0_ms_per_step_min": round(per_step[0], 2),
           "rank0_ms_per_step_median": round(per_step[len(per_step) // 2], 2), "rank0_ms_per_step_max": round(per_step[-1], 2),
           "mean_nfe_forward": round(lf / args.late_steps, 1), "mean_nfe_backward": round(lb / args.late_steps, 1),
           "workload": "same step on the reference's trained gu_num=8 field (checkpoint weights embedded in d=2)"}
    if rank == 0:
        res["roofline"] = adjoint_roofline(late, batches[1], ctx["spread"], reps=3)
        res["roofline_forward"] = forward_roofline(late, batches[1], reps=3)
        if world == 1 and args.cpu_late_iters > 0 and "cpu" in set(args.extras.split(",")):
            # the CPU oracle on the same field and batches: where BASELINE config 2 spends > 99 % of its iterations
            import oracle
            ncores = min(16, os.cpu_count() or 1)
            torch.set_num_threads(ncores)
            torch.manual_seed(1)
            om = embed_trained_multimodal(oracle.build_cnf(2)).train()
            oopt = torch.optim.Adam(om.parameters(), lr=1e-5, weight_decay=1e-5)
            t0 = time.perf_counter()
            for i in range(args.cpu_late_iters):
                xb = batches[1 + i].cpu()
                oopt.zero_grad(set_to_none=True)
                z, dl = om(xb, torch.zeros(xb.shape[0], 1))
                (-(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()).backward()
                oopt.step()
            cel = time.perf_counter() - t0
            st = om.chain[0].stats
            res["cpu_baseline"] = {"value": round(args.cpu_late_iters / cel, 5), "unit": "iters/s", "cores": ncores, "kind": "port",
                                   "sample": "%d training iteration(s) of the PyTorch-CPU oracle on the same field and batch "
                                             "(%.1f s; forward %d + %d attempts, adjoint %d + %d), torch.set_num_threads(%d)"
                                             % (args.cpu_late_iters, cel, st["forward"]["n_accepted"], st["forward"]["n_rejected"],
                                                st["backward"]["n_accepted"], st["backward"]["n_rejected"], ncores)}
            res["vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    return res


def adjoint_roofline(model, x, spread, reps=20):
    """The adjoint solve kernel of `model` on batch `x`: algorithmic FLOP per launch (rows x NFE x 3 x F_eval, NFE as the reference
    counts it; the state work a two-workgroups-per-tile launch does twice is counted once) / mean kernel duration, measured
    with HIP events recorded by the library on the launch stream right around the kernel (cnf_debug_kernel_timing)."""
    from gan_flow_amd import _native
    blk = model.chain[0]
    fn = blk.odefunc.odefunc
    net = fn.net_desc()
    params = fn.diffeq.flat_parameters().detach()
    opts = _native.make_opts("dopri5", blk.rtol, blk.atol, flags=_native.FLAG_SPREAD if spread else 0)   # as the timed step launches it
    T = float(blk.sqrt_end_time.detach() ** 2)
    rows = x.shape[0]
    z1, l1, _ = _native.solve_fwd(net, params, x, None, 0.0, T, opts)
    gz, gl = z1 / rows, torch.full_like(l1, 1.0 / rows)
    _native.solve_bwd(net, params, z1, l1, gz, gl, 0.0, T, opts)
    _native.kernel_timing(True)
    ms = call_ms = 0.0
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        st = _native.solve_bwd(net, params, z1, l1, gz, gl, 0.0, T, opts)[3]
        b.record()
        ms += _native.last_kernel_ms() / reps
        torch.cuda.synchronize()
        call_ms += a.elapsed_time(b) / reps
    _native.kernel_timing(False)
    st = _native.Stats(st)
    flops = rows * st.nfe * 3 * F_EVAL_D2
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "dopri5_bwd_split_kernel<2,false,%s>" % ("true,true" if spread else "false,false"),
            "achieved": round(achieved, 4), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 5), "launch_ms": round(ms, 4), "call_ms": round(call_ms, 4),
            "nfe": st.nfe, "evals_run": st.evals_run, "us_per_evaluation": round(1e3 * ms / st.evals_run, 3),
            "flop_per_launch": flops,
            "peak_note": "dense fp32 matrix peak (the hidden-layer products run on the bf16 pipe with three-piece operands: six "
                         "bf16 MFMAs of 16 clocks do the work of sixteen fp32 ones of 32; FLOP are counted as fp32 work)"}


def forward_roofline(model, x, reps=20):
    """The forward solve kernel of the timed step, as adjoint_roofline: rows x NFE x F_eval / mean kernel duration (HIP events
    recorded by the library around the kernel).  It is the kernel furthest below its roof: 128 workgroups on 256 compute units,
    a dependent chain per evaluation (three waves per SIMD: value stream, two tangent streams; cnf_split_fwd_tw.h)."""
    from gan_flow_amd import _native
    blk = model.chain[0]
    fn = blk.odefunc.odefunc
    net = fn.net_desc()
    params = fn.diffeq.flat_parameters().detach()
    opts = _native.make_opts("dopri5", blk.rtol, blk.atol)
    T = float(blk.sqrt_end_time.detach() ** 2)
    rows = x.shape[0]
    _native.solve_fwd(net, params, x, None, 0.0, T, opts)
    _native.kernel_timing(True)
    ms = 0.0
    for _ in range(reps):
        st = _native.solve_fwd(net, params, x, None, 0.0, T, opts)[2]
        ms += _native.last_kernel_ms() / reps
    _native.kernel_timing(False)
    st = _native.Stats(st)
    flops = rows * st.nfe * F_EVAL_D2
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "dopri5_split_tw_kernel<2,false,2>", "achieved": round(achieved, 4), "peak": PEAK_FP32_MATRIX_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 5), "launch_ms": round(ms, 4), "nfe": st.nfe,
            "us_per_evaluation": round(1e3 * ms / st.nfe, 3), "flop_per_launch": flops,
            "peak_note": "dense fp32 matrix peak; FLOP counted as fp32 work (hidden layers on the bf16 pipe, three-piece operands)"}


def measure_config1(ctx):
    """BASELINE configs[0]: gu_maf --gu_num 1 (unimodal 2-D), B=2048, dopri5, default ODEnet; the GPU runs the W + K loop of
    the headline on gu_num=1 batches, the CPU oracle (rank 0, N=1) the 100 iterations the config names (README.md:41)."""
    from gan_flow_amd.gu import make_gu
    args, world, rank, dev = ctx["args"], ctx["world"], ctx["rank"], ctx["dev"]
    data = make_gu(1, device="cpu", seed=1 + rank, extend_dim=True)
    n_cpu = args.config1_cpu_iters if (rank == 0 and world == 1 and "cpu" in set(args.extras.split(","))) else 0
    xs = [data.get_sample(args.batch) for _ in range(max(args.warmup + args.steps, 5 + n_cpu))]
    _, el, (nf, nb) = ctx["fused_run"]([x.to(dev) for x in xs[:args.warmup + args.steps]])
    res = {"value": round(world * args.steps / el, 3), "unit": "iters/s", "ms_per_step": round(1e3 * el / args.steps, 4),
           "nfe_forward": nf, "nfe_backward": nb,
           "workload": "FFJORD CNF gu_maf --gu_num 1 (unimodal 2-D), B=%d, dopri5 1e-5, default ODEnet, seed-1 init" % args.batch}
    if n_cpu > 0:
        import oracle
        ncores = min(16, os.cpu_count() or 1)
        torch.set_num_threads(ncores)
        torch.manual_seed(1)
        om = oracle.build_cnf(2).train()
        oopt = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=1e-5)

        def ostep(x):
            oopt.zero_grad(set_to_none=True)
            z, dl = om(x, torch.zeros(x.shape[0], 1))
            (-(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()).backward()
            oopt.step()

        for x in xs[:5]:
            ostep(x)
        t0 = time.perf_counter()
        for x in xs[5:5 + n_cpu]:
            ostep(x)
        cel = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(n_cpu / cel, 4), "unit": "iters/s", "cores": ncores, "kind": "port",
                               "sample": "%d training iterations after 5 warm-up (the run BASELINE configs[0] names) of the "
                                         "PyTorch-CPU oracle, same seed-1 init and batches, %.1f s, torch.set_num_threads(%d)"
                                         % (n_cpu, cel, ncores)}
        res["vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    return res


def measure_d16(ctx):
    """BASELINE config 5: d=16 Gaussian mixture, Hutchinson divergence, B=2048, one independent seed per GPU."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gmm import GaussianMixture
    args, world, rank, dev = ctx["args"], ctx["world"], ctx["rank"], ctx["dev"]
    data = GaussianMixture(16, device=dev, sample_seed=1 + rank)
    torch.manual_seed(1 + rank)
    model = gu_ffjord.build_model(16, divergence_fn="approximate").to(dev).train()
    xs = [data.get_sample(args.batch) for _ in range(args.warmup + args.steps)]
    step = gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=True, spread=ctx["spread"],
                                    warmup_batches=xs[:max(1, args.warmup)])
    ctx["barrier"]()
    t0 = time.perf_counter()
    for x in xs[args.warmup:]:
        step(x)
    ctx["barrier"]()
    el = ctx["max_over_ranks"](time.perf_counter() - t0)
    nfe_f, nfe_b = step.check()
    return {"value": round(world * args.steps / el, 2), "unit": "iters/s", "ms_per_step": round(1e3 * el / args.steps, 4),
            "nfe_forward": nfe_f, "nfe_backward": nfe_b, "scaling": "weak",
            "workload": "FFJORD CNF on the synthetic d=16 8-component Gaussian mixture, B=%d, ODEnet 16-64-64-64-16, "
                        "Hutchinson divergence (Gaussian probe, fresh per iteration), dopri5 1e-5, adjoint, Adam; "
                        "seed 1+rank per GPU" % args.batch}


def measure_rk4(ctx):
    """The headline's training iteration with the fixed-grid solver north_star names for parity (`gu_maf --solver rk4
    --step_size 0.05`: 10 steps of the 3/8 rule each way, NFE 40 / 40) on the one-tile-per-workgroup fixed-grid kernels."""
    from gan_flow_amd import gu_ffjord
    args, dev = ctx["args"], ctx["dev"]
    torch.manual_seed(1 + ctx["rank"])
    model = gu_ffjord.build_model(2, solver="rk4").to(dev).train()
    for blk in model.chain:
        blk.solver_options = {"step_size": 0.05}
    xs = ctx["batches"]
    step = gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=True, spread=ctx["spread"],
                                    warmup_batches=xs[:max(1, args.warmup)])
    ctx["barrier"]()
    t0 = time.perf_counter()
    for x in xs[args.warmup:args.warmup + args.steps]:
        step(x)
    ctx["barrier"]()
    el = ctx["max_over_ranks"](time.perf_counter() - t0)
    nfe_f, nfe_b = step.check()
    return {"value": round(ctx["world"] * args.steps / el, 2), "unit": "iters/s", "ms_per_step": round(1e3 * el / args.steps, 4),
            "nfe_forward": nfe_f, "nfe_backward": nfe_b,
            "workload": "the headline's model and batches trained with --solver rk4 --step_size 0.05 (3/8 rule, fixed grid, "
                        "adjoint on the same grid), B=%d" % args.batch}


def measure_pair(ctx):
    """Two independent trials of the headline workload per GPU (gu_ffjord.ConcurrentTrials): a B=2048 solve occupies 128
    of the 256 compute units, so a sweep can run two trials in the time of one.  Reported beside the headline, never as
    it: the headline is ONE training run."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    args, world, rank, dev = ctx["args"], ctx["world"], ctx["rank"], ctx["dev"]
    out = {}
    for name, late, steps in (("early", False, args.steps), ("late", True, args.late_steps)):
        trials, feeds = [], []
        for k in range(2):
            data = make_gu(8, device="cpu", seed=101 + 2 * rank + k, extend_dim=True)
            xs = [data.get_sample(args.batch).to(dev) for _ in range(args.warmup + steps)]
            torch.manual_seed(101 + 2 * rank + k)
            model = gu_ffjord.build_model(2)
            if late:
                model = embed_trained_multimodal(model)
            model = model.to(dev).train()
            trials.append(gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=True, spread=False,
                                                   warmup_batches=xs[:1 if late else max(1, args.warmup)]))
            feeds.append(xs[args.warmup:])
        both = gu_ffjord.ConcurrentTrials(trials)
        ctx["barrier"]()
        t0 = time.perf_counter()
        for i in range(steps):
            both([feeds[0][i], feeds[1][i]])
        ctx["barrier"]()
        el = ctx["max_over_ranks"](time.perf_counter() - t0)
        nfe = both.check()
        out[name] = {"value": round(2 * world * steps / el, 2), "unit": "iters/s (sum of both trials)",
                     "ms_per_step_pair": round(1e3 * el / steps, 4), "nfe": [list(v) for v in nfe],
                     "launch": "graph replays" if both.replay else "eager launches (no stream pair found on which graph replays overlap)"}
    out["workload"] = "two concurrent trials of the headline workload per GPU, each on its own stream (one hipGraph replay per trial and iteration)"
    return out


def measure_wide(ctx):
    """BASELINE config 4 (forward path): 4 stacked CNF blocks of hidden width 256, d=2, random init seed 1.  Density
    evaluation at B=2048 with the reference's batch-global step control, and at B=65 536 with tile-local control."""
    from gan_flow_amd import gu_ffjord
    args, world, rank, dev = ctx["args"], ctx["world"], ctx["rank"], ctx["dev"]
    torch.manual_seed(1)
    model = gu_ffjord.build_model(2, dims=(256, 256, 256), num_blocks=4).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(11 + rank)
    out = {}
    for name, rows, tile in (("b2048_batch_global", 2048, 0), ("b65536_tile_local", 65536, 16)):
        for b in model.chain:
            b.tile_rows = tile
        x = torch.randn(rows, 2, generator=g, device=dev)
        zero = torch.zeros(rows, 1, device=dev)
        with torch.no_grad():
            model(x, zero)
            reps = 5
            evals = 0
            ctx["barrier"]()
            t0 = time.perf_counter()
            for _ in range(reps):
                model(x, zero)
                evals += sum(b.last_stats["forward"].row_evals for b in model.chain)
            ctx["barrier"]()
            el = ctx["max_over_ranks"](time.perf_counter() - t0)
        out[name] = {"rows_per_s": round(world * reps * rows / el, 1), "ms_per_pass": round(1e3 * el / reps, 3),
                     "nfe_per_block": [b.last_stats["forward"].nfe for b in model.chain],
                     "rank0_tflops": round(evals * F_EVAL_D2_H256 / el / 1e12, 2)}
    # training iteration of the same model (FusedTrainStep: 4 forward solves, loss, 4 adjoint solves, 4 Adam updates)
    from gan_flow_amd.gu import make_gu
    for b in model.chain:
        b.tile_rows = 0
    model.train()
    data = make_gu(8, device="cpu", seed=1 + rank, extend_dim=True)
    n_it = max(3, args.steps // 3)
    xs = [data.get_sample(args.batch).to(dev) for _ in range(2 + n_it)]
    step = gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=True, warmup_batches=xs[:2], spread=ctx["spread"])
    ctx["barrier"]()
    t0 = time.perf_counter()
    for x in xs[2:]:
        step(x)
    ctx["barrier"]()
    el = ctx["max_over_ranks"](time.perf_counter() - t0)
    nfe_f, nfe_b = step.check()
    flops = args.batch * (nfe_f + 3 * nfe_b) * F_EVAL_D2_H256
    out["train_b%d" % args.batch] = {"value": round(world * n_it / el, 2), "unit": "iters/s", "ms_per_step": round(1e3 * el / n_it, 3),
                                     "nfe_forward": nfe_f, "nfe_backward": nfe_b, "rank0_tflops": round(flops * n_it / el / 1e12, 2)}
    out["workload"] = ("4 stacked CNF blocks, ODEnet 2-256-256-256-2, exact divergence, dopri5 1e-5, random init seed 1: "
                       "log-density passes, and training iterations on gu_num=8 batches")
    return out


def measure_eval(ctx):
    """BASELINE config 3 (bounded): log-likelihood of eval_rows synthetic points, batch-sharded, one all-reduce."""
    import numpy as np
    from gan_flow_amd import gu_ffjord, parallel_eval
    args, world, rank, dev = ctx["args"], ctx["world"], ctx["rank"], ctx["dev"]
    a = np.load(os.path.join(ROOT, "tests", "golden", "ffjord_multimodal.npz"))
    em = gu_ffjord.build_model(1)
    em.load_state_dict({k: torch.from_numpy(a[k]) for k in a.files})
    em = em.to(dev).eval()
    for b in em.chain:
        b.tile_rows = 16
    # Config 3's points, addressed by GLOBAL row index (cnf_gu_sample_rows: counter-based, stateless): rows [0, N/2) are draws of
    # the training distribution, rows [N/2, N) of N(45, 23^2) (the tails).  Every world size evaluates the SAME N rows, so
    # mean_log_likelihood is comparable between the N = 1, 2, 4, 8 runs of a scaling sweep (to summation order).
    from gan_flow_amd.gu import DeviceNormalRows, make_gu
    lo, hi = parallel_eval.shard_bounds(args.eval_rows, rank, world)
    n_loc = hi - lo
    n_data = args.eval_rows // 2
    mix = make_gu(8, device="cpu", seed=1234, extend_dim=False).device_sampler(dev)
    tails = DeviceNormalRows(dev, seed=4321, mean=45.0, sigma=23.0)
    xs = torch.cat([mix.get_rows(min(lo, n_data), min(hi, n_data)),
                    tails.get_rows(max(lo, n_data) - n_data, max(hi, n_data) - n_data)])
    assert xs.shape == (n_loc, 1)
    work = {"row_evals": 0, "nfe_max": 0}

    def lp(v):
        parts = []
        for vc in torch.split(v, EVAL_LAUNCH_ROWS):   # tile-local control needs no residency: few, large launches
            parts.append(parallel_eval.model_log_prob(em, vc, chunk=EVAL_LAUNCH_ROWS))
            st = em.chain[0].last_stats["forward"]
            work["row_evals"] += st.row_evals
            work["nfe_max"] = max(work["nfe_max"], st.nfe)
        return torch.cat(parts, 0)

    sample_fn, _ = gu_ffjord.get_transforms(em)
    with torch.no_grad():
        parallel_eval.sharded_mean_log_likelihood(lp, xs[: 1 << 14])
        work["row_evals"], work["nfe_max"] = 0, 0
        ctx["barrier"]()
        t0 = time.perf_counter()
        mean_lp, rows = parallel_eval.sharded_mean_log_likelihood(lp, xs)
        ctx["barrier"]()
        eel = ctx["max_over_ranks"](time.perf_counter() - t0)
        ll_evals = work["row_evals"]
        # sampling half of config 3: eval_rows prior draws through sample_fn (reverse solve), samples stay sharded
        del xs
        ctx["barrier"]()
        t0 = time.perf_counter()
        prior = DeviceNormalRows(dev, seed=7)
        smp = parallel_eval.sharded_sample(sample_fn, args.eval_rows, 1, rank, world, dev, chunk=EVAL_LAUNCH_ROWS,
                                           prior_rows=lambda a, b: prior.get_rows(a, b))
        ctx["barrier"]()
        sel = ctx["max_over_ranks"](time.perf_counter() - t0)
        smp_mean = smp.double().mean().item()
        # the study's metric over ALL ranks' samples (model_plot.py:93-101, util.py:16-20): generated vs. fresh data rows
        t0 = time.perf_counter()
        real = mix.get_rows(lo, hi, call=1)
        wd = parallel_eval.sharded_w_distance(real, smp, args.eval_rows, args.eval_rows)
        ctx["barrier"]()
        wel = ctx["max_over_ranks"](time.perf_counter() - t0)
    tfl = ll_evals * F_EVAL_D1 / eel / 1e12
    return {"value": round(rows / eel, 1), "unit": "samples/s", "rows": rows, "scaling": "strong",
            "mean_log_likelihood": round(mean_lp, 5), "rank0_mean_nfe": round(ll_evals / max(1, n_loc), 1),
            "rank0_max_nfe": work["nfe_max"],
            "rank0_tflops": round(tfl, 3),
            "roofline": eval_roofline(tfl),
            "sampling": {"value": round(args.eval_rows / sel, 1), "unit": "samples/s", "rows": args.eval_rows,
                         "rank0_sample_mean": round(smp_mean, 4), "w_distance": wd, "w_distance_s": round(wel, 3),
                         "w_distance_note": "1-D Wasserstein distance of ALL ranks' samples against as many fresh data rows "
                                            "(two all_gathers + sort on every rank); not in the timed sampling region"},
            "data": "points and prior draws addressed by global row index (cnf_gu_sample_rows): identical for every world size",
            "workload": "log-likelihood of %d synthetic points, then %d samples, under the trained gu_num=8 CNF (d=1 "
                        "checkpoint), tile-local dopri5 1e-5, batch-sharded, one all-reduce of 2 x f64"
                        % (rows, args.eval_rows)}


def eval_roofline(tflops):
    """The tile-local evaluation kernel against the two bounds that apply to it.  Its matrix work runs on the bf16 pipe
    (three-piece operands), so the fp32 matrix peak is a yardstick for the fp32-equivalent FLOP rate, not the limit; the
    limit is vector-instruction issue.  `valu_issue.busy_frac` = SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / waves per SIMD)
    from this round's rocprofv3 --pmc pass over tools/time_eval.py (profiles/): not measurable in-process."""
    import glob
    out = {"bound": "valu_issue", "kernel": "dopri5_kernel<1,false,false,4,true>", "achieved": None, "peak": 1.0,
           "unit": "busy fraction of the vector ALU's issue slots", "frac": None,
           "fp32_equivalent": {"achieved": round(tflops, 3), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tflops / PEAK_FP32_MATRIX_TFLOPS, 4),
                               "note": "yardstick only: FLOP counted as fp32 work against the dense fp32 matrix peak; the kernel's "
                                       "hidden layers run on the bf16 pipe, which is not what bounds it"},
           "valu_issue": None}
    path = (sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_eval_sq_pmc_summary.txt"))) or [""])[-1]
    if os.path.exists(path):
        tot = {}
        with open(path) as fh:
            for line in fh:
                if "dopri5_kernel<1, false, false" in line and " total " in line:
                    parts = line.split()
                    name = [w for w in parts if w.startswith("SQ_")][0]
                    tot[name] = float(parts[parts.index("total") + 1])
        if "SQ_ACTIVE_INST_VALU" in tot and "SQ_WAVE_CYCLES" in tot:
            waves_per_simd = 2   # amdgpu_waves_per_eu of the kernel (cnf_fwd.hip, kEvalB16Waves)
            out["valu_issue"] = {"busy_frac": round(tot["SQ_ACTIVE_INST_VALU"] / (tot["SQ_WAVE_CYCLES"] / waves_per_simd), 3),
                                 "valu_per_mfma": round(tot.get("SQ_INSTS_VALU", 0.0) / max(1.0, tot.get("SQ_INSTS_MFMA", 1.0)), 2),
                                 "source": os.path.relpath(path, ROOT)}
            out["achieved"] = out["frac"] = out["valu_issue"]["busy_frac"]
    return out


def measure_cpu(ctx):
    """cpu_baseline: the oracle (PyTorch-CPU restatement of the reference path) on the same first batches from the same
    seed-1 init; the window is repeated (model and optimizer rebuilt outside the timed region) until --cpu-seconds of
    CPU work have been timed."""
    import oracle
    args, batches = ctx["args"], ctx["batches"]
    ncores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    cpu_batches = [b.cpu() for b in batches[: args.cpu_iters + 1]]
    done, cel, passes = 0, 0.0, 0
    while cel < args.cpu_seconds and passes < 64:
        torch.manual_seed(1)
        om = oracle.build_cnf(2).train()
        oopt = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=1e-5)

        def ostep(x):
            oopt.zero_grad(set_to_none=True)
            z, dl = om(x, torch.zeros(x.shape[0], 1))
            loss = -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()
            loss.backward()
            oopt.step()

        ostep(cpu_batches[0])
        t0 = time.perf_counter()
        for x in cpu_batches[1:]:
            ostep(x)
        cel += time.perf_counter() - t0
        done += len(cpu_batches) - 1
        passes += 1
    return {"value": round(done / cel, 4), "unit": "iters/s", "cores": ncores, "kind": "port",
            "sample": "%d training iterations (%d passes over the first %d iterations of the same workload: same seed-1 "
                      "init, same batches; %.1f s of CPU time) on the PyTorch-CPU oracle, torch.set_num_threads(%d)"
                      % (done, passes, len(cpu_batches) - 1, cel, ncores)}


if __name__ == "__main__":
    main()
