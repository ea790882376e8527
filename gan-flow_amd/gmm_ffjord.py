"""FFJORD with the Hutchinson divergence estimator on the synthetic d=16 Gaussian mixture (BASELINE config 5):
`python -m gan_flow_amd.gmm_ffjord --seed S --cuda I` runs ONE independent trial on one GPU; the config's "8 seeds,
one per GPU" is 8 such processes (replicas only, no collective).  The reference's gu_maf driver has the same
structure (gu_ffjord.py); only the data, the dimension and `divergence_fn='approximate'` differ."""
import argparse
import os
import time

import torch

from .gmm import GaussianMixture
from .gu_ffjord import FusedTrainStep, build_model
from .util import count_parameters, get_logger, makedirs


def parse_args(argv=None):
    p = argparse.ArgumentParser("FFJORD CNF, Hutchinson divergence, d-dimensional Gaussian mixture (MI355X HIP path)")
    p.add_argument("--dim", type=int, default=16)
    p.add_argument("--cuda", type=int, default=0)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--niters", type=int, default=10000)
    p.add_argument("--batch_size", type=int, default=2048)
    p.add_argument("--log_interval", type=int, default=500)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--time_length", type=float, default=0.5)
    p.add_argument("--atol", type=float, default=1e-5)
    p.add_argument("--rtol", type=float, default=1e-5)
    p.add_argument("--rademacher", type=int, default=0)
    p.add_argument("--solver", type=str, default="dopri5", choices=["dopri5", "rk4", "euler", "midpoint"])
    p.add_argument("--step_size", type=float, default=None, help="fixed-grid solvers: grid spacing (default: one step over [0, T])")
    p.add_argument("--save", type=str, default="experiments/gmm_ffjord")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("gmm_ffjord needs a HIP device: the CNF integrator is a gfx950 kernel and has no CPU fallback")
    device = torch.device("cuda:%d" % args.cuda)
    torch.cuda.set_device(device)
    makedirs(args.save)
    logger = get_logger(os.path.join(args.save, "logs_seed%d" % args.seed))
    logger.info(str(args))
    data = GaussianMixture(args.dim, device=device, sample_seed=args.seed)
    torch.manual_seed(args.seed)
    model = build_model(args.dim, time_length=args.time_length, divergence_fn="approximate", solver=args.solver, atol=args.atol,
                        rtol=args.rtol, rademacher=bool(args.rademacher)).to(device).train()
    logger.info("parameters: %d | data entropy (loss floor) %.4f" % (count_parameters(model), data.entropy_estimate()))
    for blk in model.chain:
        if args.solver != "dopri5" and args.step_size:
            blk.solver_options["step_size"] = args.step_size
    # library kernels only, one hipGraph replay per iteration; with a fixed-grid solver at d = 16 the tail is cnf_nll_loss +
    # cnf_solve_bwd (fixed_bwd_split16_kernel) + cnf_adam_step
    step = FusedTrainStep(model, data.get_sample(args.batch_size), lr=args.lr, weight_decay=args.weight_decay)
    check = step.check
    t_last = time.time()
    for itr in range(1, args.niters + 1):
        loss = step(data.get_sample(args.batch_size))
        if itr % args.log_interval == 0 or itr == 1:
            nfe_f, nfe_b = check()
            now = time.time()
            logger.info("iter %06d | loss %.6f | NFE fwd %d bwd %d | %.4f s/iter" % (
                itr, loss.item(), nfe_f, nfe_b, (now - t_last) / max(1, args.log_interval if itr > 1 else 1)))
            torch.save(model.state_dict(), os.path.join(args.save, "ffjord_state_dict_seed%d.pth" % args.seed))
            t_last = time.time()
    return model


if __name__ == "__main__":
    main()
