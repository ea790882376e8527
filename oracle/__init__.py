"""CPU oracle for the FFJORD CNF hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package, and only as the checker / the timed CPU baseline.  Nothing under `gan-flow_amd/` imports it.

What it restates (plain PyTorch CPU ops + autograd, fp32 by default, any dtype accepted):
  * the FFJORD module stack the reference drives through `ffjord.gu_ffjord` / `ffjord.gu_maf`
    (/root/reference/model_plot.py:2,72-74,93-101; /root/reference/README.md:41,57):
    ConcatSquashLinear -> ODEnet -> ODEfunc (divergence_bf / divergence_approx) -> RegularizedODEfunc
    -> CNF -> SequentialFlow, and `get_transforms`;
  * torchdiffeq==0.0.1 (/root/reference/requirements.txt:17): `odeint` with `dopri5` (adaptive, batch-global
    step control) and fixed-grid `rk4`/`euler`/`midpoint`, and `odeint_adjoint`.

Neither library is present in /root/reference (empty git submodule `ffjord/`, .gitmodules:4-6; pip
dependency not installed), so the restatement follows their published algorithms as recorded in
SURVEY.md section 2a and Appendices B and D.  It is pinned against the in-container artefacts the
reference does hold (tests/test_oracle_golden.py):
  * the two trained checkpoints `model_results/{unimodal,multimodal}/ffjord.pth` (weights cut out as
    data by tests/golden/make_fixtures.py): the stored NFE counter of the last sampling solve
    (116) is reproduced exactly by this dopri5, and the model densities integrate to 1 and match the
    analytic training density of /root/reference/gu.py:70-92;
  * the known-answer table of SURVEY.md Appendix C (fp64 DOP853 evaluation of the same weights).
The adjoint (Appendix D) has no reference artefact to pin it: it is checked for self-consistency
against autograd through an unrolled fixed-step solve ("parity unpinned" for that part).
"""
from .odeint import odeint, odeint_adjoint, SOLVERS  # noqa: F401
from .layers import (  # noqa: F401
    ConcatSquashLinear, ODEnet, ODEfunc, RegularizedODEfunc, CNF, SequentialFlow,
    divergence_bf, divergence_approx, get_transforms, build_cnf, standard_normal_logprob,
    l1_regularzation_fn, l2_regularzation_fn,
)
