#!/usr/bin/env python3
"""Cut weight fixtures out of the reference's trained FFJORD checkpoints WITHOUT unpickling them.

Source files (read-only, only present in the build container, never on the GPU box):
    /root/reference/model_results/unimodal/ffjord.pth
    /root/reference/model_results/multimodal/ffjord.pth
    (loaded by the reference at model_plot.py:72-74)

`torch.load(..., weights_only=True)` refuses these files (legacy torch<=1.5 whole-module pickles whose
globals are `ffjord.lib.layers.*`).  No unpickler of any kind is run on them here.  Instead this script
  1. lists the pickle opcodes with `pickletools.genops` (a disassembler: it parses bytes, it constructs
     nothing and imports nothing), and folds that listing into an inert tree of python primitives --
     a GLOBAL opcode becomes the *string* of its name, REDUCE/NEWOBJ/BUILD become tagged tuples;
  2. reads tensor metadata (storage key, offset, size, stride) and plain attributes (solver name,
     tolerances, ...) off that tree;
  3. reads the raw little-endian fp32 storages that follow the pickles with numpy.

Outputs (committed; data only):
    tests/golden/ffjord_unimodal.npz, tests/golden/ffjord_multimodal.npz   (state_dict-keyed fp32 arrays)
    tests/golden/ffjord_meta.json                                          (hyper-parameters, stored NFE)

Run from the repo root:  python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_flow_amd.legacy_checkpoint import read_checkpoint  # noqa: E402  (the non-executing reader lives in the package)

REF = "/root/reference/model_results"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    meta = {}
    for which in ("unimodal", "multimodal"):
        arrays, attrs = read_checkpoint(os.path.join(REF, which, "ffjord.pth"))
        np.savez(os.path.join(HERE, "ffjord_%s.npz" % which), **arrays)
        meta[which] = {
            "state_dict_shapes": {k: list(v.shape) for k, v in arrays.items()},
            "stored_num_evals": float(arrays["chain.0.odefunc.odefunc._num_evals"]),
            "modules": attrs,
        }
        print(which, "tensors:", len(arrays), "floats:", sum(a.size for a in arrays.values()))
    with open(os.path.join(HERE, "ffjord_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
