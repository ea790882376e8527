"""Read the reference's legacy FFJORD checkpoints (`model_results/*/ffjord.pth`, torch <= 1.5 whole-module pickles,
loaded by the reference at /root/reference/model_plot.py:72-74) WITHOUT unpickling them, and build the MI355X model
that carries their weights.

`torch.load(..., weights_only=True)` refuses these files (their globals are `ffjord.lib.layers.*` classes) and
`weights_only=False` would execute them.  No unpickler of any kind runs here.  Instead
  1. the pickle opcodes are listed with `pickletools.genops` (a disassembler: it parses bytes, constructs nothing and
     imports nothing) and folded into an inert tree of python primitives -- a GLOBAL opcode becomes the *string* of
     its name, REDUCE / NEWOBJ / BUILD become tagged lists;
  2. tensor metadata (storage key, offset, size, stride) and plain attributes (solver name, tolerances, ...) are read
     off that tree;
  3. the raw little-endian fp32 storages that follow the pickles are read with numpy.
"""
import io
import json
import pickletools
import struct

import numpy as np


class _Mark:
    pass


def inert_tree(buf, start):
    """Fold the opcode listing of ONE pickle starting at `start` into inert primitives.

    Returns (tree, end_offset).  Nothing is imported, called or instantiated: callables named in the
    stream are kept as ('global', 'module name') tuples and applications of them as tagged tuples.
    """
    bio = io.BytesIO(buf)
    bio.seek(start)
    stack, memo = [], {}

    def pop_to_mark():
        items = []
        while not isinstance(stack[-1], _Mark):
            items.append(stack.pop())
        stack.pop()
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(bio):
        n = op.name
        if n == "PROTO":
            pass
        elif n == "MARK":
            stack.append(_Mark())
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "BININT", "BININT1", "BININT2", "BINFLOAT", "LONG1",
                   "SHORT_BINSTRING", "BINSTRING", "LONG4"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "GLOBAL":
            stack.append(("global", arg))
        elif n in ("BINPUT", "LONG_BINPUT"):
            memo[arg] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif n == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "APPEND":
            v = stack.pop()
            _seq(stack[-1]).append(v)
        elif n == "APPENDS":
            items = pop_to_mark()
            _seq(stack[-1]).extend(items)
        elif n == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            _map(stack[-1])[k] = v
        elif n == "SETITEMS":
            items = pop_to_mark()
            d = _map(stack[-1])
            for i in range(0, len(items), 2):
                d[items[i]] = items[i + 1]
        elif n == "BINPERSID":
            stack[-1] = ("pers", stack[-1])
        elif n == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            stack.append(["reduce", fn, args, {}, []])  # [tag, fn, args, items, state]
        elif n == "NEWOBJ":
            args = stack.pop()
            cls = stack.pop()
            stack.append(["obj", cls, args, {}, []])
        elif n == "BUILD":
            state = stack.pop()
            stack[-1][4].append(state)
        elif n == "STOP":
            return stack.pop(), bio.tell()
        else:
            raise ValueError("opcode %s not expected in a torch<=1.5 checkpoint" % n)
    raise ValueError("no STOP")


def _map(node):
    return node if isinstance(node, dict) else node[3]        # OrderedDict() stand-in keeps items in [3]


def _seq(node):
    return node if isinstance(node, list) and (not node or node[0] not in ("reduce", "obj")) else node[3].setdefault("_list", [])


def _is(node, tag):
    return isinstance(node, list) and len(node) == 5 and node[0] == tag


def _global_name(node):
    if isinstance(node, tuple) and len(node) == 2 and node[0] == "global":
        return node[1]
    if isinstance(node, tuple) and len(node) == 2 and node[0] == "pers":
        # ('module', <global>, source_path, None) persistent id of a class
        return node[1][1][1]
    return None


def tensor_meta(node):
    """-> (storage_key, numel_storage, offset, size, stride, requires_grad) or None."""
    if not _is(node, "reduce"):
        return None
    fn = _global_name(node[1])
    if fn == "torch._utils _rebuild_parameter":
        inner = tensor_meta(node[2][0])
        return inner[:5] + (bool(node[2][1]),)
    if fn == "torch._utils _rebuild_tensor_v2":
        pers, offset, size, stride, rg = node[2][0], node[2][1], node[2][2], node[2][3], node[2][4]
        _tag, _cls, key, _dev, numel, _view = pers[1]
        return (key, numel, offset, tuple(size), tuple(stride), bool(rg))
    return None


def module_state(node):
    assert _is(node, "obj"), node
    st = {}
    for s in node[4]:
        st.update(s)
    return st


def walk_modules(node, prefix, tensors, attrs):
    st = module_state(node)
    cls = _global_name(node[1])
    attrs[prefix.rstrip(".") or "<root>"] = {"class": cls}
    for group in ("_parameters", "_buffers"):
        for name, t in _map(st.get(group, {})).items():
            if t is None:
                continue
            tensors[prefix + name] = tensor_meta(t)
    for k, v in st.items():
        if k.startswith("_") and k in ("_parameters", "_buffers", "_modules", "_backward_hooks", "_forward_hooks",
                                       "_forward_pre_hooks", "_state_dict_hooks", "_load_state_dict_pre_hooks"):
            continue
        tm = tensor_meta(v)
        if tm is not None:
            attrs[prefix.rstrip(".") or "<root>"][k] = {"tensor_shape": list(tm[3])}
        elif isinstance(v, (int, float, str, bool, type(None))):
            attrs[prefix.rstrip(".") or "<root>"][k] = v
        elif isinstance(v, (tuple, list, dict)) and not _is(v, "obj") and not _is(v, "reduce"):
            name = _global_name(v)
            attrs[prefix.rstrip(".") or "<root>"][k] = ("<global %s>" % name) if name else json.loads(json.dumps(v, default=str))
    for name, child in _map(st.get("_modules", {})).items():
        walk_modules(child, prefix + name + ".", tensors, attrs)


def read_checkpoint(path):
    buf = open(path, "rb").read()
    pos = 0
    trees = []
    for _ in range(5):            # magic, protocol, sys-info, model, storage keys  
        tree, pos = inert_tree(buf, pos)
        trees.append(tree)
    assert trees[0] == 0x1950A86A20F9469CFC6C and trees[1] == 1001, "not a legacy torch checkpoint"
    assert trees[2]["little_endian"] is True
    keys = trees[4]
    storages = {}
    for key in keys:
        (numel,) = struct.unpack_from("<q", buf, pos)
        pos += 8
        storages[key] = np.frombuffer(buf, dtype="<f4", count=numel, offset=pos).copy()
        pos += 4 * numel
    assert pos == len(buf), (pos, len(buf))
    tensors, attrs = {}, {}
    walk_modules(trees[3], "", tensors, attrs)
    arrays = {}
    for name, (key, numel, offset, size, stride, _rg) in tensors.items():
        st = storages[key]
        assert st.size == numel
        n = int(np.prod(size)) if size else 1
        # all tensors in these files are contiguous
        expect = tuple(int(np.prod(size[i + 1:])) for i in range(len(size)))
        assert tuple(stride) == expect, (name, size, stride)
        arrays[name] = st[offset:offset + n].reshape(size).astype(np.float32)
    return arrays, attrs




def state_dict_from_checkpoint(path):
    """-> ({state_dict key: torch.Tensor}, attrs) of a legacy whole-module checkpoint; nothing in the file is executed."""
    import torch
    arrays, attrs = read_checkpoint(path)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrays.items()}, attrs


def load_legacy_ffjord(path, device=None):
    """The CNF model stored in a legacy `ffjord.pth`, rebuilt from the library's modules: architecture from the tensor
    shapes (data dimension, hidden widths, number of blocks), solver / tolerances / divergence estimator from the
    stored attributes, weights by state_dict key.  Returns the model in eval mode (on `device` if given)."""
    from .gu_ffjord import build_model
    sd, attrs = state_dict_from_checkpoint(path)
    blocks = sorted({int(k.split(".")[1]) for k in sd if k.startswith("chain.")})
    pre = "chain.0.odefunc.odefunc.diffeq.layers."
    n_layers = 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))
    dims = tuple(int(sd["%s%d._layer.weight" % (pre, i)].shape[0]) for i in range(n_layers - 1))
    dim = int(sd[pre + "0._layer.weight"].shape[1])
    cnf = attrs.get("chain.0", {})
    fn = attrs.get("chain.0.odefunc.odefunc", {})
    div = str(fn.get("divergence_fn", ""))
    model = build_model(dim, dims=dims, num_blocks=len(blocks), time_length=float(sd["chain.0.sqrt_end_time"]) ** 2,
                        train_T="chain.0.sqrt_end_time" in sd, divergence_fn="approximate" if "approx" in div else "brute_force",
                        solver=cnf.get("solver", "dopri5"), atol=float(cnf.get("atol", 1e-5)), rtol=float(cnf.get("rtol", 1e-5)),
                        rademacher=bool(fn.get("rademacher", False)))
    model.load_state_dict(sd)
    for i, blk in enumerate(model.chain):
        a = attrs.get("chain.%d" % i, {})
        blk.test_solver = a.get("test_solver", blk.solver)
        blk.test_atol = float(a.get("test_atol", blk.atol))
        blk.test_rtol = float(a.get("test_rtol", blk.rtol))
    model.eval()
    return model.to(device) if device is not None else model
