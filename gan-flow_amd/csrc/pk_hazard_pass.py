#!/usr/bin/env python3
"""Assembly-level hazard pass for gfx950 packed fp32 (part of the build of cnf_split.hip, see Makefile and DESIGN.md section 4).

Measured on MI355X (tools/hazard_probe.hip, profiles/r03_hazard_probe.txt): a packed fp32 instruction -- v_pk_fma_f32,
v_pk_mul_f32, v_pk_add_f32 -- whose LOW lane reads the HIGH dword of src1 (op_sel bit 1 set) returns a wrong low half in
lanes 48..63, now and then, when the other wave of its SIMD is issuing bf16 MFMAs back to back.  The same selection on src0
or src2, a crossed HIGH lane (op_sel_hi 0), and all forms with one wave per SIMD are right in every run.  The compiler forms
such instructions freely (a scalar broadcast from the upper half of a register pair, or library code such as powf), and the
eight-wave adjoint kernel runs two waves per SIMD beside MFMA streams by design.

The pass rewrites every such instruction into the equivalent one with src0 and src1 exchanged (multiplication and
addition commute bit for bit; the per-operand modifier bits move with their operands), so the crossed read lands on
src0, where it is harmless.  It refuses (exit 1) what it cannot rewrite: both of src0 and src1 crossed in the low lane.
usage: pk_hazard_pass.py in.s out.s        (prints how many instructions it rewrote)
       pk_hazard_pass.py --check file.s    (exit 1 if an instruction of the hazardous form is present)"""
import re
import sys

INS = re.compile(r"^(\s*)(v_pk_(?:fma|mul|add)_f32)(\s+)(.*?)(\s*(?:;.*)?)$")
MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01](?:,[01])*)\]")


def split_operands(text):
    """'v[0:1], v[2:3], s[4:5] op_sel:[0,1] neg_lo:[1,0]' -> (['v[0:1]', 'v[2:3]', 's[4:5]'], ' op_sel:[0,1] neg_lo:[1,0]')"""
    m = re.search(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b", text)
    ops, mods = (text[:m.start()], text[m.start():]) if m else (text, "")
    return [o.strip() for o in ops.split(",")], mods


def hazardous(mods):
    m = re.search(r"\bop_sel:\[([01](?:,[01])*)\]", mods)
    return bool(m) and m.group(1).split(",")[1] == "1"


def rewrite(line):
    m = INS.match(line)
    if not m:
        return line, False
    indent, op, gap, rest, tail = m.groups()
    ops, mods = split_operands(rest)
    if not hazardous(mods):
        return line, False
    sel = re.search(r"\bop_sel:\[([01](?:,[01])*)\]", mods).group(1).split(",")
    if sel[0] == "1":
        raise SystemExit("pk_hazard_pass: cannot rewrite (src0 and src1 both read their high dword in the low lane): " + line.strip())
    n_src = len(ops) - 1
    if n_src not in (2, 3):
        raise SystemExit("pk_hazard_pass: unexpected operand count: " + line.strip())
    ops[1], ops[2] = ops[2], ops[1]

    def swap(mm):
        bits = mm.group(2).split(",")
        bits[0], bits[1] = bits[1], bits[0]
        return "%s:[%s]" % (mm.group(1), ",".join(bits))

    mods = MOD.sub(swap, mods)
    # op_sel_hi defaults to all ones: exchanging two ones changes nothing, an absent list stays absent
    return "%s%s%s%s%s%s" % (indent, op, gap, ", ".join(ops), mods, tail), True


def main(argv):
    if len(argv) == 3 and argv[1] == "--check":
        bad = [l for l in open(argv[2]) if INS.match(l.rstrip("\n")) and hazardous(split_operands(INS.match(l.rstrip("\n")).group(4))[1])]
        for l in bad[:20]:
            print("pk_hazard_pass: hazardous form present:", l.strip())
        return 1 if bad else 0
    if len(argv) != 3:
        print(__doc__)
        return 2
    out, n = [], 0
    for line in open(argv[1]):
        new, changed = rewrite(line.rstrip("\n"))
        n += changed
        out.append(new)
    open(argv[2], "w").write("\n".join(out) + "\n")
    print("pk_hazard_pass: %d packed fp32 instructions rewritten (src0 <-> src1) in %s" % (n, argv[1]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
