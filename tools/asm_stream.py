#!/usr/bin/env python3
"""Prints the instruction-class stream of one kernel from a device assembly file (hipcc --cuda-device-only -S):
M mfma, v valu, T transcendental, d DPP valu, r/w LDS read/write, a LDS atomic, B barrier, W waitcnt, n nop, s salu,
g global/scratch, J branch.  usage: asm_stream.py file.s kernel_name_substring [width]"""
import sys


def cls(l):
    t = l.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.'):
        return None
    op = t[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith(('v_exp', 'v_rcp', 'v_sqrt', 'v_log', 'v_rsq')): return 'T'
    if op.startswith('v_'): return 'd' if ('dpp' in op or 'row_' in l or 'quad_perm' in l) else 'v'
    if op.startswith(('ds_read', 'ds_load')): return 'r'
    if op.startswith('ds_add'): return 'a'
    if op.startswith('ds_'): return 'w'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_nop'): return 'n'
    if op.startswith(('s_cbranch', 's_branch')): return 'J'
    if op.startswith('s_'): return 's'
    if op.startswith(('global', 'buffer', 'scratch', 'flat')): return 'g'
    return '?'


lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 160
start = [k for k, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith(l.split(':')[0])][0]
body = []
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'):
        break
    body.append(l)
seq = ''.join(c for c in map(cls, body) if c)
print(len(seq), {c: seq.count(c) for c in sorted(set(seq))})
for k in range(0, len(seq), width):
    print(seq[k:k + width])
