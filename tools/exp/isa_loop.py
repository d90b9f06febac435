"""Instruction mix between consecutive MFMAs of one kernel's ISA (a .s from hipcc -S): which vector / LDS / memory instructions
the compiler placed in each MFMA gap.   python tools/exp/isa_loop.py file.s kernel_name_substring"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
name = m.group(1)
body = s[m.start():s.index('.end_amdhsa_kernel', m.start())].split('\n')
ins = [l.strip().split()[0] for l in body if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
mf = [n for n, x in enumerate(ins) if x.startswith('v_mfma')]
print(name, 'instructions', len(ins), 'MFMAs', len(mf))
def cls(x):
    if x.startswith('v_mfma'): return 'M'
    if x.startswith('v_exp'): return 'e'
    if x.startswith('v_'): return 'v'
    if x.startswith('ds_'): return 'L'
    if x.startswith(('buffer_', 'global_', 'scratch_')): return 'G'
    if x.startswith('s_waitcnt'): return 'w'
    if x.startswith('s_barrier'): return 'B'
    if x.startswith('s_nop'): return 'n'
    if x.startswith('s_cbranch') or x.startswith('s_branch'): return 'J'
    return 's'
print(''.join(cls(x) for x in ins[mf[0] - 60: mf[-1] + 120]))
for k in ('vgpr_count', 'vgpr_spill_count', 'sgpr_spill_count'):
    mm = re.findall(r'\.name:\s+' + re.escape(name) + r'\n.*?\.' + k + r':\s+(\d+)', s, re.S)
    print(k, mm[:1])
