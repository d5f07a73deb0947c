"""Compact schedule of one kernel from hipcc -S output: runs of loads / LDS / MFMA / waits, one line per run.
usage: python tools/isa_trace.py file.s <mangled-name-regex> [--coarse]"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + sys.argv[2] + r'\S*):', s, re.M)
body = s[m.start():s.index('.end_amdhsa_kernel', m.start())].split('\n')
coarse = '--coarse' in sys.argv
out, last, cnt = [], None, 0
for l in body:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'): k = 'mfma'
    elif op.startswith('buffer_load') or op.startswith('global_load'): k = 'LOAD'
    elif op.startswith('global_store') or op.startswith('buffer_store'): k = 'STORE'
    elif op.startswith('ds_write') or op.startswith('ds_store'): k = 'dswrite'
    elif op.startswith('ds_read') or op.startswith('ds_load'): k = 'dsread'
    elif op.startswith('s_waitcnt'): k = l if 'vmcnt' in l else ('lgkm' if coarse else l)
    elif op.startswith('s_barrier'): k = 'BARRIER'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): k = l
    elif l.endswith(':'): k = l
    elif op.startswith('scratch_'): k = 'SCRATCH'
    else: k = 'alu'
    if coarse and k in ('mfma', 'dsread', 'alu', 'lgkm'):
        k = 'compute'
    if k == last:
        cnt += 1
    else:
        if last: out.append(f"{last} x{cnt}")
        last, cnt = k, 1
out.append(f"{last} x{cnt}")
print('\n'.join(out))
