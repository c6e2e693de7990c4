#!/usr/bin/env python3
"""Static instruction counts of kernels from the compiler's assembly:  tools/isa_count.py file.hip [name-substring ...]
(vector / scalar / LDS / global instructions per kernel; loops count once -- a first look at what a kernel's wavefront issues)."""
import re, subprocess, sys, os, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; pats = sys.argv[2:]
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
name = None; cnt = {}
for l in open(out):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name = m.group(1); cnt[name] = [0, 0, 0, 0, 0]; continue
    if l.startswith('.Lfunc_end'):
        name = None
    if name is None: continue
    t = l.strip()
    if not t or t[0] in '.;' or t.endswith(':'): continue
    c = cnt[name]; c[0] += 1
    if t.startswith('v_'): c[1] += 1
    elif t.startswith('s_'): c[2] += 1
    elif t.startswith('ds_'): c[3] += 1
    elif t.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): c[4] += 1
for n, c in cnt.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split('(')[0]
    if pats and not any(p in d for p in pats): continue
    print("%-44s total %5d  valu %5d  salu %5d  lds %4d  mem %4d" % (d[:44], *c))
