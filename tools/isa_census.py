"""Per-basic-block census of the solve kernels' ISA: fp64 ops, LDS ops, scratch (spill) traffic.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only rl-mpc-locomotion_amd/csrc/mpc_batch.hip -o /tmp/all.s
  python tools/isa_census.py /tmp/all.s [min_scratch_ops]"""
import re, sys
txt = open(sys.argv[1]).read()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for m in re.finditer(r'\n(_ZN[^\n]*mpc_solve_kernelILi(\d+)E[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end', txt, re.S):
    H, body = m.group(2), m.group(3)
    blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', body)
    tot = [0, 0, 0]
    print(f"H={H}: {len(blocks)} blocks")
    for b in blocks:
        lab = b.split(':')[0].strip()[:12]
        ins = [l for l in b.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
        f64 = sum(1 for l in ins if re.match(r'\tv_(fma|mul|add|div_fmas|div_fixup|rcp|rsq|min|max)_f64', l))
        lds = sum(1 for l in ins if l.startswith('\tds_'))
        sl = sum(1 for l in ins if l.startswith('\tscratch_load')); ss = sum(1 for l in ins if l.startswith('\tscratch_store'))
        tot[0] += f64; tot[1] += sl; tot[2] += ss
        if sl + ss >= thr: print(f"   {lab:12s} ins {len(ins):5d} f64 {f64:5d} lds {lds:4d} scratch ld {sl:4d} st {ss:4d}")
    print(f"   total f64 {tot[0]} scratch ld {tot[1]} st {tot[2]}")


def loops(path, H):
    """Aggregate per innermost loop (the assembler's 'in Loop: Header=.. Depth=..' comments)."""
    txt = open(path).read()
    m = re.search(r'\n(_ZN[^\n]*mpc_solve_kernelILi%dE[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end' % H, txt, re.S)
    agg = {}
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', m.group(2)):
        first = b.split('\n')[0]
        lab = b.split(':')[0]
        h = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', first)
        if 'Loop Header' in first:
            d = re.search(r'Depth=(\d+)', first).group(1); key = (lab[2:], d)
        elif h: key = (h.group(1), h.group(2))
        else: continue
        ins = [l for l in b.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
        a = agg.setdefault(key, [0, 0, 0, 0, 0])
        a[0] += len(ins); a[1] += sum(1 for l in ins if re.match(r'\tv_(fma|mul|add|fmac)_f64', l))
        a[2] += sum(1 for l in ins if l.startswith('\tds_')); a[3] += sum(1 for l in ins if l.startswith('\tscratch'))
        a[4] += sum(1 for l in ins if l.startswith('\ts_barrier'))
    for k, a in agg.items():
        if a[0] > 60: print(f"   loop {k[0]:10s} depth {k[1]} ins {a[0]:5d} f64 {a[1]:4d} lds {a[2]:4d} scratch {a[3]:4d} barriers {a[4]}")


if len(sys.argv) > 3:
    loops(sys.argv[1], int(sys.argv[3]))


def spill_cost(path, H):
    """Estimated scratch instructions executed per wave and solve: per-block static counts weighted by the trip counts of
    the enclosing loops (sweep loops G, the Ruiz loop 10, the ADMM inner loop 50, the check loop 2, other loops 4)."""
    txt = open(path).read()
    m = re.search(r'\n(_ZN[^\n]*mpc_solve_kernelILi%dE[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end' % H, txt, re.S)
    blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', m.group(2))
    info = {}
    for b in blocks:                      # loop key -> barriers
        first = b.split('\n')[0]; lab = b.split(':')[0]
        h = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', first)
        key = lab[2:] if 'Loop Header' in first else (h.group(1) if h else None)
        if key: info[key] = info.get(key, 0) + b.count('\ts_barrier')
    G = 2 * H
    def trips(key):
        nb = info.get(key, 0)
        return {6: G, 4: 50, 15: 2, 3: 10}.get(nb, 4)
    total = 0; detail = {}
    for b in blocks:
        first = b.split('\n')[0]; lab = b.split(':')[0]
        sc = sum(1 for l in b.split('\n') if l.startswith('\tscratch'))
        if not sc: continue
        w = 1
        h = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', first)
        par = re.search(r'Parent Loop (BB\d+_\d+) Depth=(\d+)', b.split('\n')[0])
        if 'Loop Header' in first: w = trips(lab[2:])
        elif h: w = trips(h.group(1))
        if (h and h.group(2) == '2') or ('Depth=2' in first): w *= 2          # nested in the check loop
        total += sc * w
        k = (lab[2:] if 'Loop Header' in first else (h.group(1) if h else 'straight'))
        detail[k] = detail.get(k, 0) + sc * w
    print(f"H={H}: weighted scratch ops per wave-solve ~ {total}  " + ' '.join(f"{k}:{v}" for k, v in sorted(detail.items(), key=lambda x: -x[1])[:6]))


if len(sys.argv) > 4 and sys.argv[4] == 'cost':
    spill_cost(sys.argv[1], int(sys.argv[3]))
