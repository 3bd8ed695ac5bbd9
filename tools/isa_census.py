"""Census of the solve kernels' ISA: fp64 ops, LDS ops and scratch (register spill) instructions per basic block and per loop.

The solve kernel keeps a 6x6 fp64 tile per thread in registers for its whole life; whether hipcc's allocator spills part of it
inside the hot loops decides the kernel's speed by integer factors, and the decision moves with any change
of the code shape.  This tool reads the assembly and says where the scratch instructions are.

  python -c "import sys; sys.path.insert(0, 'tools'); import isa_census as c; c.compile_to_asm('rl-mpc-locomotion_amd/csrc', '/tmp/all.s', (10, 16, 20))"
  python tools/isa_census.py /tmp/all.s                 # per horizon: loops, weighted spill estimate
  python tools/isa_census.py /tmp/all.s --blocks 8      # also every block with >= 8 scratch instructions

Loops are recognised by the assembler's "in Loop: Header=.. Depth=.." comments and classified by what they contain: a sweep loop
has six v_rcp_f64 (six pivot steps per trip); the ADMM iteration is the depth-2 loop with the quad exchanges (v_mov_b32_dpp);
the depth-1 loop around both is the check / refactor loop.  (The h = 10 solve kernel is a single wave: it has no s_barrier.)
(tests/test_isa_budget.py asserts the hot loops of the benchmark horizon stay free of scratch traffic.)"""
import re
import subprocess
import sys

# (the kernel the OSQP mode runs: the persistent job kernel mpc_solve_jobs_kernel<H>; set KERNEL = "one" for the one-job-per-workgroup
# instantiation mpc_solve_kernel<H, false>; the exact-mode one, <H, true>, is not performance critical)
KERNELS = {"jobs": r'\n(_ZN[^\n]*mpc_solve_jobs_kernelILi%sEE[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end',
           "one": r'\n(_ZN[^\n]*mpc_solve_kernelILi%sELb0E[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end'}
KERNEL = "jobs"
KERNEL_RE = KERNELS[KERNEL]


def compile_to_asm(csrc_dir, out, horizons=(10, 12, 16, 20), extra=()):
    """Device assembly of the kernel sets of the given planning horizons, concatenated into `out` (hipcc cross-compiles gfx950 without a
    GPU; every horizon is its own translation unit, csrc/mpc_horizon.hip -DMPC_H=h: compiled in parallel)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    src = os.path.join(csrc_dir, "mpc_horizon.hip")

    def one(h):
        o = f"{out}.h{h}.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DMPC_H={h}", "-S", "--cuda-device-only", *extra, src, "-o", o]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return o
    with ThreadPoolExecutor(max_workers=min(len(horizons), len(os.sched_getaffinity(0)))) as pool:
        parts = list(pool.map(one, horizons))
    with open(out, "w") as f:
        for pth in parts:
            f.write(open(pth).read())
            f.write("\n")
    return out


def horizons(txt):
    return sorted(int(h) for h in set(re.findall(r'mpc_solve_kernelILi(\d+)E', txt)))


def _blocks(txt, H):
    m = re.search(KERNEL_RE % H, txt, re.S)
    if not m:
        raise KeyError(f"no mpc_solve_kernel<{H}> in the assembly")
    return re.split(r'\n(?=\.LBB\d+_\d+:)', m.group(2))


def _instructions(block):
    return [l for l in block.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]


def _loop_of(block):
    """(header label, depth) of the innermost loop a block belongs to, or None."""
    lines = block.split('\n')
    first = lines[0]
    lab = block.split(':')[0]
    head = ' '.join(l for l in lines[:4] if l.lstrip().startswith((';', '.L')))   # an inner loop header carries a "Parent Loop .." line per enclosing loop first, then "=> This Inner Loop Header"
    if 'Loop Header' in head:
        return lab[2:], int(re.findall(r'Loop Header: Depth=(\d+)', head)[-1])
    h = re.search(r'Header=(BB\d+_\d+) Depth=(\d+)', first)
    return (h.group(1), int(h.group(2))) if h else None


def block_stats(txt, H):
    """[(label, n_instructions, n_f64, n_lds, scratch_loads, scratch_stores)] for every basic block."""
    out = []
    for b in _blocks(txt, H):
        ins = _instructions(b)
        out.append((b.split(':')[0].strip(), len(ins),
                    sum(1 for l in ins if re.match(r'\tv_(fma|fmac|mul|add|div_fmas|div_fixup|rcp|rsq|min|max)_f64', l)),
                    sum(1 for l in ins if l.startswith('\tds_')),
                    sum(1 for l in ins if l.startswith('\tscratch_load')),
                    sum(1 for l in ins if l.startswith('\tscratch_store'))))
    return out


def loop_stats(txt, H):
    """{header: dict(depth, ins, f64, lds, scratch, barriers, role)} aggregated over the blocks of each innermost loop."""
    agg = {}
    for b in _blocks(txt, H):
        key = _loop_of(b)
        if key is None:
            continue
        ins = _instructions(b)
        a = agg.setdefault(key[0], dict(depth=key[1], ins=0, f64=0, lds=0, scratch=0, barriers=0))
        a["ins"] += len(ins)
        a["f64"] += sum(1 for l in ins if re.match(r'\tv_(fma|mul|add|fmac)_f64', l))
        a["lds"] += sum(1 for l in ins if l.startswith('\tds_'))
        a["scratch"] += sum(1 for l in ins if l.startswith('\tscratch'))
        a["barriers"] += sum(1 for l in ins if l.startswith('\ts_barrier'))
        a["rcp"] = a.get("rcp", 0) + sum(1 for l in ins if l.startswith('\tv_rcp_f64'))
        a["dpp"] = a.get("dpp", 0) + sum(1 for l in ins if l.startswith('\tv_mov_b32_dpp'))
    for a in agg.values():
        # a sweep trip is six pivot steps with a reciprocal each (long horizons), or three pivot pairs with one reciprocal (of the
        # pair's 2 x 2 determinant) each (h = 10); the ADMM iteration is the tight loop with the quad exchanges (DPP moves) and no
        # reciprocal; the loop around it (iterations + check + refactor) is the big one with DPP moves.  (Depths differ between the
        # one-job kernel and the persistent job kernel, whose job loops sit outside: roles go by content.)
        if a.get("rcp", 0) in (3, 6) and a["ins"] < 1000 and a["lds"] > 40:      # (a sweep trip publishes and fetches pivot rows: 100+ LDS instructions)
            a["role"] = "sweep"
        elif a.get("dpp", 0) and a["ins"] < 700 and not a.get("rcp", 0):
            a["role"] = "admm-iteration"
        elif a["ins"] > 2000 and a.get("dpp", 0):
            a["role"] = "check-loop"
        else:
            a["role"] = ""
    return agg


def spill_cost(txt, H):
    """Estimated scratch instructions executed per wave and solve: static counts weighted by rough trip counts (a sweep loop
    runs H trips -- each of the three sweep loops about once per solve --, the ADMM iteration ~50, the check loop ~2, anything else 4;
    straight-line code once)."""
    loops = loop_stats(txt, H)
    trips = {"sweep": H, "admm-iteration": 50, "check-loop": 2, "ruiz-pass": 10}
    total, detail = 0, {}
    for b in _blocks(txt, H):
        sc = sum(1 for l in _instructions(b) if l.startswith('\tscratch'))
        if not sc:
            continue
        key = _loop_of(b)
        w, name = 1, "straight-line"
        if key is not None:
            a = loops[key[0]]
            w = trips.get(a["role"], 4)
            name = f"{key[0]}({a['role'] or 'loop'})"
        total += sc * w
        detail[name] = detail.get(name, 0) + sc * w
    return total, detail


def main(argv):
    txt = open(argv[1]).read()
    thr = int(argv[argv.index("--blocks") + 1]) if "--blocks" in argv else None
    for H in horizons(txt):
        bs = block_stats(txt, H)
        print(f"h={H}: {len(bs)} blocks, fp64 instructions {sum(b[2] for b in bs)}, scratch loads {sum(b[4] for b in bs)} stores {sum(b[5] for b in bs)}")
        if thr is not None:
            for lab, n, f64, lds, sl, ss in bs:
                if sl + ss >= thr:
                    print(f"   {lab:12s} ins {n:5d} f64 {f64:5d} lds {lds:4d} scratch ld {sl:4d} st {ss:4d}")
        for k, a in loop_stats(txt, H).items():
            if a["ins"] > 60:
                print(f"   loop {k:10s} depth {a['depth']} ins {a['ins']:5d} f64 {a['f64']:4d} lds {a['lds']:4d} scratch {a['scratch']:4d} barriers {a['barriers']:2d} {a['role']}")
        total, detail = spill_cost(txt, H)
        print(f"   weighted scratch instructions per wave and solve ~ {total}: " +
              ' '.join(f"{k}:{v}" for k, v in sorted(detail.items(), key=lambda x: -x[1])[:6]))


if __name__ == "__main__":
    main(sys.argv)
