"""Register-budget regression guard for the solve kernel the OSQP mode runs -- the persistent job kernel mpc_solve_jobs_kernel<H>
(no GPU needed: hipcc cross-compiles gfx950).

The kernel's speed hinges on the register allocator keeping the 6x6 fp64 tile of every thread out of scratch memory inside
the hot loops (a handful of spill instructions per sweep step cost integer factors, because all CUs
spill at once).  That property is invisible to the parity tests and moves with any edit, so it is asserted on the ISA."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_census  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("isa") / "mpc_horizons.s")
    isa_census.compile_to_asm(os.path.join(ROOT, "rl-mpc-locomotion_amd", "csrc"), out, (10, 12, 16, 20))
    return open(out).read()


def test_tuned_horizons_are_instantiated(asm):
    assert isa_census.horizons(asm) == [10, 12, 16, 20]


def test_job_results_are_complete_before_they_are_published(asm):
    """mpc_solve_jobs_kernel hands an ADMM job's results to a polish job in another workgroup (possibly on another XCD) through device-coherent
    stores and a relaxed flag (`ready[pos]`): every thread must have WAITED for its stores (s_waitcnt vmcnt(0)) before the flag goes out.  The
    wait is spelled out in the source (__builtin_amdgcn_s_waitcnt), not left to the lowering of a workgroup-scope fence: in every
    instantiation a full s_waitcnt vmcnt(0) sits between the last result store and the atomic add on `sched[tail]` that opens the publish."""
    import re
    for h in (10, 12, 16, 20):
        body = re.search(isa_census.KERNELS["jobs"] % h, asm, re.S).group(2)
        lines = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        # the publish is the atomic add on sched[kSchedTail] (byte offset 8; the job fetches are on offsets 0 and 4)
        pubs = [i for i, l in enumerate(lines) if l.startswith("global_atomic_add") and "offset:8" in l]
        assert len(pubs) == 1, (h, pubs)
        i = pubs[0]
        prev_store = max(j for j in range(i) if lines[j].startswith("global_store"))
        assert "sc1" in lines[prev_store], lines[prev_store]          # the results go out as device-coherent (write-through) stores
        assert any(re.match(r"s_waitcnt\s+vmcnt\(0\)", l) for l in lines[prev_store:i]), (h, lines[prev_store:i])


def test_benchmark_horizon_hot_loops_are_scratch_free(asm):
    loops = isa_census.loop_stats(asm, 10)
    sweeps = [a for a in loops.values() if a["role"] == "sweep"]
    admm = [a for a in loops.values() if a["role"] == "admm-iteration"]
    assert len(sweeps) == 3 and len(admm) == 1, {k: (a["ins"], a["depth"], a["role"]) for k, a in loops.items()}   # factor, refactor, polish
    for a in sweeps + admm:
        assert a["scratch"] == 0 and a["barriers"] == 0, a          # one wavefront per robot: no workgroup barrier anywhere
    # three pivot pairs per trip: 72 FMA + 24 for the pair's B-transformed row per thread and pair are the floor (32 per pivot step);
    # everything else stays below 95 instructions per pivot step (round 3: 87-91, depending on what the allocator parks in AGPRs)
    for a in sweeps:
        assert a["ins"] <= 6 * 95, a
    # the ADMM iteration: tile product (72 FMA) + the foot phase.  Round 2 ended at 427 instructions; round 3 (quad reduce-scatter instead
    # of all-sum + select, [row][slot] partials, G S^-1 form) at 375-389 -- a branch cascade or a layout that costs the reader twice the
    # loads shows up here
    assert admm[0]["ins"] <= 400 and admm[0]["f64"] <= 210, admm[0]


def test_spill_estimate_stays_bounded(asm):
    # weighted scratch instructions per wave and solve (tools/isa_census.py).  h = 10: one wave per SIMD with the full register budget
    # (AGPRs as spill space), no scratch memory at all.  h = 16 / 20: multi-wave workgroups at two waves per SIMD (256 registers), which
    # was measured 25 % faster in spite of the spill code it needs; the bound keeps that spill code from growing.
    limits = {10: 50, 12: 2100, 16: 2100, 20: 2800}      # (round 3: 8 / 1814 / 1825 / 2274; h = 12, 16: tile and foot lanes are different threads
    #  that share their registers, Cfg::FOOT0)
    for h, lim in limits.items():
        total, detail = isa_census.spill_cost(asm, h)
        assert total <= lim, (h, total, detail)


def test_long_horizon_admm_iteration_stays_nearly_scratch_free(asm):
    for h, lim_admm, lim_sweep in ((12, 4, 2), (16, 4, 2), (20, 16, 6)):      # (h = 12, 16 with separate foot threads: none at all)
        admm = [a for a in isa_census.loop_stats(asm, h).values() if a["role"] == "admm-iteration"]
        assert admm and all(a["scratch"] <= lim_admm for a in admm), (h, admm)
        sweeps = [a for a in isa_census.loop_stats(asm, h).values() if a["role"] == "sweep"]
        assert len(sweeps) == 3 and all(a["scratch"] <= lim_sweep for a in sweeps), (h, sweeps)      # round 2: up to 29 per trip


def test_exact_mode_kernel_runs_without_scratch(asm):
    """mpc_exact_kernel<10> (the exact mode's first launch: active-set method + polish): one wave per SIMD, no scratch access in any
    loop (a few bytes spilled once outside the loops are tolerated: the allocator moves them with every edit)."""
    import re
    m = re.search(r'\.amdhsa_kernel [^\n]*mpc_exact_kernelILi10E.*?\.end_amdhsa_kernel', asm, re.S)
    assert m, "no mpc_exact_kernel<10> in the assembly"
    size = int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(0)).group(1))
    assert size <= 32, size
    body = re.search(r'\n(_ZN[^\n]*mpc_exact_kernelILi10EE[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end', asm, re.S).group(2)
    for block in re.split(r'\n(?=\.LBB\d+_\d+:)', body):
        if "Loop" in block.split("\n")[0] or "Loop" in " ".join(block.split("\n")[:3]):
            assert not re.search(r'\n\tscratch_', block), block.split("\n")[0]
