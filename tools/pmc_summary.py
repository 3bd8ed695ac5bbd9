"""Condense the rocprofv3 --pmc CSVs written by tools/pmc_passes.sh into profiles/rNN_pmc_summary.json.
usage: python tools/pmc_summary.py <dir with pmc_*.csv> <out.json> [n_robots] [horizon] [kernel name substring, default mpc_solve_kernel]"""
import csv, glob, json, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
h = int(sys.argv[4]) if len(sys.argv) > 4 else 10
kname = sys.argv[5] if len(sys.argv) > 5 else "mpc_solve_kernel"
per = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value
for f in sorted(glob.glob(os.path.join(src, "*pmc_*.csv"))):
    for row in csv.DictReader(open(f)):
        if kname not in row["Kernel_Name"]:   # (default: the dominant kernel)
            continue
        per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
counters = {}
for name, d in per.items():
    vals = [d[k] for k in sorted(d)][-5:]       # the 5 timed (warm-started) dispatches of bench.py --steps 5 --warmup 2
    counters[name] = sum(vals) / len(vals)
traffic = (2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024
N, M = 12 * h, 20 * h
alg = n * ((56 + 4 * h) * 4 + (2 * N + 2 * M + 2) * 8 * 2 + N * 8 + 8 * 4)   # path level: input record, warm-start state r+w, forces, info
rec = n * ((2 * N + M + 36 * h + 4) + (2 * N + 16 + 116)) * 8 * 2                     # scale + QP records handed from the prep kernel to the solve kernel (written once, read once; csrc/mpc_core.h: SC_LEN, QP_LEN)
c = counters
sha = None
for f in glob.glob(os.path.join(src, "*source.sha256")):
    sha = open(f).read().strip()
summary = {
    "kernel_source_sha256": sha,    # rl_mpc_locomotion_amd._lib.kernel_source_hash() on the box that measured (bench.py quotes this file only when the tree still matches)
    "command": "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-control-loop --no-secondary (tools/pmc_passes.sh: one rocprofv3 --pmc pass per counter group, --kernel-trace only)",
    "kernel": f"{kname}<{h}>, {n} robots per launch, mean of the 5 timed (warm-started) dispatches",
    "counters_per_launch": counters,
    "hbm_traffic_bytes_per_launch": traffic,
    "traffic_rule": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section) so it is doubled; other access widths and WRITE_SIZE are uncalibrated there, and the fabric counters include Infinity-Cache hits -- treat as an upper bound on HBM bytes",
    "algorithmic_bytes_per_launch": alg,
    "inter_kernel_record_bytes_per_launch": rec,
    "derived": {
        "lds_bank_conflict_frac_of_lds_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        "wave_cycles_parked_frac (SQ_WAIT_ANY/SQ_WAVE_CYCLES)": c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
        "wave_cycles_issue_stall_frac (SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)": c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
        "wave_cycles_issuing_frac (SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES)": c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
    },
}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary["derived"]), "traffic GB", traffic / 1e9)
