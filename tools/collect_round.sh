#!/bin/bash
# After tools/gpu/final.sh (through gpurun): copy the round's artefacts from gpurun_out/ to profiles/ and condense the PMC passes.
set -e
cd "$(dirname "$0")/.."
TAG=${TAG:-r06}
for f in bench_n1.json kernel_stats_bench_steps10.csv kernel_stats_config4_4096xh16.csv kernel_stats_config5_4096xh20.csv exact_mode_sweep.json parity_sweep.json controller_parity.json; do cp gpurun_out/${TAG}_$f profiles/${TAG}_$f; done
for h in 10 16 20; do
  d=$(mktemp -d)
  for p in fetch write sq1 sq2; do cp gpurun_out/${TAG}_pmc_h${h}_$p.csv profiles/; cp gpurun_out/${TAG}_pmc_h${h}_$p.csv $d/; done; cp gpurun_out/${TAG}_pmc_h${h}_source.sha256 $d/
  python tools/pmc_summary.py $d profiles/${TAG}_pmc_summary_h$h.json 4096 $h mpc_solve_jobs_kernel > /dev/null
  python tools/pmc_summary.py $d profiles/${TAG}_pmc_summary_prep_h$h.json 4096 $h mpc_prep_kernel > /dev/null
  rm -rf $d
done
python - <<'PY'
import json, os
TAG = os.environ.get("TAG", "r03")
d = json.load(open(f"profiles/{TAG}_parity_sweep.json"))
print("parity", sum(v["solves"] for v in d.values()), "solves, mismatches", sum(v["decision_mismatch"] for v in d.values()), "max rel err", max(v["max_rel_err"] for v in d.values()))
d = json.load(open(f"profiles/{TAG}_bench_n1.json")); r = d["roofline"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "solve", round(r["kernel_ms"], 4), "prep", round(r["prep_kernel_ms"], 4), "frac", round(r["frac"], 5), "traffic MB", r["traffic"] and round(r["traffic"] / 1e6, 1))
for k, v in d["secondary"].items(): print(" ", k, round(v["ms_per_step"], 3), round(v.get("prep_kernel_ms", 0), 3), round(v.get("solve_kernel_ms", v.get("solve_kernels_ms", 0)), 3), round(v["control_steps_per_s"]))
print("  clock", d["device_state"]["before"]["shader_clock_ghz"], d["device_state"]["after"]["shader_clock_ghz"])
print("  control loop", round(d["control_loop"]["ms_per_tick"], 4), round(d["control_loop"]["robot_ticks_per_s"]), "| with resets", round(d["control_loop_with_resets"]["ms_per_tick"], 4), round(d["control_loop_with_resets"]["robot_ticks_per_s"]), "| incl. torque map", round(d["control_steps_per_s_incl_torque_map"]))
for h in (10, 16, 20):
    for k in ("", "prep_"):
        dd = json.load(open(f"profiles/{TAG}_pmc_summary_{k}h{h}.json")); c = dd["counters_per_launch"]
        print("  pmc", h, k or "solve", "MB", round(dd["hbm_traffic_bytes_per_launch"] / 1e6, 1), "VALU/robot", round(c["SQ_INSTS_VALU"] / 4096), "LDS/robot", round(c["SQ_INSTS_LDS"] / 4096))
PY
