# kernel trace of the control loop WITH device-side resets (bench.control_loop_leg reset_every=37): per-kernel durations of the small (de-phased) ticks
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kbr
rocprofv3 --kernel-trace --output-format csv -d /tmp/kbr -o k -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import os; os.chdir('$GRAFT_REPO_ROOT')
import bench, torch
print(bench.env_bridge_loop_leg(4096, 10, torch.device('cuda:0'))['ms_per_tick'])" 2>&1 | tail -1
f=$(find /tmp/kbr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
out = []
for r in rows:
    n = r['Kernel_Name']
    for a, b in (('(anonymous namespace)::', ''), ('void ', ''), ('mpc::', '')):
        n = n.replace(a, b)
    out.append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - int(r['Start_Timestamp']), n[:40]))
# the last 60 kernels: a few ticks of the timed region
prev_end = None
for s, d, n in out[-64:]:
    gap = '' if prev_end is None else f'gap {(s - prev_end) / 1e3:7.1f} us'
    print(f'{s / 1e3:12.1f} us  {d / 1e3:8.1f} us  {n:40s} {gap}')
    prev_end = s + d
PY
