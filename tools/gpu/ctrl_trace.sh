# Kernel trace of the control-loop leg of bench.py (controller kernels around the two solver kernels).
export TMPDIR=/tmp; cd /tmp && rm -rf /tmp/ctl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ctl -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ctl/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0][-40:]:42s} calls {r['Calls']:>5s} avg us {float(r['AverageNs'])/1e3:9.2f} min {float(r['MinNs'])/1e3:8.2f} total ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
