cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kcl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kcl -o k -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import os; os.chdir('$GRAFT_REPO_ROOT')
import bench, torch
print(bench.control_loop_leg(4096, 10, torch.device('cuda:0'))['ms_per_tick'])" 2>&1 | tail -1
f=$(find /tmp/kcl -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | sed 's/(anonymous namespace):://; s/_ZN12_GLOBAL__N_1//' | cut -c1-50,140-400 | head -9
