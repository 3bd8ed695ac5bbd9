cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for i in 1 2; do
python bench.py --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PRODUCT', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4))"
done
bash tools/gpu/ab_lib.sh gs
