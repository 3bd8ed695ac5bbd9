V=rl-mpc-locomotion_amd/csrc/variants
for rep in 1 2; do
for lib in new trk; do
  for cfg in "10 4096" "16 2048" "20 1024"; do
    set -- $cfg
    if [ $lib = new ]; then unset MPC_LIB_PATH; else export MPC_LIB_PATH=$V/libmpc_batch_$lib.so; fi
    echo -n "== $lib h=$1 n=$2: "
    timeout 300 python bench.py --horizon $1 --robots $2 --steps 20 --warmup 3 --no-cpu-baseline --no-control-loop 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('assemble_kernel_ms'))"
  done
done
done
