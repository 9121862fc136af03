# K-loop data-gradient product (config 5): row blocks per fragment x tiles per wave x steps per slab x workgroups aimed for
cd /tmp && export TMPDIR=/tmp
for cfg in "4 1 12 256" "2 2 24 256" "2 2 12 512" "4 2 12 256"; do
  set -- $cfg
  rm -rf /tmp/abt
  FDCAP_KLOOP_RB=$1 FDCAP_KLOOP_T=$2 FDCAP_KLOOP_SLAB=$3 FDCAP_KLOOP_WGS=$4 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/abt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --value-only --steps 1 --warmup 0 --iters 60 --config c5 $BENCH_ARGS > /tmp/abt.log 2>&1
  echo "rb $1 t $2 slab $3 wgs $4: $(python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/abt/t_results.db /dev/null | grep -i 'kloop\|part_sum' | cut -c1-100 | tr '\n' ' ')"
done
