cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/p5_$v
FDCAP_SKIN_SPLIT=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p5_$v -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config c5 --value-only --steps 1 --warmup 0 --iters 100 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/p5_$v/t_results.db /dev/null | sed -n 4,12p | cut -c1-100
done
