# tools/prof_series.sh <kernel substring> [env assignments...]: per-block mean duration of a kernel over one bench step
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
env "$@" true
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace -d /tmp/ps -o ps -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_series.py /tmp/ps/ps_results.db "$pat" 50
