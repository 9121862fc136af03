cd $GRAFT_REPO_ROOT
for n in 300 512 1024; do python tools/modes_trace.py local $n 2>&1 | grep "fit 1"; done
for n in 300 1024; do python tools/modes_trace.py dct $n 2>&1 | grep "fit 1"; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_modes
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_modes -o t -- python3 $GRAFT_REPO_ROOT/tools/modes_trace.py local 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_outliers.py /tmp/prof_modes/t_results.db 20 | head -24
