#!/bin/bash
# the whole GPU suite + smoke, repeated (flakiness check on one box)
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/t_gpu_$i.log 2>&1; echo "gpu run $i rc=$?"; tail -2 $O/t_gpu_$i.log
done
