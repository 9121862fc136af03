# per-rank cost of the strong-scaling shards on ONE GPU: the 1024-frame workload cut to 128/256/512 frames
cd $GRAFT_REPO_ROOT
for mode in ${MODES:-1 4 41}; do
for f in ${FRAMES:-128 1024}; do
  FDCAP_NN_STREAM=$mode python bench.py --frames $f --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print("stream mode $mode frames", d["config"]["frames"], "ms/step", round(d["ms_per_step"],1), "us/iter", round(d["ms_per_step"]*2,1), "in-loop NN ms", round(d["roofline"]["ms_per_launch"],3))"
done
done
