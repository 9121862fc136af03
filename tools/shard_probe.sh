# per-rank cost of the strong-scaling shards on ONE GPU: the 1024-frame workload cut to 128/256/512 frames
# MODES: FDCAP_NN_STREAM values to compare (-2 = default choice by launch size); FRAMES: clip lengths
cd $GRAFT_REPO_ROOT
for mode in ${MODES:--2}; do
for f in ${FRAMES:-128 256 512 1024}; do
  FDCAP_NN_STREAM=$mode python bench.py --frames $f --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-logging-run 2>/dev/null | tail -1 | python tools/shard_line.py $mode
done
done
