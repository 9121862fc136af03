# Round-4 auxiliary measurements for DESIGN.md section 6 / 5: shard sizes on one GPU, host issue time per iteration (one GPU and a
# one-rank RCCL group, old and new entry points), skinning-weight sparsity
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
bash tools/shard_probe.sh > $O/r4_shard_probe.txt 2>&1; cat $O/r4_shard_probe.txt
python tools/host_issue_probe.py 128 > $O/r4_host_issue.txt 2>&1
FDCAP_FORCE_EXCHANGE=1 python tools/host_issue_probe.py 128 >> $O/r4_host_issue.txt 2>&1
python tools/host_issue_probe.py 1024 >> $O/r4_host_issue.txt 2>&1
grep "^frames" $O/r4_host_issue.txt
for k in 4 8 12; do
  python bench.py --steps 3 --warmup 1 --value-only --lbs-nnz $k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('lbs_nnz', d['config']['lbs_weights_per_vertex'], 'packed', d['value'], d['ms_per_step'])
"
  FDCAP_SKIN_VEC=0 python bench.py --steps 3 --warmup 1 --value-only --lbs-nnz $k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('lbs_nnz', d['config']['lbs_weights_per_vertex'], 'scalar', d['value'], d['ms_per_step'])
"
done > $O/r4_lbs_nnz.txt 2>&1; cat $O/r4_lbs_nnz.txt
