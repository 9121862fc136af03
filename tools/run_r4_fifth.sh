set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or reproducible or checkpoint or trajectory or local_mode or gradients" > $O/t_defer.log 2>&1; echo defer rc=$?
tail -15 $O/t_defer.log
for f in 1 0 1 0; do FDCAP_DEFER_STEP=$f python bench.py --steps 5 --warmup 1 --value-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('defer=$f', d['value'], d['ms_per_step'])
"; done
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip 2>/dev/null && /tmp/valu_issue_probe > $O/valu_issue_probe.txt 2>&1
cut -c1-40 $O/valu_issue_probe.txt; awk '{print $1, $(NF-8), $(NF-7)}' $O/valu_issue_probe.txt
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 900 python tools/pk_bisect.py 300 pk > $O/pk_bisect.txt 2>&1
cat $O/pk_bisect.txt | grep -v Warning
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo gpu rc=$?
tail -5 $O/t_gpu.log
