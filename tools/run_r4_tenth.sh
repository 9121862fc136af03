set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 900 /tmp/pk_repro 40 > $O/pk_repro.txt 2>&1
tail -5 $O/pk_repro.txt
for v in 7; do
  FDC_PK=+ bash tools/build_variant.sh pkx$v -DFDC_PKX=$v > $O/build_pkx$v.log 2>&1 || tail -5 $O/build_pkx$v.log
  timeout 600 python tools/pk_bisect.py 300 pkx$v 2>&1 | grep "^A\.\|library" | cut -c1-260
done
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 600 python tools/pk_bisect.py 300 pk 2>&1 | grep "^A\.\|^B\.\|library" | cut -c1-260
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or reproducible or checkpoint" > $O/t_defer.log 2>&1; echo defer rc=$?
for f in 1 0 1 0; do FDCAP_DEFER_STEP=$f python bench.py --steps 5 --warmup 1 --value-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('defer=$f', d['value'], d['ms_per_step'])
"; done
