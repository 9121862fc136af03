set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for v in 7; do
  FDC_PK=+ bash tools/build_variant.sh pkx$v -DFDC_PKX=$v > $O/build_pkx$v.log 2>&1 || tail -5 $O/build_pkx$v.log
  timeout 600 python tools/pk_bisect.py 300 pkx$v 2>&1 | grep "^A\.\|library\|Error" | cut -c1-260
done
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 600 python tools/pk_bisect.py 300 pk 2>&1 | grep "^A\.\|^B\.\|library\|Error" | cut -c1-260
