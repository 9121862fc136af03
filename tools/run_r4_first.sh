set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke rc=$?
python tools/fit500_dump.py split3 > gpurun_out/fit500.log 2>&1
FDCAP_GEMM_SPLIT3=0 python tools/fit500_dump.py fp32 >> gpurun_out/fit500.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_general_k.py -x -q > gpurun_out/t_generalk.log 2>&1; echo generalk rc=$?
for k in 4 8 12; do
  python bench.py --steps 3 --warmup 1 --value-only --lbs-nnz $k > gpurun_out/b_k$k.json 2> gpurun_out/b_k$k.err
  FDCAP_SKIN_VEC=0 python bench.py --steps 3 --warmup 1 --value-only --lbs-nnz $k > gpurun_out/b_k${k}_scalar.json 2>> gpurun_out/b_k$k.err
done
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo gpu rc=$?
tail -3 gpurun_out/*.log; cat gpurun_out/b_k*.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['lbs_weights_per_vertex'], d['value'], d['ms_per_step'])
"
