# round 5: per-rank cost of the strong-scaling shards on ONE GPU, for BASELINE config 3 and config 5 (DESIGN section 6's projection table)
cd $GRAFT_REPO_ROOT
echo "# config 3: 1024-frame clip, 500k-pt scene, 500 contact verts -- one rank's share at 1 / 2 / 4 / 8 GPUs"
for f in 1024 512 256 128; do
  python bench.py --frames $f --steps 2 --warmup 1 --value-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3 frames %4d: %.2f ms/step  %.1f us/iteration' % ($f, d['ms_per_step'], d['ms_per_step']*1e3/500))"
done
echo "# config 5: 512-frame clip, 2M-pt scene, 10475 contact verts -- one rank's share at 1 / 2 / 4 / 8 GPUs"
for f in 512 256 128 64; do
  python bench.py --config c5 --frames $f --steps 2 --warmup 1 --value-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 frames %4d: %.2f ms/step  %.1f us/iteration' % ($f, d['ms_per_step'], d['ms_per_step']*1e3/500))"
done
echo "# the exchange's own cost on a one-rank RCCL group (FDCAP_FORCE_EXCHANGE=1; pack + ncclAllGather of 1.5 KB + unpack per iteration)"
for f in 128; do
  FDCAP_FORCE_EXCHANGE=1 python bench.py --frames $f --steps 2 --warmup 1 --value-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3 frames %4d with the exchange tail: %.2f ms/step  %.1f us/iteration' % ($f, d['ms_per_step'], d['ms_per_step']*1e3/500))"
done
