# rate of the wavefront workloads against the persistent intersect grid (RT_WF_IBLOCKS) and the number of pools
for p in ${POOLS:-2 3}; do
for n in ${IBLOCKS:-384 512 640 768 896 1024}; do
  for wl in "final --width 1920 --height 1080 --spp 64" "tri_mesh --spp 32" "tri_mesh"; do
    RT_WF_POOLS=$p RT_WF_IBLOCKS=$n timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $wl | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pools $p iblocks $n', j['config']['workload'], round(j['value'],1), round(j['ms_per_step'],1))" || exit 1
  done
done
done
