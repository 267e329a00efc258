# rate of the wavefront workloads against the number of path pools (RT_WF_POOLS)
for n in 1 2 3 4; do
  for wl in "final --width 1920 --height 1080 --spp 64" "tri_mesh --spp 32" "tri_mesh"; do
    RT_WF_POOLS=$n timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $wl | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('pools $n', j['config']['workload'], round(j['value'],1), round(j['ms_per_step'],1))" || exit 1
  done
done
