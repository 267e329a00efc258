for sl in 8388608 16777216 33554432 67108864; do RT_WF_SLOTS=$sl timeout -k 10 120 python bench.py --no-cpu-baseline --workload tri_mesh --spp 32 --steps 2 --warmup 1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$sl', j['value'], j['ms_per_step'])" || exit 1; done
