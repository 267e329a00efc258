# rate of the Cornell bench against the number of sample runs of a megakernel call (RT_MEGA_GROUPS)
for g in 8 16 24 32 64; do RT_MEGA_GROUPS=$g timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('runs $g', round(j['value'],1), round(j['ms_per_step'],1))" || exit 1; done
