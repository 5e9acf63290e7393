# Sweep of the coarse block's first level / steps on the 64-chunk scene (run on the GPU box)
for lv in ${1:-2 3}; do
 for st in ${2:-8 12}; do
  NKSR_PC_LEVEL=$lv NKSR_PC_STEPS=$st timeout 200 python bench.py --scene terrain --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; s=d['stages_s_per_step']
print('level $lv steps $st ms %.1f pcg %.1f asm %.1f iters avg %.2f max %d' % (d['ms_per_step'], s['t_pcg']*1e3, s['t_assemble']*1e3, c['pcg_iters_per_chunk'], c['pcg_iters_max_chunk']))"
 done
done
