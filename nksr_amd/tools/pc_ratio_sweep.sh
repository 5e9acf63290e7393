# Sweep of the Chebyshev interval ratio (lambda_max / lambda_min) of the coarse block on the 64-chunk scene (run on the GPU box)
for r in ${1:-30 60 200}; do
 for st in ${2:-12}; do
  NKSR_PC_RATIO=$r NKSR_PC_STEPS=$st timeout 200 python bench.py --scene terrain --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; s=d['stages_s_per_step']
print('ratio $r steps $st ms %.1f pcg %.1f iters avg %.2f max %d' % (d['ms_per_step'], s['t_pcg']*1e3, c['pcg_iters_per_chunk'], c['pcg_iters_max_chunk']))"
 done
done
