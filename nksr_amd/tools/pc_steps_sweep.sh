# Chebyshev chain of the coarse-level block on the 64-chunk scene: steps x interval ratio (and first level 3), run on the GPU box:
#   bash nksr_amd/tools/pc_steps_sweep.sh "6 8 10" "20 40" [level]
for st in ${1:-6 8 10}; do
 for r in ${2:-20 40}; do
  NKSR_PC_LEVEL=${3:-2} NKSR_PC_RATIO=$r NKSR_PC_STEPS=$st timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cloud --no-small-inputs --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; s=d['stages_s_per_step']
print('level ${3:-2} steps $st ratio $r: ms %.1f pcg %.1f asm %.1f iters avg %.2f max %d apply %.2f ms' % (d['ms_per_step'], s['t_pcg']*1e3, s['t_assemble']*1e3, c['pcg_iters_per_chunk'], c['pcg_iters_max_chunk'], d['roofline']['avg_launch_us']/1e3))"
 done
done
