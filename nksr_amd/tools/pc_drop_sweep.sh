# Sweep of the coarse-block knobs on the 64-chunk scene (run on the GPU box): bash nksr_amd/tools/pc_drop_sweep.sh "<drops>" "<steps>"
for d in ${1:-0 0.002 0.006 0.02}; do
 for st in ${2:-12}; do
  NKSR_PC_DROP=$d NKSR_PC_STEPS=$st timeout 200 python bench.py --scene terrain --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; s=d['stages_s_per_step']
print('drop $d steps $st ms %.1f pcg %.1f asm %.1f iters avg %.2f max %d' % (d['ms_per_step'], s['t_pcg']*1e3, s['t_assemble']*1e3, c['pcg_iters_per_chunk'], c['pcg_iters_max_chunk']))"
 done
done
