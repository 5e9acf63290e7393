# Does the caching allocator reach a steady state on the 64-chunk scene?  (run on the GPU box)
for conf in "" "expandable_segments:True" "max_split_size_mb:4096"; do
  echo "== PYTORCH_HIP_ALLOC_CONF='$conf'"
  PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_ALLOC_CONF=$conf NKSR_BENCH_STEP_TIMES=1 timeout 250 python bench.py --scene terrain --steps 5 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "^\[step\]\|Error\|error" | head -12
done
