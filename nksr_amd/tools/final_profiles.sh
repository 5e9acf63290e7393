# Everything profiles/ holds for a round, in one call on the GPU box:  bash nksr_amd/tools/final_profiles.sh r04
tag=${1:-rNN}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root && mkdir -p gpurun_out
rm -f gpurun_out/${tag}_parity_report.txt
NKSR_PARITY_REPORT=$root/gpurun_out/${tag}_parity_report.txt timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
tail -2 gpurun_out/${tag}_pytest.log
# the bench line as the driver runs it
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
# per-kernel tables: the configs[4] scene (the headline), the configs[2] cloud through both solves
KSTATS_ROWS=60 KSTATS_TOP=6 timeout 400 bash nksr_amd/tools/kstats.sh ${tag}_scene --no-cloud --no-small-inputs --no-adaptive --no-live-pmc > /dev/null
KSTATS_ROWS=60 KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kprof.sh ${tag}_cloud_fused python -m nksr_amd.tools.prof_cloud 1000000 3 > /dev/null
KSTATS_ROWS=60 KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kprof.sh ${tag}_cloud_csr python -m nksr_amd.tools.prof_cloud 1000000 3 --non-fused > /dev/null
# HBM traffic from the counters (separate --pmc passes, --kernel-trace only): operator probe, CSR SpMV probe, the scene step
timeout 300 python -m nksr_amd.tools.fused_pmc gpurun_out/${tag}_fused_pmc.json > /dev/null 2>&1
timeout 300 python -m nksr_amd.tools.spmv_pmc gpurun_out/${tag}_spmv_pmc.json > /dev/null 2>&1
timeout 900 python -m nksr_amd.tools.scene_pmc gpurun_out/${tag}_scene_fused_pmc.json > gpurun_out/${tag}_scene_fused_pmc.txt 2>&1
# the kernel-rows study: the launches of a scene step replayed (per set / merged, per level), the bare store patterns, the merged kernel's counters
ROWS_PROBE_ONLY=0 timeout 300 python -m nksr_amd.tools.rows_probe 10000000 3 > gpurun_out/${tag}_rows_probe.txt 2>&1
(cd nksr_amd/tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 store_probe.hip -o /tmp/store_probe 2>/dev/null && /tmp/store_probe 12 > $root/gpurun_out/${tag}_store_probe.txt 2>&1)
ROWS_PROBE_ONLY=merged NKSR_PMC_GROUPS=0,1,2,3,5,7 timeout 600 python -m nksr_amd.tools.cmd_pmc gpurun_out/${tag}_rows_merged_pmc.json k_kernel_rows_merged -- python -m nksr_amd.tools.rows_probe 10000000 1 > gpurun_out/${tag}_rows_merged_pmc.txt 2>&1
timeout 300 python -m nksr_amd.tools.cheb_rows_probe > gpurun_out/${tag}_cheb_rows.txt 2>&1
timeout 300 python -m nksr_amd.tools.small_pc_sweep > gpurun_out/${tag}_small_pc_sweep.txt 2>&1
# everything ONE rank of 8 does after its solve, on one GPU (collectives replaced by a dictionary)
timeout 600 python -m nksr_amd.tools.prof_rank_tail 8 3 > gpurun_out/${tag}_rank_tail_8.txt 2>&1
NKSR_TAIL_GRAPH=adaptive timeout 600 python -m nksr_amd.tools.prof_rank_tail 8 3 > gpurun_out/${tag}_rank_tail_8_adaptive.txt 2>&1
# the N = 2 launch path on this one GPU (gloo: two ranks share the device; a protocol run, not a scaling measurement)
NKSR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 2> /dev/null | grep '^{' > gpurun_out/${tag}_bench_two_processes_one_gpu.json
NKSR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 1 --warmup 1 2> /dev/null | grep '^{' > gpurun_out/${tag}_bench_eight_processes_one_gpu.json
NKSR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --dual-graph adaptive 2> /dev/null | grep '^{' > gpurun_out/${tag}_bench_two_processes_one_gpu_adaptive.json
NKSR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 1 --warmup 1 --dual-graph adaptive 2> /dev/null | grep '^{' > gpurun_out/${tag}_bench_eight_processes_one_gpu_adaptive.json
ls -la gpurun_out | grep ${tag}_
