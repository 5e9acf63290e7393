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
KSTATS_ROWS=60 KSTATS_TOP=6 timeout 400 bash nksr_amd/tools/kstats.sh ${tag}_scene --no-cloud --no-small-inputs > /dev/null
KSTATS_ROWS=60 KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kprof.sh ${tag}_cloud_fused python -m nksr_amd.tools.prof_cloud 1000000 3 > /dev/null
KSTATS_ROWS=60 KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kprof.sh ${tag}_cloud_csr python -m nksr_amd.tools.prof_cloud 1000000 3 --non-fused > /dev/null
# HBM traffic from the counters (separate --pmc passes, --kernel-trace only): operator probe, CSR SpMV probe, the scene step
timeout 300 python -m nksr_amd.tools.fused_pmc gpurun_out/${tag}_fused_pmc.json > /dev/null 2>&1
timeout 300 python -m nksr_amd.tools.spmv_pmc gpurun_out/${tag}_spmv_pmc.json > /dev/null 2>&1
timeout 600 python -m nksr_amd.tools.scene_pmc gpurun_out/${tag}_scene_fused_pmc.json > gpurun_out/${tag}_scene_fused_pmc.txt 2>&1
# the N = 2 launch path on this one GPU (gloo: two ranks share the device; a protocol run, not a scaling measurement)
NKSR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 > gpurun_out/${tag}_bench_two_processes_one_gpu.json 2> /dev/null
ls -la gpurun_out | grep ${tag}_
