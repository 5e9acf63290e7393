# Everything profiles/ holds for a round, in one call on the GPU box:  bash nksr_amd/tools/final_profiles.sh r03
tag=${1:-rNN}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root && mkdir -p gpurun_out
rm -f gpurun_out/${tag}_parity_report.txt
NKSR_PARITY_REPORT=$root/gpurun_out/${tag}_parity_report.txt timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1
tail -2 gpurun_out/${tag}_pytest.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kstats.sh ${tag}_fused --no-scale-scene --no-other-mode > /dev/null
KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kstats.sh ${tag}_csr --non-fused --no-scale-scene --no-other-mode > /dev/null
KSTATS_TAIL_MS=${SCENE_TAIL_MS:-420} KGAPS_TAIL_MS=${SCENE_TAIL_MS:-420} KSTATS_TOP=6 timeout 300 bash nksr_amd/tools/kstats.sh ${tag}_scene --scene terrain --steps 2 > /dev/null
timeout 400 python -m nksr_amd.tools.scene_pmc gpurun_out/${tag}_scene_fused_pmc.json > gpurun_out/${tag}_scene_fused_pmc.txt 2>&1
ls -la gpurun_out | grep ${tag}_
