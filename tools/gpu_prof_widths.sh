# rocprofv3 evidence for the panel widths that select the narrow / regrouped / padded builds of k_step_tile and the
# padded-panel copies (VERDICT r3 "Next 3"): per width one --kernel-trace --stats run and separate --pmc runs of
# bench.py --nsig W on the headline graph.  usage: bash tools/gpu_prof_widths.sh "1 2 4 5 8 12 16 24" [f64|f32]
R=${GRAFT_REPO_ROOT:-$(pwd)}
WIDTHS=${1:-"1 2 4 5 8 12 16 24"}
DT=${2:-f64}
export PROF_PASSES="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
export PROF_KERNELS="k_step,k_permute,k_combine"
for W in $WIDTHS; do
  bash $R/tools/gpu_prof.sh narrow_${DT}_$W --dtype $DT --nsig $W --no-live-traffic > /dev/null 2>&1
  cp $R/gpurun_out/prof_narrow_${DT}_$W/summary.txt $R/gpurun_out/r04_narrow_${DT}_${W}_rocprofv3_summary.txt
  cp $R/gpurun_out/prof_narrow_${DT}_$W/stats_bench.json $R/gpurun_out/r04_narrow_${DT}_${W}_stats_bench.json
  rm -rf $R/gpurun_out/prof_narrow_${DT}_$W
done
python $R/tools/width_fracs.py $R/gpurun_out $DT $WIDTHS > $R/gpurun_out/r04_narrow_${DT}_table.md
cat $R/gpurun_out/r04_narrow_${DT}_table.md
