# rocprofv3 evidence for BASELINE configs 2 (ER filterbank) and 3 (SBM): kernel stats + PMC passes
# (separate --pmc runs, never combined with a trace: MI355X_MICROARCH.md / gpurun rules).
# usage: bash tools/gpu_prof_configs.sh c2|c3   (PROF_PASSES="a;b c" replaces the default counter passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$CFG
mkdir -p $OUT
ARGS="--no-headline --only-config $CFG --config-reps 1 --config-oracle-cols 0 --no-cpu --calibrate-copy"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py $ARGS > $OUT/stats_bench.json 2> $OUT/stats.err
DEFAULT_PASSES="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum;TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum;SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
IFS=';' read -r -a PASSES <<< "${PROF_PASSES:-$DEFAULT_PASSES}"
for pass in "${PASSES[@]}" ; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
