# rocprofv3 counters of the two-orders-per-launch kernel beside k_step_tile, same process (tools/pair_ladder.py):
#   bash tools/pair_prof.sh <tag> "<pair_ladder args>"      -> gpurun_out/pair_prof_<tag>/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ARGS=$2
export TMPDIR=/tmp GSPX_PAIR_EXPERIMENT=1 GSPX_LIB_PATH=$R/pygsp_amd/_lib/libgspx_exp.so
cd /tmp
OUT=$R/gpurun_out/pair_prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/tools/pair_ladder.py $ARGS > $OUT/stats_run.json 2> $OUT/stats.err
DEFAULT_PASSES="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
IFS=';' read -r -a PASSES <<< "${PROF_PASSES:-$DEFAULT_PASSES}"
for pass in "${PASSES[@]}" ; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python $R/tools/pair_ladder.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
done
PROF_KERNELS=k_cheb_pair,k_step_tile PROF_STATS_ROWS=6 python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
grep -v "^$" $OUT/summary.txt | cut -c1-200 | head -60
