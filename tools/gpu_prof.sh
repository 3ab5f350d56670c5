# rocprofv3 evidence for the bench workload.  usage: bash tools/gpu_prof.sh <tag> [bench args...]
# The kernel trace runs bench.py's default set-up INCLUDING the placement tuning (its launches are in the trace and in the
# per-kernel average of kernel_stats.csv; the summary compares the LAST launches - the timed region - with the HIP events
# of the same launches).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-f64}; shift
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-newton --no-mix --no-e2e --no-configs --no-f32 --calibrate-copy "$@" > $OUT/stats_bench.json 2> $OUT/stats.err
# (counter passes: one rocprofv3 run each, --pmc only - never combined with a trace domain.  PROF_PASSES, a
# ';'-separated list, replaces the default set: e.g. PROF_PASSES="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum")
DEFAULT_PASSES="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum;SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU;SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS;TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
IFS=';' read -r -a PASSES <<< "${PROF_PASSES:-$DEFAULT_PASSES}"
for pass in "${PASSES[@]}" ; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  # (the counter passes skip the placement tuning of the set-up: the bytes a launch moves do not depend on where its panels lie)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-newton --no-mix --no-e2e --no-configs --no-f32 --calibrate-copy --tune-candidates 0 "$@" > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
