# rocprofv3 evidence for the graph set-up kernels (gspx_graph_setup: k_w_inspect, radix sort, k_locality, Laplacian and
# tile builders, device k-NN) and for the kernels of the pipelined host-array call (VERDICT r3 "Next 3").
# One --kernel-trace --stats run of the default bench.py (with its numpy-in / numpy-out leg) and two --pmc runs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_setup
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu --no-newton --no-mix --no-configs --no-f32 --no-chain --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py $ARGS > $OUT/stats_bench.json 2> $OUT/stats.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
done
PROF_STATS_ROWS=45 PROF_KERNELS="k_w_inspect,k_radix,k_locality,k_tiles,k_lap_build,k_degree,k_internal_build,k_curve_keys,k_knn,k_setup_convert,k_inverse_perm,k_scan,k_lmax_bounds,k_permute,k_step" \
  python $R/tools/prof_summary.py $OUT > $R/gpurun_out/r04_setup_hostpipe_rocprofv3_summary.txt 2>&1
cp $OUT/stats_bench.json $R/gpurun_out/r04_setup_hostpipe_stats_bench.json
rm -rf $OUT
tail -5 $R/gpurun_out/r04_setup_hostpipe_rocprofv3_summary.txt
