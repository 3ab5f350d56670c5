# Fabric-counter calibration on gathers (VERDICT r5 "Next 3"): tools/gather_calibration.py under separate rocprofv3
# --pmc passes (never combined with a trace), then counted / true per row width -> gpurun_out/gather_calib/summary.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/gather_calib
mkdir -p $OUT
python $R/tools/gather_calibration.py > $OUT/plain.json 2> $OUT/plain.err
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_(EA0?_)?(RDREQ|RD_|WRREQ|REQ|READ|HIT|MISS|BUBBLE|TAG)[A-Za-z0-9_]*" | sort -u > $OUT/tcc_counters_available.txt
DEFAULT_PASSES="FETCH_SIZE;TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum;TCC_HIT_sum TCC_MISS_sum;TCC_REQ_sum TCC_READ_sum;TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum;TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum"
IFS=';' read -r -a PASSES <<< "${PROF_PASSES:-$DEFAULT_PASSES}"
for pass in "${PASSES[@]}" ; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- python $R/tools/gather_calibration.py > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/gather_calibration_summary.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
