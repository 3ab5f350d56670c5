# zones against the address-translation counters: gpurun -- 'bash tools/gpu_zone_tlb.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/${ZONE_OUT:-zone_tlb}
mkdir -p $O
rocm-smi --showuniqueid | grep -i "unique id" > $O/card.txt
i=0
# ZONE_PASSES: ';'-separated counter sets (one rocprofv3 --pmc run each); default: the address-translation counters
DEFAULT="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum;TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum;GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
IFS=';' read -r -a PASSES <<< "${ZONE_PASSES:-$DEFAULT}"
for pass in "${PASSES[@]}" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d $O/pass$i -o pmc -- python $R/tools/zone_tlb.py 16 16000 > $O/pass$i.json 2> $O/pass$i.err
  echo "== pass $i: $pass"; tail -1 $O/pass$i.json
  python $R/tools/zone_tlb_summary.py $O/pass$i | tee $O/pass$i.summary.txt
  find $O/pass$i -name "*.csv" -size +20M -delete
done
