# the memory zones against the LOAD PER L2 CHANNEL: gpurun -- 'bash tools/gpu_zone_channels.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/zone_chan
rm -rf $O; mkdir -p $O
rocm-smi --showuniqueid | grep -i "unique id" > $O/card.txt
DEFAULT="TCC_REQ TCC_WRITE TCC_READ;TCC_BUSY TCC_TAG_STALL;TCC_EA0_WRREQ TCC_EA0_RDREQ TCC_EA0_WRREQ_STALL"
IFS=';' read -r -a PASSES <<< "${ZONE_PASSES:-$DEFAULT}"
i=0
for pass in "${PASSES[@]}" ; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $pass --output-format json -d $O/pass$i -o pmc -- python $R/tools/zone_tlb.py 16 16000 > $O/pass$i.json 2> $O/pass$i.err
  echo "== pass $i: $pass"; tail -1 $O/pass$i.json
  python $R/tools/zone_channels.py $O/pass$i | tee $O/pass$i.summary.txt
  rm -rf $O/pass$i
done
