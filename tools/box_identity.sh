# what kind of card is this box: ids, firmware, memory vendor, power / perf settings - next to a short headline measurement
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for f in --showuniqueid --showmemvendor --showvbios --showperflevel --showprofile --showmaxpower --showfwinfo --showproductname --showmeminfo\ vram --showclkvolt --showpids --showxgmierr --showrasinfo --showretiredpages --showpendingpages --showunreservablepages; do
  rocm-smi $f 2>/dev/null | grep "GPU\[" | head -40
done
echo "--- partition"; rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep "GPU\["
echo "--- headline"; timeout 200 python tools/ab_headline.py f64 | tail -1; timeout 200 python tools/ab_headline.py f32 | tail -1
