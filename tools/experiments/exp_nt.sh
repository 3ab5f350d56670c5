for aux in 0 2 1 3; do
  make -C pygsp_amd/csrc -B EXTRA="-DGSPX_STREAM_AUX=$aux" > /dev/null 2>&1
  echo "== GSPX_STREAM_AUX=$aux"
  python tools/experiments/exp_sweep_dir.py 2>&1 | grep "alt 1 remap 1"
done
make -C pygsp_amd/csrc -B > /dev/null 2>&1
