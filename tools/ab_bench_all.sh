# A/B of the current build against another build of libgspx (default: pygsp_amd/_lib/libgspx_old.so) over the
# panel shapes that select the different builds of k_step_tile.  One box: boxes differ by several percent.
OLD=${1:-pygsp_amd/_lib/libgspx_old.so}
for shape in "--dtype f64" "--dtype f32" "--dtype f64 --nsig 32 --vertices 500000" "--dtype f32 --nsig 32 --vertices 500000" "--dtype f64 --nsig 128 --vertices 500000" "--dtype f32 --nsig 128 --vertices 500000" "--dtype f64 --nsig 16 --vertices 1000000"; do bash tools/ab_bench.sh $OLD $shape | tail -4; done
