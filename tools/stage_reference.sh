#!/bin/bash
# Builder-side helper (the container that HAS /root/reference): stage the reference's pure-Python package for ONE GPU
# job, so that the real pygsp and a real MI355X run in one process (VERDICT r4 "Next 3").  The copy lives in
# _ref_stage/ (git-ignored, not gpurun-ignored: it travels with the snapshot like _lib/*.so), is never committed,
# is not on any product path, and is removed with `tools/stage_reference.sh clean` right after the job.
#   tools/stage_reference.sh            # stage
#   gpurun -- 'bash tools/gpu_real_pygsp.sh'
#   tools/stage_reference.sh clean      # remove
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "clean" ]; then rm -rf _ref_stage; echo "removed _ref_stage"; exit 0; fi
[ -d /root/reference/pygsp ] || { echo "no /root/reference here"; exit 1; }
rm -rf _ref_stage && mkdir -p _ref_stage
cp -r /root/reference/pygsp _ref_stage/pygsp
find _ref_stage -name __pycache__ -prune -exec rm -rf {} +
git check-ignore -q _ref_stage || { echo "_ref_stage is not git-ignored: refusing"; rm -rf _ref_stage; exit 1; }
du -sh _ref_stage
