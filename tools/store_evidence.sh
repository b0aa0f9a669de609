#!/bin/bash
# Copy the files of one tools/r6_regen.sh call (gpurun_out/r6_<new>_*) into profiles/, replacing the previous tag's:
#   bash tools/store_evidence.sh <old tag> <new tag>
set -e
OLD=$1; NEW=$2
for f in gpurun_out/r6_${NEW}_*; do cp "$f" profiles/; done
cp gpurun_out/r6_kernel_avg.json profiles/r6_kernel_avg.json
sed -i "s#gpurun_out/r6_${NEW}_#profiles/r6_${NEW}_#g" profiles/r6_kernel_avg.json
git rm -q --cached profiles/r6_${OLD}_* 2>/dev/null || true
rm -f profiles/r6_${OLD}_*
sed -i "s/r6_${OLD}_/r6_${NEW}_/g; s/r6_regen.sh ${OLD}/r6_regen.sh ${NEW}/g" profiles/README.md DESIGN.md
tail -1 profiles/r6_${NEW}_pytest_gpu.txt
