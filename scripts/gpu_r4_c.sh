#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r4c
python scripts/dev/nominal_fuzz_gpu.py 417:1:18,27 63:1:7 447:1:15 99:1:0 261:1:7 339:1:4 387:1:31 409:1:3 453:1:23 486:2:24 124:2:16 228:1:14 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/selected.txt
python scripts/dev/nominal_fuzz_gpu.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c/all.txt; tail -5 gpurun_out/r4c/all.txt
