#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/gpu_r4_bn.sh
bash scripts/gpu_r4_z.sh
