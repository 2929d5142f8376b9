#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 300 python scripts/dev/pit_stamps_batch.py 80
