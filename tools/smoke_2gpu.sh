#!/bin/bash
# tools/smoke_2gpu.sh : first thing to run when two GPUs are visible - the RCCL branch of the view-parallel path
# (tests/test_gpu_batch.py / test_gpu_api_contract.py skip it on one GPU) and a 2-rank bench line.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
python -m pytest tests -m gpu -q -k "rccl or two_gpus" -x
python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --no-extra
