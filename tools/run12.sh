#!/bin/bash
# a variant build under the parity + batch suites (LD_PRELOAD) and one single-view bench line
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
LD_PRELOAD=$R/variants/$V/libhgs_rast.so HGS_LIB=$R/variants/$V/libhgs_rast.so timeout 120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -4 > $O/pt_$V.log
tail -3 $O/pt_$V.log
LD_PRELOAD=$R/variants/$V/libhgs_rast.so timeout 60 python bench.py --no-cpu-baseline --no-extra --warmup 5 --steps 60 2>/dev/null > $O/ab_c1_$V.json
python -c "
import json
b=json.load(open('gpurun_out/ab_c1_$V.json')); print('$V','ms %.4f'%b['ms_per_step'],{k:round(v,1) for k,v in b['stage_us'].items()})"
