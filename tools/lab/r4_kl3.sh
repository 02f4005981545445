#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/lab
timeout -k 10 600 python -m pytest tests/test_gpu_aoadmm.py tests/test_gpu_admm.py tests/test_gpu_fullsize.py -x -q -m gpu -k "kl or KL" > gpurun_out/lab/t_kl.log 2>&1
rc=$?
tail -4 gpurun_out/lab/t_kl.log
[ $rc -ne 0 ] && exit $rc
bash tools/lab/r4_kl2.sh | grep -o 'lib [a-z]*\|"solver": "[a-z_ ]*"\|"ms_per_iter": [0-9.]*\|"kl_vaux": \[[0-9., ]*\]'
