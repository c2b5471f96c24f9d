#!/bin/bash
# One GPU call for work on the small-lattice kernel: parity of everything that runs through it, the phase stamps (cold
# and instruction-cache-warm), and the C-ABI timing against the previous commit's library.
set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_batcher.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_anyn.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/small_parity.log 2>&1 || { tail -30 gpurun_out/small_parity.log; exit 1; }
tail -1 gpurun_out/small_parity.log
GPT_LIB=$PWD/gptools_amd/libvar_stamps.so timeout -k 10 300 python scripts/small_stamps.py > gpurun_out/small_stamps.jsonl 2> gpurun_out/small_stamps.err
GPT_LIB=$PWD/gptools_amd/libvar_stamps2.so timeout -k 10 300 python scripts/small_stamps.py > gpurun_out/small_stamps2.jsonl 2>> gpurun_out/small_stamps.err
cat gpurun_out/small_stamps.jsonl gpurun_out/small_stamps2.jsonl
scripts/ab_small_lds.sh
