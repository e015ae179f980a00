ulimit -c 0; export HSA_ENABLE_COREDUMP=0
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mh.py -x -q -m gpu > gpurun_out/c5_tests.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/c5_tests.log
timeout -k 10 300 python tools/bench_jit_vs_stream_mh.py > gpurun_out/mhmw_bench.log 2>&1; echo "bench rc $?"; grep -A1 "c5\|hier_scale   C= 65536 k_mh_mw_jit" gpurun_out/mhmw_bench.log
