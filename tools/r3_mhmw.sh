ulimit -c 0; export HSA_ENABLE_COREDUMP=0
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_jit.py tests/test_gpu_mh.py -x -q -m gpu > gpurun_out/mhmw_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/mhmw_tests.log
timeout -k 10 300 python tools/bench_mh_nostream.py > gpurun_out/mhns_bench.log 2>&1; echo "bench rc $?"; grep -v "k_mh_jit_steps" gpurun_out/mhns_bench.log
timeout -k 10 300 python tools/bench_jit_vs_stream_mh.py > gpurun_out/mhmw_bench.log 2>&1; echo "bench rc $?"; grep "mw_jit" gpurun_out/mhmw_bench.log
