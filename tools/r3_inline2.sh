ulimit -c 0; export HSA_ENABLE_COREDUMP=0
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_hmc_jit_zoo.py > gpurun_out/inl_hmc.log 2>&1; cat gpurun_out/inl_hmc.log
timeout -k 10 300 python tools/bench_jit_vs_stream_mh.py > gpurun_out/inl_mh2.log 2>&1; grep "mw_jit" gpurun_out/inl_mh2.log
