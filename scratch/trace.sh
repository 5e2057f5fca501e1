# kernel trace of the default bench command -> gpurun_out/trace/run_kernel_trace.csv
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/trace -o run --output-format csv -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/trace/bench.json 2> $R/gpurun_out/trace/log.txt
