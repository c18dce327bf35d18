mkdir -p gpurun_out/r2c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2c/gputests.txt
python bench.py > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c/prof -o stats -- python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2c/prof_bench.json 2> gpurun_out/r2c/prof.err
python tools/rocpd_summary.py $(ls gpurun_out/r2c/prof/*/*.db gpurun_out/r2c/prof/*.db 2>/dev/null | head -1) > gpurun_out/r2c/kernel_stats.txt 2>&1
cat gpurun_out/r2c/gputests.txt; cut -c1-600 gpurun_out/r2c/bench.json; head -12 gpurun_out/r2c/kernel_stats.txt
