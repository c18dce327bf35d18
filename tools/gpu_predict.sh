O=gpurun_out/r3b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_rl.py -m gpu -x -q -k "small or predict or rl" 2>&1 | grep -v amdgpu.ids | tail -4
python tools/predict_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/predict_latency.txt
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python tools/predict_latency.py > /dev/null 2> $O/prof.err
python tools/rocpd_summary.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) 2>&1 | grep -E "kernel |k_predict|k_gnn|k_mlp" | tee $O/predict_kernels.txt
