# PMC passes of the headline step (separate rocprofv3 runs, counters only: no trace domains next to --pmc)
mkdir -p gpurun_out/r2i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-graph --no-cpu-baseline --no-roofline --steps 40 --warmup 5 --min-seconds 0"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2i/fetch -o pmc -- $B > /dev/null 2> gpurun_out/r2i/fetch.err
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2i/write -o pmc -- $B > /dev/null 2> gpurun_out/r2i/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/r2i/mfma -o pmc -- $B > /dev/null 2> gpurun_out/r2i/mfma.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d gpurun_out/r2i/mops -o pmc -- $B > /dev/null 2> gpurun_out/r2i/mops.err
for d in fetch write mfma mops; do f=$(ls gpurun_out/r2i/$d/*/*.db gpurun_out/r2i/$d/*.db 2>/dev/null | head -1); echo "== $d $f"; python tools/rocpd_summary.py $f > gpurun_out/r2i/$d.txt 2>&1; tail -12 gpurun_out/r2i/$d.txt; done
python tools/make_traffic_json.py $(ls gpurun_out/r2i/fetch/*/*.db gpurun_out/r2i/fetch/*.db 2>/dev/null | head -1) $(ls gpurun_out/r2i/write/*/*.db gpurun_out/r2i/write/*.db 2>/dev/null | head -1) > gpurun_out/r2i/hbm_traffic.json
python tools/predict_latency.py > gpurun_out/r2i/predict_latency.txt 2>&1; cat gpurun_out/r2i/predict_latency.txt | grep -v amdgpu
