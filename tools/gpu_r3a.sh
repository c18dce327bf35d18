# round 3, first GPU call: the whole -m gpu suite (incl. the new full-size oracle parity tests), the headline bench,
# the per-GPU shares of a fixed global batch of 4096 on one GPU, rocprofv3 kernel stats and the stall-reason PMC passes
O=gpurun_out/r3a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 > $O/gputests.txt
tail -5 $O/gputests.txt
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-900 $O/bench.json
for g in 2 4 8; do python bench.py --shard-of $g --no-cpu-baseline --no-edge-gather > $O/bench_b$((4096/g)).json 2>/dev/null; done
for f in $O/bench_b*.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d.get('ms_per_step'), d.get('value'), {k:v['avg_us'] for k,v in (d.get('kernels') or {}).items()})")"; done
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python bench.py --no-cpu-baseline --no-roofline --no-edge-gather --min-seconds 0.5 > $O/prof_bench.json 2> $O/prof.err
python tools/rocpd_summary.py $(ls $O/prof/*/*.db $O/prof/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt 2>&1; head -10 $O/kernel_stats.txt
rocprofv3 -L > $O/counters_list.txt 2>&1
B="python bench.py --no-graph --no-cpu-baseline --no-roofline --no-edge-gather --steps 40 --warmup 5 --min-seconds 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/st1 -o pmc -- $B > /dev/null 2> $O/st1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/st2 -o pmc -- $B > /dev/null 2> $O/st2.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM -d $O/st3 -o pmc -- $B > /dev/null 2> $O/st3.err
for d in st1 st2 st3; do f=$(ls $O/$d/*/*.db $O/$d/*.db 2>/dev/null | head -1); echo "== $d $f"; python tools/rocpd_summary.py $f > $O/$d.txt 2>&1; tail -3 $O/$d.err; done
grep -c . $O/st1.txt $O/st2.txt $O/st3.txt
