#!/bin/bash
# run ON the GPU box (gpurun): kernel-trace summary of the bench command, then the counter passes, each in its own run
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu > $O/bench_under_trace.json 2> $O/trace_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_frame -- python3 $R/tools/prof_shade.py frame > $O/trace_frame.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/prof_shade.py full > $d.log 2>&1
done
cd $R && python3 bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.log
echo done
