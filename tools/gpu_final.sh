#!/bin/bash
# The round's record in ONE GPU-box session: parity tests, smoke, bench (+ rocprofv3 kernel stats of the same command), kernel timelines of
# the pipelined and the serialised step, the streaming benches and the timeline of a streaming call.  Usage: tools/gpu_final.sh <tag>
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
tools/gpu_round.sh $TAG tests smoke bench prof
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv   # (the trace below reuses $OUT/prof)
tools/gpu_trace.sh $TAG
mv $OUT/prof.log $OUT/trace_prof.log 2>/dev/null
tools/gpu_trace.sh ${TAG}_serial DFX_STREAMS=0
timeout 300 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 4 16 --calls 1000 > $OUT/stream_ll.jsonl 2> $OUT/stream_ll.err
timeout 300 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 --gating >> $OUT/stream_ll.jsonl 2>> $OUT/stream_ll.err
timeout 300 python tools/bench_stream.py --model df3 --streams 4096 --frames-per-call 1 4 16 --calls 500 > $OUT/stream_df3.jsonl 2> $OUT/stream_df3.err
tools/gpu_trace_stream.sh $TAG
python tools/stream_call_timeline.py $OUT/stream_trace_tail.csv > $OUT/stream_call_timeline.txt
tail -3 $OUT/stream_ll.jsonl | cut -c1-250
du -sh $OUT
