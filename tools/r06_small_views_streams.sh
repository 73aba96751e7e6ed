#!/bin/bash
# Round 6 (r06g): episodes in flight for the SMALL-view settings of the reference's -b (ttl.py:389): 8 / 16 / 32 views x streams 3 / 4 / 6 / 8, graph replay.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --no-parity --precision fp16 --sustain-seconds 0 --graph 1"
{
for v in 8 16 32; do
  for s in 3 4 6 8; do
    python3 bench.py $Q --views $v --classes 10 --steps 400 --repeats 3 --streams $s 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('views %3d  streams %d  %8.1f images/s (%.1f-%.1f)  host enqueue %.3f ms/image' % ($v, $s, d['value'], d['value_min'], d['value_max'], d['host_enqueue_ms_per_image']))"
  done
done
} | tee gpurun_out/r06/small_views_streams.txt
