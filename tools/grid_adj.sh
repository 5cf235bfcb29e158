#!/bin/bash
# ExSUM step time against the number of workgroups added to the streaming kernel's grid (EXBLAS_GRID_ADJ)
for adj in 0 -1 -3 0 -1 -3 -5 -17; do
  for op in exsum; do
    EXBLAS_GRID_ADJ=$adj python bench.py --op $op --no-cpu-baseline --no-secondary --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('adj $adj $op: step %.4f ms kernel %.4f ms value %.1f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
  done
done
