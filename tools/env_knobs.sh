#!/bin/bash
# runtime knobs re-tested with the current tree (one box, alternating): kernel arguments in device memory, hardware-queue count
for i in 1 2; do for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v B=50', d['ms_per_step'], 'ms')"
  HIP_FORCE_DEV_KERNARG=$v python3 bench.py --no-cpu-baseline --batch 7 --steps 60 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v B=7 ', d['ms_per_step'], 'ms')"
done; done
for q in 2 4 8; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --steps 60 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q B=50', d['ms_per_step'], 'ms')"
done
