#!/bin/bash
# one short bench run, prints "<label>: utt/s ms/step" (GPU box helper for A/B sweeps)
label=$1; shift
timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], 'utt/s', d['ms_per_step'], 'ms')"
