#!/bin/bash
# GPU box: the -m gpu test suite, then (unless the tests were killed) the default bench line.  Logs under gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 ${T_TESTS:-700} python -m pytest tests -m gpu -q -s -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/tests.log 2>&1
rc=$?
echo "pytest exit $rc"; tail -n 25 gpurun_out/tests.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: not starting the bench"; exit $rc; fi
timeout -k 10 ${T_BENCH:-420} python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
rb=$?
echo "bench exit $rb"; cat gpurun_out/bench.json | cut -c1-2500; grep -v Warning gpurun_out/bench.err | tail -n 12 | cut -c1-300
exit $(( rc != 0 ? rc : rb ))
