#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3full2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/pytest_all.log
tail -12 $O/pytest_all.log
bash tools/measure_r03.sh all > $O/measure.log 2>&1
tail -5 $O/measure.log | cut -c1-600
