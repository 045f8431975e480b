#!/bin/bash
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --render-frames 0 2>&1 | tail -1 | cut -c1-200; done
timeout 600 python bench.py --no-cpu-baseline --render-frames 0 --prof-all 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --render-frames 0 --mode events 2>&1 | tail -1 | cut -c1-200
