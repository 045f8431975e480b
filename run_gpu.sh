#!/bin/bash
python tools/cpu_profile_events.py 2>&1 | grep enqueue
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --render-frames 0 --mode events 2>&1 | tail -1 | cut -c1-160; done
