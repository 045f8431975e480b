#!/bin/bash
for wb in 256 384 512 768 1024; do echo "wgrad blocks $wb"; python tools/bench_mlp32.py --wgrad-blocks $wb 2>&1 | grep inference | cut -c1-140; done
