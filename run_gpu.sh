#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "saturated or empty_inputs" 2>&1 | tail -40 | cut -c1-200
