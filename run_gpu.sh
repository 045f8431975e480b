#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-200
