#!/bin/bash
python -m pytest tests/test_hip_primitives.py tests/test_baselines.py tests/test_anogan_mygan.py -x -q -m gpu -k "upsample or pool or baseline or mygan" 2>&1 | tail -2
python tools/probe/upsample_bench.py 2>/dev/null
