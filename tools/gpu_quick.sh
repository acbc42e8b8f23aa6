#!/bin/bash
# quick GPU check after a kernel change: parity tests, section profile, three bench runs
timeout 400 python -m pytest tests/test_hip_parity.py tests/test_env_gpu.py -m gpu -q -x 2>&1 | tail -3
cp wiki-grx-gym_amd/csrc/libgrx_hip.so /tmp/libgrx_keep.so
timeout 300 python tools/gpu_sections.py 2>&1 | grep -E "relative|obs|total|rows|store|update|reset|reward"
cp /tmp/libgrx_keep.so wiki-grx-gym_amd/csrc/libgrx_hip.so
for i in 1 2 3; do timeout 200 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; done
