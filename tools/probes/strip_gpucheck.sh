#!/bin/bash
# GPU box, repo root: the strip kernel's checks in one call -- stress (bitwise repeatability over shapes), parity against the per-layer
# path and the oracle, and the C3 tile batch timed against another build (usage: strip_gpucheck.sh [other.so])
timeout -k 10 100 python tools/probes/strip_stress.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-150 &&
timeout -k 10 100 python tools/probes/strip_check.py 2>&1 | grep -v amdgpu | cut -c1-130 &&
timeout -k 10 400 python tools/abn.py ${1:-neural_enhanced_super_resolution_amd/libnesr_hip.so} neural_enhanced_super_resolution_amd/libnesr_hip.so --dtype bf16 --c3 --rounds 4 --iters 2 --env "NESR_STRIP=1;NESR_STRIP=1" 2>&1 | tail -2
