#!/bin/bash
# Compile csrc/gswm_mm.hip for gfx950 and list registers / scratch of the engine's kernels (no GPU needed).  usage: tools/mm_res.sh [grep pattern]
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -c "$R/a-watermark-for-diffusion-models_amd/csrc/gswm_mm.hip" -o /tmp/mm.o -Rpass-analysis=kernel-resource-usage 2> /tmp/mm_res.txt
grep " error" /tmp/mm_res.txt | head -5
python3 "$R/tools/kernel_resources.py" /tmp/mm_res.txt | grep -E "${1:-Li8E}"
