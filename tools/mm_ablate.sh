#!/bin/bash
# Ablation builds of the matmul engine (results wrong by design): side libraries under tools/ubench/bin/, run under tools/power_probe.py through GSWM_LIB -- which part of the
# main loop is the POWER (and with it the clock) of the convolutions?   usage: bash tools/mm_ablate.sh build (here, cross-compiles) | bash tools/mm_ablate.sh run (on the GPU box)
# variants: NOMFMA (no v_mfma), NOREADS (no LDS fragment reads), NOA / NOW (12-wave form: no LDS-DMA of the activation / weight pieces), combined by '_'
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/a-watermark-for-diffusion-models_amd/csrc
B=$R/tools/ubench/bin
VARIANTS=${VARIANTS:-"BASE NOMFMA NOREADS NOA_NOW NOREADS_NOA_NOW NOMFMA_NOREADS NOMFMA_NOA_NOW"}
if [ "$1" = "build" ]; then
  mkdir -p $B
  for v in $VARIANTS; do
    defs=""; for d in ${v//_/ }; do case $d in BASE) ;; *) defs="$defs -DMM_ABL_$d";; esac; done
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $defs -c $C/gswm_mm.hip -o /tmp/mm_$v.o &
  done; wait
  for v in $VARIANTS; do
    hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/gswm.map $C/build/gswm_kernels.o $C/build/gswm_conv.o $C/build/gswm_image.o $C/build/gswm_attn.o /tmp/mm_$v.o $C/build/gswm_small.o -o $B/libgswm_mm_$v.so
  done; ls -la $B | grep mm_
else
  for v in $VARIANTS; do echo "== $v"; GSWM_LIB=$B/libgswm_mm_$v.so python3 $R/tools/power_probe.py mm 2>&1 | grep -v "amdgpu.ids\|^sensors\|^idle"; done
fi
