#!/bin/bash
# Ablation builds of the attention kernel (results wrong by design): side libraries under tools/ubench/bin/, timed by tools/attn_bench.py through GSWM_LIB.
# Which part of the flash-attention loop is the time?   usage: bash tools/attn_ablate.sh build   (here, cross-compiles)  |  bash tools/attn_ablate.sh run   (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/a-watermark-for-diffusion-models_amd/csrc
B=$R/tools/ubench/bin
VARIANTS=${VARIANTS:-"BASE NOEXP NOQK NOPV NORESCALE NOEXP_NOPV NOQK_NOPV"}
if [ "$1" = "build" ]; then
  mkdir -p $B
  for v in $VARIANTS; do
    defs=""; for d in ${v//_/ }; do case $d in BASE) ;; W3) defs="$defs -DATTN_MINWAVES=3";; W4) defs="$defs -DATTN_MINWAVES=4";; NOSTAGE) defs="$defs -DATTN_ABL_NOSTAGE";; *) defs="$defs -DATTN_ABL_$d";; esac; done
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $defs -c $C/gswm_attn.hip -o /tmp/attn_$v.o &
  done; wait
  for v in $VARIANTS; do
    hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/gswm.map $C/build/gswm_kernels.o $C/build/gswm_conv.o $C/build/gswm_image.o /tmp/attn_$v.o $C/build/gswm_mm.o $C/build/gswm_small.o -o $B/libgswm_attn_$v.so
  done; ls -la $B | grep attn_
else
  for v in $VARIANTS; do echo "== $v"; GSWM_LIB=$B/libgswm_attn_$v.so python3 $R/tools/attn_bench.py 128 2>&1 | grep "^S=4096\|^S=1024"; done
fi
