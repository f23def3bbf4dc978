#!/bin/bash
# Knock-out experiment for the fused conv2 backward: builds (HERE, before gpurun) variant libraries with single phases
# removed (-DGDM_KO_*; results are wrong on purpose) and times the entry points with tools/bench_op.py on the GPU box.
#   build:  tools/ko_sweep.sh build      run (on the GPU box):  tools/ko_sweep.sh run
set -e
cd "$(dirname "$0")/.."
VARIANTS="gather:-DGDM_KO_GATHER expand:-DGDM_KO_EXPAND mfma:-DGDM_KO_MFMA ${GDM_EXTRA_VARIANTS}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    tag=${v%%:*}; flags=${v#*:}
    GDM_BUILD_TAG=ko_$tag GDM_HIPCC_FLAGS="$flags" python -m gan_des_midi_music_gen_amd.build > /dev/null
    echo built ko_$tag "$flags"
  done
else
  echo "== shipped"; python tools/bench_op.py
  for v in $VARIANTS; do
    tag=${v%%:*}
    echo "== ko_$tag"; GDM_LIB_TAG=ko_$tag python tools/bench_op.py | grep -E "fused|bwd_data|bwd_weight"
  done
fi
