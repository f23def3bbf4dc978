mkdir -p gpurun_out; L=gpurun_out/r2_stamps_occ.log; : > $L
for cap in 256 512 768; do
  echo "== fused, GDM_BD_CAP=$cap (workgroups per CU: $((cap/256)))" >> $L
  GDM_LIB_TAG=stamps144 GDM_BD_CAP=$cap NB=$cap python tools/stamps.py bwd 2>/dev/null >> $L
done
for cap in 256 512 768 1024; do
  echo "== weight, GDM_BW_CAP=$cap" >> $L
  GDM_LIB_TAG=stamps144 GDM_BW_CAP=$cap NB=$cap python tools/stamps.py bww 2>/dev/null >> $L
done
cat $L
