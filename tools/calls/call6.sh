mkdir -p gpurun_out; L=gpurun_out/r2_overlap.log; : > $L
python tools/bench_conv_overlap.py >> $L 2>&1
GDM_BD_CAP=512 GDM_BW_CAP=256 python tools/bench_conv_overlap.py >> $L 2>&1
GDM_BD_CAP=256 GDM_BW_CAP=512 python tools/bench_conv_overlap.py >> $L 2>&1
GDM_BD_CAP=256 GDM_BW_CAP=256 python tools/bench_conv_overlap.py >> $L 2>&1
GDM_LIB_TAG=xw144 GDM_BD_CAP=768 python tools/bench_conv_overlap.py >> $L 2>&1
GDM_LIB_TAG=xw144 GDM_BD_CAP=512 python tools/bench_conv_overlap.py >> $L 2>&1
GDM_LIB_TAG=xw144 GDM_BD_CAP=512 GDM_BW_CAP=256 python tools/bench_conv_overlap.py >> $L 2>&1
GDM_LIB_TAG=xw144 GDM_BD_CAP=1024 python tools/bench_conv_overlap.py >> $L 2>&1
grep caps $L
