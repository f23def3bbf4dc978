# build + time kernel variants selected by -D switches (run on the GPU box)
for v in "-DGDM_BD_PREFETCH=0" "-DGDM_BD_PREFETCH=1"; do
  echo "== variant $v"; rm -rf gan_des_midi_music_gen_amd/csrc/_obj/simnn_disc.o
  GDM_HIPCC_FLAGS="$v" python -m gan_des_midi_music_gen_amd.build > /dev/null 2>&1 && python tools/bench_op.py
done
