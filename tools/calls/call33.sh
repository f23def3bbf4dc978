cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v Warn | tail -40
import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import torch
from gan_des_midi_music_gen_amd import ops, network_tests as NT, synthetic
from gan_des_midi_music_gen_amd.train import MmganTrainer
dev = "cuda"
torch.manual_seed(0)
mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, 50), input_dim=50, output_dim=20, device=dev)
mm.train()
tr = MmganTrainer(mm, compute_dtype="bf16")
d = synthetic.mmgan_inputs(256, 50, seed=1, device=dev)
args = (d["piano_roll"], d["durations"], d["beats"], d["noise1"], d["noise2"], d["fake_a"], d["fake_b"])
for _ in range(5): tr.step(*args, g1_in_a=d["g1_in_a"], g1_in_b=d["g1_in_b"])
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): tr.step(*args, g1_in_a=d["g1_in_a"], g1_in_b=d["g1_in_b"])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host per step us", (t1 - t) / 20 * 1e6, " incl. drain", (t2 - t) / 20 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): tr.step(*args, g1_in_a=d["g1_in_a"], g1_in_b=d["g1_in_b"])
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
PY
