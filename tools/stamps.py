#!/usr/bin/env python3
"""Diagnostic: where does a persistent conv kernel's loop (conv2: fwd / bwd / bww, conv1 forward: c1) spend its cycles?

Build the library with -DGDM_STAMPS (GDM_HIPCC_FLAGS=-DGDM_STAMPS python -m gan_des_midi_music_gen_amd.build, then
restore the normal build), run this on the GPU box: it launches the kernel once and prints the mean per-workgroup
cycles of each stamped phase.  The stamped build is never shipped or benchmarked.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gan_des_midi_music_gen_amd import ops, synthetic, _lib
from gan_des_midi_music_gen_amd.ops import BF16

def read(n_blocks):
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    lib.gdm_debug_read_stamps.restype = ctypes.c_int
    lib.gdm_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = lib.gdm_debug_read_stamps(buf, 1024 * 8)
    assert rc == 0, rc
    return np.array(buf, dtype=np.uint64).reshape(1024, 8)[:n_blocks].astype(np.float64)

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    B, H, W = int(os.environ.get("B", 512)), 128, 256
    dev = "cuda"
    torch.manual_seed(0)
    x = synthetic.spectrogram_batch(B, (H, W), seed=1, device=dev)
    w1 = (torch.randn(16, 1, 2, 2) * 0.1).to(dev); b1 = torch.full((16,), 2.0, device=dev)
    w2 = (torch.randn(32, 16, 3, 3) * 0.05).to(dev); b2 = torch.zeros(32, device=dev)
    p1, code1 = ops.simnn_conv1_fwd(x, w1, b1, BF16)
    pack = ops.simnn_conv2_pack(w2, BF16)
    p2, code2 = ops.simnn_conv2_fwd(p1, pack, b2)
    dp2 = torch.randn_like(p2.float()).to(torch.bfloat16)
    names = {"fwd": ["store->barrier", "issue", "mfma", "epilogue", "end barrier", "prologue", "wait+ds_write", "-"],
             "bwd": ["barrier after expand", "issue (loads 2 steps ahead)", "data-gradient mfma", "loop bookkeeping", "conv1-dW epilogue", "end barrier", "prologue + WAIT FOR THIS STEP'S LOADS", "expand + x planes"],
             "bww": ["barrier", "issue", "mfma", "end barrier", "prologue", "wait for loads", "expand + p1 store", "-"],
             "c1": ["wait for the unit's loads", "4 MFMAs until the result is there", "pool / argmax / ReLU (VALU)", "address + 2 stores", "bookkeeping + next loads", "-", "-", "-"]}[which]
    fn = {"fwd": lambda: ops.simnn_conv2_fwd(p1, pack, b2),
          "bwd": lambda: ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x),
          "bww": lambda: ops.simnn_conv2_bwd_weight(dp2, code2, p1),
          "c1": lambda: ops.simnn_conv1_fwd(x, w1, b1, BF16)}[which]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    n_blocks = int(os.environ.get("NB", 768))
    st = read(n_blocks)
    tot = st.sum(1)
    print(f"{which}: launch {a.elapsed_time(b) * 1e3:.1f} us; per-workgroup total cycles mean {tot.mean():.0f} "
          f"min {tot.min():.0f} max {tot.max():.0f}  (100 MHz-independent shader cycles)")
    for k, nme in enumerate(names):
        print(f"  phase {k} {nme:24s} mean {st[:, k].mean():10.0f}  ({100 * st[:, k].mean() / tot.mean():5.1f} %)")

if __name__ == "__main__":
    main()
