"""Drop-in surface of the G -> DES bridges' numpy prologue on MI355X.

    matrix_to_midi(gen1_output, gen2_output, adj_size=(32,32), instrument=None, start=0, end=150, count=0,
                   generate=False)                      MMGAN_MIDI_DES/matrix_sim_process.py:15-195
    matrix_to_wav(matrices, size=20, use_same_instrument=None, start=0, end=174, device='cpu')
                                                        GAN_DES/matrix_sim_process.py:17-137

Both reference functions do, per generated sample, (1) a block of numpy arithmetic that turns the generator's matrix
into the constructor arguments of the discrete-event simulator and (2) the simulation / MIDI / audio rendering.  Part
(2) -- simulation_v3.Sim, log parsing, FluidSynth -- is CPU / file / wall-clock bound and out of scope (SURVEY.md
section 2 rows 3, 5-7, 9): it is INJECTED here as ``simulate`` (a callable), exactly where the reference constructs
``Sim``.  Part (1) runs batched on the device (``ops.des_scan`` / ``ops.des_routing``, csrc/des_prologue.hip): the
generator output never leaves HBM as a whole; what crosses to the host is the per-row masks the RNG bookkeeping needs
and the final float64 routing matrices the simulator consumes.

numpy's GLOBAL legacy RNG is part of the reference's behaviour (random sources, the random column that takes a row's
rounding residue, the per-sample reseed ``np.random.seed(np.random.randint(0, 99999, size=1))``); how much of the
stream a draw consumes depends on the data, so the draws are made here on the host, per sample, with the same calls in
the same order -- under the same ``np.random.seed`` the specs are bit-identical to the reference's and the stream ends
at the same position (tests/golden/des_prologue.npz).  Reference quirks kept: matrix_to_midi ALWAYS draws random
sources (its emptiness test at line 42 is always true); matrix_to_wav raises ValueError for more than one thresholded
source (line 30) and IndexError for a thresholded column >= dim (line 67); an all-zero row raises ValueError from
``np.random.choice([])``.
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops


@dataclass
class DesSpec:
    """Arguments the reference hands to ``Sim(sim_matrix, distributions, queue_list, seeds=..., max_sim_time=...)``,
    ``Sim.run(number_of_customers=...)`` and ``process_adjsim_log(instruments=..., note_levels=...)``."""
    sim_matrix: np.ndarray            # (dim, dim) float64: row-stochastic routing, diagonal +1 (source) / -1 (server)
    distributions: list               # dim x ['normal', mean, std] (np.float32 scalars, as upstream)
    queue_list: list                  # [254] * dim
    seeds: np.ndarray                 # shape (1,)
    num_customers: int
    max_sim_time: float
    instruments: np.ndarray           # (dim,) float64 holding integers (np.zeros(dim) upstream), or int array
    note_levels: np.ndarray           # (dim,) float64
    sources: np.ndarray = field(default=None)   # node indices that are sources


def _draw_residue_columns(zero_mask, src, dim):
    """One ``np.random.choice`` per row over the columns that are off-diagonal and non-zero after source zeroing
    (matrix_sim_process.py:101-102 / 85-86).  zero_mask: (dim,) int64 bit patterns."""
    cols = np.empty(dim, dtype=np.int32)
    zm = zero_mask.view(np.uint64)
    shifts = np.arange(dim, dtype=np.uint64)
    for i in range(dim):
        nz = ((zm[i] >> shifts) & np.uint64(1)) == 0
        nz &= ~src
        nz[i] = False
        cols[i] = np.random.choice(np.flatnonzero(nz).tolist())       # ValueError on an empty list, like upstream
    return cols


def _reseed():
    np.random.seed(np.random.randint(0, 99999, size=1))
    return np.random.randint(0, 99999, size=1)


def _check_finite(flags):
    if int(flags.max()) != 0:
        raise ValueError("generated matrix holds non-finite values: the DES prologue is defined for finite inputs only")


def midi_prologue(gen1_output, gen2_output, adj_size=(32, 32), instrument=None):
    """Device-batched head of matrix_to_midi: gen1_output (B,1,S,S), gen2_output (B,n2) device tensors -> [DesSpec]."""
    size = adj_size[0]
    dim = size - 3
    g1 = gen1_output.detach()
    if not g1.is_cuda:
        raise ops.GdmError("matrix_sim_process runs the prologue on a HIP device; move the generator outputs there")
    g1 = g1.float()
    b = g1.shape[0]
    scan = ops.des_scan(g1, size, dim, note_mod=True)
    g2 = gen2_output.detach().float().cpu().numpy()
    inst_h = scan["instruments"].cpu().numpy()
    notes_h = scan["note_levels"].cpu().numpy()
    zmask_h = scan["zero_mask"].cpu().numpy()
    _check_finite(scan["flags"].cpu().numpy())
    src_all = np.zeros((b, dim), dtype=bool)
    cols_all = np.empty((b, dim), dtype=np.int32)
    seeds = []
    for i in range(b):                                                   # global-RNG order of the reference, per sample
        sources = np.random.choice(dim, size=dim // 4, replace=False)   # line 43 (the test at 42 is always true)
        src_all[i, sources] = True
        cols_all[i] = _draw_residue_columns(zmask_h[i], src_all[i], dim)
        seeds.append(_reseed())
    dev = g1.device
    routing = ops.des_routing(g1, size, dim, torch.from_numpy(src_all.astype(np.uint8)).to(dev),
                              torch.from_numpy(cols_all).to(dev)).cpu().numpy()
    specs = []
    for i in range(b):
        p = g2[i]
        d_src = (np.abs(p[1] * 50), np.abs(p[2] * 50))
        d_srv = (np.abs(p[3] * 10), np.abs(p[4] * 10))
        dist = [["normal", *(d_src if src_all[i, k] else d_srv)] for k in range(dim)]
        instruments = inst_h[i].astype(np.float64) if instrument is None else np.array([instrument] * dim)
        specs.append(DesSpec(routing[i], dist, [2 * 127] * dim, seeds[i],
                             max(200, max(1000, int(3000 * p[6]))), min(float(p[5]), 1.0), instruments,
                             notes_h[i].astype(np.float64), np.flatnonzero(src_all[i])))
    return specs


def matrix_to_midi(gen1_output, gen2_output, adj_size=(32, 32), instrument=None, start=0, end=150, count=0,
                   generate=False, simulate=None):
    """Reference signature + ``simulate``: ``simulate(spec, count=..., start=..., end=..., generate=...,
    gen2_tail=...)`` stands in for Sim + process_adjsim_log and returns (roll, durations) as (128, end-start) arrays, or
    None for a failed / timed-out simulation.  Returns (list of (2,128,end-start) float64 arrays, failed_simulations)."""
    if simulate is None:
        raise ops.GdmError("matrix_to_midi: the DES / MIDI back end is outside this package; pass simulate=callable "
                           "(it receives the DesSpec the reference would construct Sim from)")
    start, end = int(start), int(end)
    specs = midi_prologue(gen1_output, gen2_output, adj_size, instrument)
    g2 = gen2_output.detach().float().cpu().numpy()
    midi_rolls, failed = [], 0
    for index, spec in enumerate(specs):
        this_count = count if index == 0 else 1
        output = np.zeros((2, 128, end - start))
        res = simulate(spec, count=this_count, start=start, end=end, generate=generate, gen2_tail=g2[index][10:])
        if res is None or res[0] is None:
            failed += 1
        else:
            output[0], output[1] = res[0], res[1]
        midi_rolls.append(output)
    return midi_rolls, failed


def wav_prologue(matrices, size=20, use_same_instrument=None):
    """Device-batched head of matrix_to_wav: matrices (B,size,size) device tensor -> [DesSpec]."""
    dim = size - 5
    m = matrices.detach() if isinstance(matrices, torch.Tensor) else torch.as_tensor(np.asarray(matrices))
    if not m.is_cuda:
        raise ops.GdmError("matrix_sim_process runs the prologue on a HIP device; move the generated matrices there")
    m = m.float()
    b = m.shape[0]
    scan = ops.des_scan(m, size, dim, threshold=0.75, norm_aux=True)
    thr_h = scan["thr_mask"].cpu().numpy().astype(bool)
    inst_h, notes_h = scan["instruments"].cpu().numpy(), scan["note_levels"].cpu().numpy()
    zmask_h, aux_h = scan["zero_mask"].cpu().numpy(), scan["aux"].cpu().numpy()
    _check_finite(scan["flags"].cpu().numpy())
    src_all = np.zeros((b, dim), dtype=bool)
    cols_all = np.empty((b, dim), dtype=np.int32)
    seeds = []
    for i in range(b):
        hit = np.flatnonzero(thr_h[i])
        if len(hit) == 0:
            sources = np.random.choice(dim, size=size // 8, replace=False)     # line 27
        elif len(hit) == 1:
            sources = hit
        else:
            raise ValueError("The truth value of an array with more than one element is ambiguous (matrix_to_wav keeps "
                             "np.where's tuple: more than one thresholded source cannot be processed, line 30)")
        if sources.max() >= dim:
            raise IndexError(f"index {int(sources.max())} is out of bounds for axis 1 with size {dim}")   # line 67
        src_all[i, sources] = True
        cols_all[i] = _draw_residue_columns(zmask_h[i], src_all[i], dim)
        seeds.append(_reseed())
    dev = m.device
    routing = ops.des_routing(m, size, dim, torch.from_numpy(src_all.astype(np.uint8)).to(dev),
                              torch.from_numpy(cols_all).to(dev)).cpu().numpy()
    specs = []
    for i in range(b):
        r3, r4 = aux_h[i, 0], aux_h[i, 1]
        dist = [["normal", 30 * r3[k], 15 * r4[k]] if src_all[i, k] else ["normal", 5 * r3[k], 3 * r4[k]]
                for k in range(dim)]
        instruments = inst_h[i].astype(np.float64) if use_same_instrument is None else \
            np.array([use_same_instrument] * dim)
        specs.append(DesSpec(routing[i], dist, [2 * 127] * dim, seeds[i], 1000, 0.5, instruments,
                             notes_h[i].astype(np.float64), np.flatnonzero(src_all[i])))
    return specs


def matrix_to_wav(matrices, size=20, use_same_instrument=None, start=0, end=174, device="cpu", simulate=None):
    """Reference signature + ``simulate(spec, index=...)`` standing in for Sim + log->MIDI + FluidSynth + mel
    featuriser: it returns the (128, T) dB spectrogram tensor of one sample.  Returns the stacked (B,128,end-start)
    tensor on ``device``."""
    if simulate is None:
        raise ops.GdmError("matrix_to_wav: the DES / FluidSynth back end is outside this package; pass simulate=callable")
    specs = wav_prologue(matrices, size, use_same_instrument)
    spectrograms = [torch.as_tensor(simulate(spec, index=i)) for i, spec in enumerate(specs)]
    return torch.stack([s[:, start:end] for s in spectrograms]).to(device)
