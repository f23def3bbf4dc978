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
at the same position (tests/golden/des_prologue.npz).

The reference INTERLEAVES the samples: draws of sample i, ``Sim(...).run`` of sample i -- which consumes the same global
stream (simulation_v3.py:57,62: ``np.random.choice(self.children, ...)``) from the per-sample seed --, then the draws of
sample i+1.  ``matrix_to_midi`` / ``matrix_to_wav`` keep that order: the RNG-free scan is ONE batched launch, then per
sample: draws, a one-sample ``des_routing`` launch, ``simulate(spec)``, next sample (tests/golden/des_prologue_rng.npz:
recorded with a stand-in Sim that draws from ``np.random`` the way Sim does).  ``midi_prologue`` / ``wav_prologue``
return ALL specs before any simulation runs (one batched routing launch): equal to the reference only for a back end
that leaves numpy's global stream alone.  Reference quirks kept: matrix_to_midi ALWAYS draws random
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


def _midi_scan(gen1_output, gen2_output, adj_size):
    size = adj_size[0]
    dim = size - 3
    g1 = gen1_output.detach()
    if not g1.is_cuda:
        raise ops.GdmError("matrix_sim_process runs the prologue on a HIP device; move the generator outputs there")
    g1 = g1.float()
    scan = ops.des_scan(g1, size, dim, note_mod=True)
    h = {"g1": g1, "size": size, "dim": dim, "b": g1.shape[0],
         "g2": gen2_output.detach().float().cpu().numpy(), "inst": scan["instruments"].cpu().numpy(),
         "notes": scan["note_levels"].cpu().numpy(), "zmask": scan["zero_mask"].cpu().numpy()}
    _check_finite(scan["flags"].cpu().numpy())
    return h


def _midi_draws(h, i):
    """Sample i's draws from numpy's global stream, in the reference's order (lines 43, 101-102, 119-120)."""
    dim = h["dim"]
    src = np.zeros(dim, dtype=bool)
    sources = np.random.choice(dim, size=dim // 4, replace=False)       # line 43 (the test at 42 is always true)
    src[sources] = True
    cols = _draw_residue_columns(h["zmask"][i], src, dim)
    return src, cols, _reseed()


def _midi_spec(h, i, routing, src, seeds, instrument):
    dim = h["dim"]
    p = h["g2"][i]
    d_src = (np.abs(p[1] * 50), np.abs(p[2] * 50))
    d_srv = (np.abs(p[3] * 10), np.abs(p[4] * 10))
    dist = [["normal", *(d_src if src[k] else d_srv)] for k in range(dim)]
    instruments = h["inst"][i].astype(np.float64) if instrument is None else np.array([instrument] * dim)
    return DesSpec(routing, dist, [2 * 127] * dim, seeds, max(200, max(1000, int(3000 * p[6]))), min(float(p[5]), 1.0),
                   instruments, h["notes"][i].astype(np.float64), np.flatnonzero(src))


def _routing(h, lo, hi, src, cols):
    """des_routing for samples [lo, hi) of the scanned batch: src (n,dim) bool, cols (n,dim) int32 -> (n,dim,dim) f64."""
    dev = h["g1"].device
    return ops.des_routing(h["g1"][lo:hi], h["size"], h["dim"],
                           torch.from_numpy(np.ascontiguousarray(src, dtype=np.uint8)).to(dev),
                           torch.from_numpy(np.ascontiguousarray(cols, dtype=np.int32)).to(dev)).cpu().numpy()


def _batched_specs(h, draws, spec_of):
    b, dim = h["b"], h["dim"]
    src_all = np.zeros((b, dim), dtype=bool)
    cols_all = np.empty((b, dim), dtype=np.int32)
    seeds = []
    for i in range(b):                                                   # global-RNG order of the reference, per sample
        src_all[i], cols_all[i], sd = draws(h, i)
        seeds.append(sd)
    routing = _routing(h, 0, b, src_all, cols_all)
    return [spec_of(h, i, routing[i], src_all[i], seeds[i]) for i in range(b)]


def _interleaved_specs(h, draws, spec_of):
    """The reference's order: a sample's spec is complete (and handed to the caller, who simulates) before the next
    sample draws anything."""
    for i in range(h["b"]):
        src, cols, sd = draws(h, i)
        routing = _routing(h, i, i + 1, src[None], cols[None])[0]
        yield spec_of(h, i, routing, src, sd)


def midi_prologue(gen1_output, gen2_output, adj_size=(32, 32), instrument=None):
    """Device-batched head of matrix_to_midi: gen1_output (B,1,S,S), gen2_output (B,n2) device tensors -> [DesSpec].
    All draws are made before the first spec is returned (see the module docstring)."""
    h = _midi_scan(gen1_output, gen2_output, adj_size)
    return _batched_specs(h, _midi_draws, lambda h_, i, r, src, sd: _midi_spec(h_, i, r, src, sd, instrument))


def matrix_to_midi(gen1_output, gen2_output, adj_size=(32, 32), instrument=None, start=0, end=150, count=0,
                   generate=False, simulate=None):
    """Reference signature + ``simulate``: ``simulate(spec, count=..., start=..., end=..., generate=...,
    gen2_tail=...)`` stands in for Sim + process_adjsim_log and returns (roll, durations) as (128, end-start) arrays, or
    None for a failed / timed-out simulation.  Returns (list of (2,128,end-start) float64 arrays, failed_simulations)."""
    if simulate is None:
        raise ops.GdmError("matrix_to_midi: the DES / MIDI back end is outside this package; pass simulate=callable "
                           "(it receives the DesSpec the reference would construct Sim from)")
    start, end = int(start), int(end)
    h = _midi_scan(gen1_output, gen2_output, adj_size)
    g2 = h["g2"]
    midi_rolls, failed = [], 0
    specs = _interleaved_specs(h, _midi_draws, lambda h_, i, r, src, sd: _midi_spec(h_, i, r, src, sd, instrument))
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


def _wav_scan(matrices, size):
    dim = size - 5
    m = matrices.detach() if isinstance(matrices, torch.Tensor) else torch.as_tensor(np.asarray(matrices))
    if not m.is_cuda:
        raise ops.GdmError("matrix_sim_process runs the prologue on a HIP device; move the generated matrices there")
    m = m.float()
    scan = ops.des_scan(m, size, dim, threshold=0.75, norm_aux=True)
    h = {"g1": m, "size": size, "dim": dim, "b": m.shape[0], "thr": scan["thr_mask"].cpu().numpy().astype(bool),
         "inst": scan["instruments"].cpu().numpy(), "notes": scan["note_levels"].cpu().numpy(),
         "zmask": scan["zero_mask"].cpu().numpy(), "aux": scan["aux"].cpu().numpy()}
    _check_finite(scan["flags"].cpu().numpy())
    return h


def _wav_draws(h, i):
    dim, size = h["dim"], h["size"]
    hit = np.flatnonzero(h["thr"][i])
    if len(hit) == 0:
        sources = np.random.choice(dim, size=size // 8, replace=False)     # line 27
    elif len(hit) == 1:
        sources = hit
    else:
        raise ValueError("The truth value of an array with more than one element is ambiguous (matrix_to_wav keeps "
                         "np.where's tuple: more than one thresholded source cannot be processed, line 30)")
    if sources.max() >= dim:
        raise IndexError(f"index {int(sources.max())} is out of bounds for axis 1 with size {dim}")   # line 67
    src = np.zeros(dim, dtype=bool)
    src[sources] = True
    cols = _draw_residue_columns(h["zmask"][i], src, dim)
    return src, cols, _reseed()


def _wav_spec(h, i, routing, src, seeds, use_same_instrument):
    dim = h["dim"]
    r3, r4 = h["aux"][i, 0], h["aux"][i, 1]
    dist = [["normal", 30 * r3[k], 15 * r4[k]] if src[k] else ["normal", 5 * r3[k], 3 * r4[k]] for k in range(dim)]
    instruments = h["inst"][i].astype(np.float64) if use_same_instrument is None else \
        np.array([use_same_instrument] * dim)
    return DesSpec(routing, dist, [2 * 127] * dim, seeds, 1000, 0.5, instruments, h["notes"][i].astype(np.float64),
                   np.flatnonzero(src))


def wav_prologue(matrices, size=20, use_same_instrument=None):
    """Device-batched head of matrix_to_wav: matrices (B,size,size) device tensor -> [DesSpec].  All draws are made
    before the first spec is returned (see the module docstring)."""
    h = _wav_scan(matrices, size)
    return _batched_specs(h, _wav_draws, lambda h_, i, r, src, sd: _wav_spec(h_, i, r, src, sd, use_same_instrument))


def matrix_to_wav(matrices, size=20, use_same_instrument=None, start=0, end=174, device="cpu", simulate=None):
    """Reference signature + ``simulate(spec, index=...)`` standing in for Sim + log->MIDI + FluidSynth + mel
    featuriser: it returns the (128, T) dB spectrogram tensor of one sample.  Returns the stacked (B,128,end-start)
    tensor on ``device``."""
    if simulate is None:
        raise ops.GdmError("matrix_to_wav: the DES / FluidSynth back end is outside this package; pass simulate=callable")
    h = _wav_scan(matrices, size)
    specs = _interleaved_specs(h, _wav_draws, lambda h_, i, r, src, sd: _wav_spec(h_, i, r, src, sd, use_same_instrument))
    spectrograms = [torch.as_tensor(simulate(spec, index=i)) for i, spec in enumerate(specs)]
    return torch.stack([s[:, start:end] for s in spectrograms]).to(device)
