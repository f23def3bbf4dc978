"""CPU restatement of the DES-matrix prologue -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the numpy head of the reference's two G->DES bridges, up to the point where ``Sim(...)`` is constructed:

  midi_prologue   MMGAN_MIDI_DES/matrix_sim_process.py:15-152 (``matrix_to_midi``; 64x64 matrix, 3 parameter rows,
                  distribution / run-length parameters from the beat generator's 20 outputs)
  wav_prologue    GAN_DES/matrix_sim_process.py:17-110 (``matrix_to_wav``; 20x20 matrix, 5 parameter rows)

Pinned by tests/golden/des_prologue.npz: the reference functions themselves were run (build container) with
``simulation_v3.Sim`` replaced by a recorder of its constructor arguments, under ``np.random.seed``; this file has to
reproduce every recorded value -- routing matrix to the last bit, integers exactly, and the position of numpy's global
legacy RNG stream afterwards.  The arithmetic lives in numpy (reference pin 1.24.4; 2.2 here; the operations used --
abs, float32 products, float64 pairwise row sums, IEEE division, RandomState.choice/randint/seed -- are unchanged).

Reference behaviours kept on purpose:
  * matrix_to_midi tests ``len(sources[0]) == 0 or len(sources[0] == dim)`` (line 42): the second operand is the length
    of a boolean array, truthy whenever the first is false, so the thresholded sources are ALWAYS discarded and
    ``dim // 4`` random sources are drawn (line 43).
  * matrix_to_wav keeps ``np.where``'s tuple when the threshold finds sources (line 26): exactly one works; two or more
    make ``x not in sources`` ambiguous -> ValueError (line 30); a thresholded column >= dim -> IndexError (line 67).
  * a row whose candidates are all zero -> ``np.random.choice([])`` -> ValueError (lines 101-102 / 85-86).
"""
import numpy as np


def _routing(absm, dim, sources):
    """Lines 77-110 (midi) / 62-92 (wav): zero source columns + diagonal, float64 row-normalise, residue to a random
    non-zero off-diagonal column, diagonal +1 / -1.  Consumes ``dim`` draws of the global RNG."""
    sim = absm[:dim, :dim].copy()
    src = np.zeros(dim, dtype=bool)
    src[np.asarray(sources, dtype=np.int64).reshape(-1)] = True
    sim[:, src] = 0.0
    sim[np.arange(dim), np.arange(dim)] = 0.0
    sim = sim.astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        sim = sim / sim.sum(axis=1, keepdims=True)
    sim[np.isnan(sim)] = 0
    for i in range(dim):
        cands = [x for x in range(dim) if x != i and sim[i, x] != 0]
        sim[i, np.random.choice(cands)] += 1 - sim[i].sum()
    sim[np.arange(dim), np.arange(dim)] = np.where(src, 1.0, -1.0)
    return sim, src


def _reseed():
    """Lines 119-120 (midi) / 104-105 (wav)."""
    np.random.seed(np.random.randint(0, 99999, size=1))
    return np.random.randint(0, 99999, size=1)


def midi_prologue(gen1_output, gen2_output, adj_size=(32, 32), instrument=None, on_spec=None):
    """gen1_output (B,1,S,S), gen2_output (B,n2) float32 arrays -> list of dicts (one per sample) with the arguments of
    Sim(...) / Sim.run / process_adjsim_log.  ``on_spec(spec)`` is called where the reference runs the simulation of
    that sample (lines 150-163) -- BEFORE the next sample's draws: Sim consumes the same global numpy stream."""
    g1 = np.asarray(gen1_output, dtype=np.float32)
    g2 = np.asarray(gen2_output, dtype=np.float32)
    size = adj_size[0]
    dim = size - 3
    specs = []
    for b in range(g1.shape[0]):
        m = np.abs(g1[b, 0])
        sources = np.random.choice(dim, size=dim // 4, replace=False)            # line 43 (always, see module doc)
        if instrument is None:
            instruments = np.array([int(m[dim + 1, i] * 126) for i in range(dim)], dtype=np.float64)
        else:
            instruments = np.array([instrument] * dim)
        note_levels = np.array([max(0, int(m[dim + 2, i] * 126) % 128) for i in range(dim)], dtype=np.float64)
        sim, src = _routing(m, dim, sources)
        p = g2[b]
        d_src = (np.abs(p[1] * 50), np.abs(p[2] * 50))
        d_srv = (np.abs(p[3] * 10), np.abs(p[4] * 10))
        dist = [["normal", *(d_src if src[i] else d_srv)] for i in range(dim)]
        seeds = _reseed()
        specs.append({"sim_matrix": sim, "distributions": dist, "queue_list": [2 * 127] * dim, "seeds": seeds,
                      "num_customers": max(200, max(1000, int(3000 * p[6]))), "max_sim_time": min(float(p[5]), 1.0),
                      "instruments": instruments, "note_levels": note_levels})
        if on_spec is not None:
            on_spec(specs[-1])
    return specs


def wav_prologue(matrices, size=20, use_same_instrument=None, on_spec=None):
    """matrices (B,size,size) float32 -> list of dicts like midi_prologue (max_sim_time 0.5, 1000 customers);
    ``on_spec`` as in midi_prologue (lines 108-110)."""
    ms = np.asarray(matrices, dtype=np.float32)
    dim = size - 5
    specs = []
    for b in range(ms.shape[0]):
        m = np.abs(ms[b])
        hit = np.where(m[dim] > 0.75)[0]
        if len(hit) == 0:
            sources = np.random.choice(dim, size=size // 8, replace=False)
        elif len(hit) == 1:
            sources = hit
        else:
            raise ValueError("The truth value of an array with more than one element is ambiguous "
                             "(matrix_to_wav: more than one thresholded source)")
        if use_same_instrument is None:
            instruments = np.array([int(m[dim + 1, i] * 126) for i in range(dim)], dtype=np.float64)
        else:
            instruments = np.array([use_same_instrument] * dim)
        note_levels = np.array([int(m[dim + 2, i] * 126) for i in range(dim)], dtype=np.float64)
        r3 = m[dim + 3] / sum(m[dim + 3])                                       # Python sum(): sequential float32
        r4 = m[dim + 4] / sum(m[dim + 4])
        sim, src = _routing(m, dim, sources)                                    # IndexError if a source is >= dim
        dist = [["normal", 30 * r3[i], 15 * r4[i]] if src[i] else ["normal", 5 * r3[i], 3 * r4[i]] for i in range(dim)]
        seeds = _reseed()
        specs.append({"sim_matrix": sim, "distributions": dist, "queue_list": [2 * 127] * dim, "seeds": seeds,
                      "num_customers": 1000, "max_sim_time": 0.5, "instruments": instruments,
                      "note_levels": note_levels})
        if on_spec is not None:
            on_spec(specs[-1])
    return specs
