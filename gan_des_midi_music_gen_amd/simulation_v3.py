"""Drop-in surface of the reference's discrete-event simulator for the way its two bridges use it
(SIMULATOR/simulation_v3.py -- byte-identical copies in GAN_DES/ and MMGAN_MIDI_DES/; constructed at
MMGAN_MIDI_DES/matrix_sim_process.py:150-151 and GAN_DES/matrix_sim_process.py:106-108):

    sim = Sim(sim_matrix, distributions, queue_list, seeds=seeds, log_path="logs/", generate_log=True, animation=False,
              record_history=False, logging_mode='Music', max_sim_time=...)
    sim.run(number_of_customers=n)

behind which sits the deterministic C++ core ``gdm_des_run`` (csrc/des_core.hip, host code).
Same constructor / ``run`` signature, same 'Music' log lines in ``logs/simulation.log`` when ``generate_log`` is set,
same consumption of numpy's GLOBAL legacy random stream (routing draws: simulation_v3.py:57,62) and of the per-node
``RandomState`` streams -- under the same seeds the event sequence is the reference's, record for record
(tests/golden/des_core.npz was recorded from the reference's own ``Sim``).

Differences, on purpose:
  * the run ends after ``max_events`` processed events (default 200 000) instead of after ``max_sim_time`` seconds of
    WALL CLOCK (simulation_v3.py:496-499: the reference's results depend on how fast the machine is; ``max_sim_time`` is
    accepted and ignored);
  * only what the bridges construct is supported: 'normal' distributions, ``logging_mode='Music'``, probability routing
    (no 'queue' / 'branch' nodes, no animation, no metric history) -- anything else raises NotImplementedError;
  * the log records are also kept as arrays (``sim.music_log``), so a caller need not go through the log file.
"""
import ctypes
import os

import numpy as np

from . import _lib

ARRIVAL, DEPARTURE, PROCESSING = 0, 1, 2
KIND_NAMES = ("arrival", "departure", "processing")
EVENT_DTYPE = np.dtype([("value", np.float64), ("event_id", np.int64), ("node", np.int32), ("kind", np.int32)])
STOP_REASONS = ("event list empty", "number_of_customers reached", "max_events reached", "error")


class Sim:
    arrival = 1
    departure = 2

    def __init__(self, adj_matrix, distributions, queue_list, seeds=None, num_runs=None, generate_log=False,
                 log_path='logs/', log_name=None, animation=False, record_history=False, logging_mode='All',
                 max_sim_time=1000, verbose=False, max_events=200000):
        if logging_mode != 'Music':
            raise NotImplementedError("only logging_mode='Music' (what matrix_to_midi / matrix_to_wav use) is built")
        if animation or record_history:
            raise NotImplementedError("animation / metric history of the reference's Sim are out of scope")
        if seeds is not None:
            self.seeds = [int(s) for s in np.asarray(seeds).reshape(-1)]
        elif num_runs is not None:
            raise TypeError("can only concatenate list (not \"int\") to list")      # simulation_v3.py:353, as upstream
        else:
            raise ValueError("Either seeds or num_runs must be provided.")
        self.num_runs = len(self.seeds)
        self.adj_matrix = np.ascontiguousarray(np.asarray(adj_matrix, dtype=np.float64))
        dim = self.adj_matrix.shape[0]
        assert self.adj_matrix.shape == (dim, dim) and len(distributions) == dim and len(queue_list) == dim
        for d in distributions:
            if d[0] != "normal":
                raise NotImplementedError(f"distribution {d[0]!r}: the deterministic core implements 'normal' only")
        # np.float32 parameters (matrix_sim_process.py:72-74) widen exactly, like `vals * scale + loc` in scipy
        self.loc = np.ascontiguousarray([float(d[1]) for d in distributions], dtype=np.float64)
        self.scale = np.ascontiguousarray([float(d[2]) for d in distributions], dtype=np.float64)
        if (self.scale < 0).any():
            raise ValueError("Domain error in arguments. The `scale` parameter must be positive for all distributions")
        self.queue_list = np.ascontiguousarray(queue_list, dtype=np.int32)
        self.distributions = distributions
        self.generate_log = generate_log
        self.logging_mode = logging_mode
        self.max_sim_time = max_sim_time
        self.max_events = int(max_events)
        self.verbose = verbose
        self.log_file = (log_path + ("simulation.log" if log_name is None else log_name)) if generate_log else None
        if self.log_file is not None:
            os.makedirs(os.path.dirname(self.log_file) or ".", exist_ok=True)
            open(self.log_file, "w").close()          # simulation_v3.py:337-341: the old log is emptied on construction
        self.music_log = np.zeros(0, dtype=EVENT_DTYPE)
        self.stop_reason = None

    def _run_once(self, seed, number_of_customers):
        lib = _lib.load()
        dim = self.adj_matrix.shape[0]
        name, key, pos, has_gauss, gauss = np.random.get_state()
        assert name == "MT19937"
        cap = 4 * self.max_events + 4 * dim + 64 if self.max_events > 0 else 1 << 16
        while True:
            k = np.ascontiguousarray(key, dtype=np.uint32).copy()
            c_pos, c_has, c_gauss = ctypes.c_int(int(pos)), ctypes.c_int(int(has_gauss)), ctypes.c_double(float(gauss))
            out = np.zeros(cap, dtype=EVENT_DTYPE)
            n_out, reason = ctypes.c_int64(0), ctypes.c_int(0)
            rc = lib.gdm_des_run(self.adj_matrix.ctypes.data, dim, self.loc.ctypes.data, self.scale.ctypes.data,
                                 self.queue_list.ctypes.data, int(seed), int(number_of_customers), self.max_events,
                                 k.ctypes.data, ctypes.byref(c_pos), ctypes.byref(c_has), ctypes.byref(c_gauss),
                                 out.ctypes.data, cap, ctypes.byref(n_out), ctypes.byref(reason))
            if rc == -3 and n_out.value > cap:        # GDM_EWORKSPACE: retry from the ORIGINAL generator state
                cap = int(n_out.value) + 64
                continue
            if rc != 0:
                msg = lib.gdm_last_error().decode(errors="replace")
                raise ValueError(msg)                 # the reference raises ValueError / KeyError in these cases
            np.random.set_state((name, k, c_pos.value, c_has.value, c_gauss.value))
            return out[:n_out.value], reason.value

    def run(self, number_of_customers=50, use_next_available_server=False):
        logs = []
        for seed in self.seeds:
            log, reason = self._run_once(seed, number_of_customers)
            logs.append(log)
            self.stop_reason = STOP_REASONS[reason]
        self.music_log = np.concatenate(logs) if logs else np.zeros(0, dtype=EVENT_DTYPE)
        if self.log_file is not None:
            with open(self.log_file, "w") as f:
                for v, eid, node, kind in self.music_log:
                    f.write(f"INFO:root:{float(v)!r} - {int(eid)} - {int(node)} - {KIND_NAMES[kind]}\n")
        return self.music_log


def run_spec(spec, max_events=200000, generate_log=False, log_path="logs/"):
    """The DES call of the bridges for one ``matrix_sim_process.DesSpec``: returns (music_log, stop_reason)."""
    sim = Sim(spec.sim_matrix, spec.distributions, spec.queue_list, seeds=spec.seeds, log_path=log_path,
              generate_log=generate_log, animation=False, record_history=False, logging_mode='Music',
              max_sim_time=spec.max_sim_time, max_events=max_events)
    sim.run(number_of_customers=spec.num_customers)
    return sim.music_log, sim.stop_reason
