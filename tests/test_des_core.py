"""The deterministic C++ discrete-event core (gan_des_midi_music_gen_amd.simulation_v3.Sim -> gdm_des_run, host code)
against tests/golden/des_core.npz: 'Music' log records written by the REFERENCE's own Sim
(SIMULATOR/simulation_v3.py:426-743) on bridge-shaped arguments, recorded by tests/golden/make_golden.py des_core.

Bar: every record identical -- values bit for bit (they are float64 sums of numpy-legacy normal draws), event ids, nodes,
kinds, their order (Python's heapq order for simultaneous events) -- and numpy's GLOBAL random stream left where the
reference leaves it (Sim's routing draws consume it).  Host code: runs without a GPU."""
import numpy as np
import pytest

from gan_des_midi_music_gen_amd import simulation_v3 as sv
from helpers import load_golden

CASES = ("midi0", "midi1", "wav0", "wav1", "hand")
SEEDS = {"midi0": 1000, "midi1": 1001, "wav0": 2000, "wav1": 2001, "hand": 77}


def _sim(g, pre, **kw):
    dist = [["normal", np.float32(a), np.float32(b)] for a, b in g[f"{pre}/dist"]]
    return sv.Sim(g[f"{pre}/sim_matrix"], dist, list(g[f"{pre}/queue_list"]), seeds=g[f"{pre}/seeds"], log_path="logs/",
                  generate_log=False, animation=False, record_history=False, logging_mode='Music', max_sim_time=1.0, **kw)


@pytest.mark.parametrize("pre", CASES)
def test_music_log_is_the_references(pre):
    g = load_golden("des_core.npz")
    sim = _sim(g, pre)
    np.random.seed(SEEDS[pre])
    log = sim.run(number_of_customers=int(g[f"{pre}/customers"]))
    after = np.random.randint(0, 2 ** 31 - 1)
    assert len(log) == len(g[f"{pre}/value"]), (len(log), len(g[f"{pre}/value"]))
    assert np.array_equal(log["kind"], g[f"{pre}/kind"])
    assert np.array_equal(log["node"], g[f"{pre}/node"])
    assert np.array_equal(log["event_id"], g[f"{pre}/event_id"])
    # the log file carries repr(float): parsing it back is exact, so the values must be BIT-identical
    assert np.array_equal(log["value"], g[f"{pre}/value"]), np.abs(log["value"] - g[f"{pre}/value"]).max()
    assert after == int(g[f"{pre}/rng_after"]), "global numpy stream position"
    assert sim.stop_reason == "number_of_customers reached"


def test_log_file_event_cap_and_errors(tmp_path, monkeypatch):
    g = load_golden("des_core.npz")
    monkeypatch.chdir(tmp_path)
    dist = [["normal", np.float32(a), np.float32(b)] for a, b in g["hand/dist"]]
    sim = sv.Sim(g["hand/sim_matrix"], dist, list(g["hand/queue_list"]), seeds=g["hand/seeds"], log_path="logs/",
                 generate_log=True, logging_mode='Music', max_sim_time=0.5)
    np.random.seed(SEEDS["hand"])
    sim.run(number_of_customers=200)
    lines = open("logs/simulation.log").read().splitlines()
    assert len(lines) == len(g["hand/value"])
    # the reference's consumers parse these lines with this pattern (sim_log_to_midi.py:243)
    import re
    pat = re.compile(r"INFO:root:([0-9]*\.[0-9]+|[0-9]+) - ([0-9]*\.[0-9]+|[0-9]+) - ([0-9]*\.[0-9]+|[0-9]+) - (arrival|departure)")
    hits = [pat.match(ln) for ln in lines]
    assert sum(h is not None for h in hits) == int((g["hand/kind"] < 2).sum())
    first = next(h for h in hits if h)
    assert float(first.group(1)) == g["hand/value"][0]
    # event-count cap (the reference's cap is wall-clock seconds): a prefix of the same run
    capped = _sim(g, "wav0", max_events=500)
    np.random.seed(SEEDS["wav0"])
    part = capped.run(number_of_customers=1000)
    assert capped.stop_reason == "max_events reached" and 500 <= len(part) <= 2000
    assert np.array_equal(part["value"], g["wav0/value"][:len(part)])
    # same arguments, same seeds -> same run; a different global seed changes the routing
    a, b = _sim(g, "wav1"), _sim(g, "wav1")
    np.random.seed(1); la = a.run(number_of_customers=120)
    np.random.seed(1); lb = b.run(number_of_customers=120)
    np.random.seed(2); lc = _sim(g, "wav1").run(number_of_customers=120)
    assert np.array_equal(la, lb) and not np.array_equal(la["node"][:len(lc)], lc["node"][:len(la)])
    # unsupported surface fails loudly
    with pytest.raises(NotImplementedError):
        sv.Sim(g["hand/sim_matrix"], [["exponential", 1.0]] * 5, [1] * 5, seeds=[1], logging_mode='Music')
    with pytest.raises(NotImplementedError):
        sv.Sim(g["hand/sim_matrix"], dist, [1] * 5, seeds=[1], logging_mode='All')
    # a customer routed to a source node: KeyError upstream, an error here
    bad = g["hand/sim_matrix"].copy()
    bad[1, 0], bad[1, 3] = 0.0, 0.5
    with pytest.raises(ValueError):
        sv.Sim(bad, dist, list(g["hand/queue_list"]), seeds=[3], logging_mode='Music').run(number_of_customers=50)
