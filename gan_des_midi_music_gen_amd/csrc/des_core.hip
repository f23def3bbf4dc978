// Deterministic core of the queueing-network discrete-event simulator the reference puts between its generators and
// discriminators (SIMULATOR/simulation_v3.py: Sim.run 426-516, Initialization 518-534, ProcessArrival 536-588,
// ScheduleDeparture 591-612, ProcessDeparture 615-677, get_destination 699-743, FlowBranchOperator 25-74), restricted
// to what the two bridges drive it with (MMGAN_MIDI_DES/matrix_sim_process.py:150-163, GAN_DES/matrix_sim_process.py:
// 106-110): 'normal' service / inter-arrival distributions, logging_mode 'Music', probability routing.
//
// HOST code ("a deterministic, event-count-capped C++ DES core"): the simulation is one
// dependent event chain per sample -- nothing for a GPU.  What makes it a drop-in is that it reproduces the reference's
// random streams bit for bit:
//   * numpy's legacy MT19937 RandomState (per-node generators seeded from RandomState(seed).randint(3, 9999999), and the
//     GLOBAL np.random stream that FlowBranchOperator.randomly_select_child draws from, simulation_v3.py:57,62 -- its
//     state is handed in and out, so the caller's np.random continues exactly where the reference's would),
//   * scipy.stats.norm(loc, scale).rvs(random_state=rng) = rng.standard_normal() * scale + loc (legacy polar gauss),
//   * Python's heapq order for simultaneous events (Event.__lt__ compares times only).
// The reference ends a run on a WALL-CLOCK cap (496-499: results depend on the machine's speed); here the cap is a number
// of processed events.  Output = the 'Music' log records (time - id - node - arrival|departure, service - id - node -
// processing) as numbers; the host wrapper formats them like logging.info did when a log file is wanted.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "gdm_common.h"

#pragma clang fp contract(off)

namespace {

// ---- numpy.random.RandomState (legacy) -------------------------------------------------------------------------------
struct LegacyRng {
  uint32_t key[624];
  int pos;
  int has_gauss;
  double gauss;

  void seed(uint32_t s) {                           // mt19937_seed: init_genrand
    for (int i = 0; i < 624; ++i) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
    }
    pos = 624;
    has_gauss = 0;
    gauss = 0.0;
  }
  void gen() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; ++i) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
    }
    for (; i < 623; ++i) {
      y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
    }
    y = (key[623] & UPPER) | (key[0] & LOWER);
    key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double next_double() {                            // random_sample: 53 bits from two draws
    const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  // randint(low, high) for a range below 2^32: masked rejection on 32-bit draws (_bounded_integers, use_masked)
  int64_t randint(int64_t low, int64_t high) {
    const uint64_t rng = (uint64_t)(high - 1 - low);
    if (rng == 0) return low;
    uint32_t mask = (uint32_t)rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = next32() & mask; } while (v > (uint32_t)rng);
    return low + (int64_t)v;
  }
  double standard_normal() {                        // legacy_gauss: polar Box-Muller with one cached value
    if (has_gauss) {
      const double t = gauss;
      has_gauss = 0;
      gauss = 0.0;
      return t;
    }
    double f, x1, x2, r2;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = std::sqrt(-2.0 * std::log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
};

// scipy.stats.norm(loc, scale).rvs(random_state=rng): no draw at all when scale == 0
inline double norm_rvs(LegacyRng& r, double loc, double scale) {
  if (scale == 0.0) return loc;
  const double z = r.standard_normal();
  return z * scale + loc;
}

// ---- FlowBranchOperator (simulation_v3.py:25-74) -------------------------------------------------------------------------
struct Branch {
  std::vector<int> children;
  std::vector<double> prob;
  bool uniform = false;      // sum(probabilities) != 1 -> np.random.choice(children) without p (line 56-58)
  bool sink = false;         // sum(children) == 0 (line 73): also true when the only destination is node 0
};

Branch make_branch(const double* row, int dim, int self) {
  Branch b;
  std::vector<double> p;
  for (int j = 0; j < dim; ++j) {
    const double pj = (j == self) ? 0.0 : row[j];
    if (pj > 0) {                                    // children / probabilities with non-zero probability (38-40)
      b.children.push_back((row[j] > 0 && j != self) ? j : 0);
      p.push_back(pj);
    }
  }
  double s = 0.0;                                    // Python sum(): left to right, starting from int 0
  for (double v : p) s += v;
  for (double v : p) b.prob.push_back(v / s);        // line 47 (sum() of the un-normalised list every time)
  double s1 = 0.0;
  for (double v : b.prob) s1 += v;
  b.uniform = s1 != 1.0;
  long cs = 0;
  for (int c : b.children) cs += c;
  b.sink = cs == 0;
  return b;
}

// np.random.choice(children[, p]) on the GLOBAL legacy stream
int select_child(const Branch& b, LegacyRng& g, bool& error) {
  const int n = (int)b.children.size();
  if (b.uniform) {
    if (n == 0) { error = true; return -1; }         // "No children available to select from"
    return b.children[(size_t)g.randint(0, n)];
  }
  if (n == 0) { error = true; return -1; }           // np.random.choice([]) raises
  // legacy choice with p: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, random_sample(), side='right')
  std::vector<double> cdf(n);
  double acc = 0.0;
  for (int i = 0; i < n; ++i) { acc += b.prob[i]; cdf[i] = acc; }
  const double last = cdf[n - 1];
  for (int i = 0; i < n; ++i) cdf[i] /= last;
  const double u = g.next_double();
  int lo = 0, hi = n;                                // first index with cdf[i] > u
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return b.children[lo < n ? lo : n - 1];
}

// ---- events and Python's heapq ---------------------------------------------------------------------------------------------
struct Ev {
  int type;            // 1 arrival, 2 departure
  double time;
  int server;
  int source;          // -1 = None
  int64_t id;
  double arrival_time;
};
struct Heap {
  std::vector<Ev> h;
  void push(const Ev& e) {                           // heapq.heappush = append + _siftdown(heap, 0, len - 1)
    h.push_back(e);
    size_t pos = h.size() - 1;
    const Ev item = h[pos];
    while (pos > 0) {
      const size_t parent = (pos - 1) >> 1;
      if (item.time < h[parent].time) { h[pos] = h[parent]; pos = parent; continue; }
      break;
    }
    h[pos] = item;
  }
  Ev pop() {                                         // heapq.heappop: last element to the root, _siftup, then _siftdown
    Ev last = h.back();
    h.pop_back();
    if (h.empty()) return last;
    const Ev ret = h[0];
    h[0] = last;
    const size_t end = h.size();
    size_t pos = 0, child = 1;
    const Ev item = h[0];
    while (child < end) {
      const size_t right = child + 1;
      if (right < end && !(h[child].time < h[right].time)) child = right;
      h[pos] = h[child];
      pos = child;
      child = 2 * pos + 1;
    }
    h[pos] = item;
    while (pos > 0) {                                // _siftdown(heap, 0, pos)
      const size_t parent = (pos - 1) >> 1;
      if (item.time < h[parent].time) { h[pos] = h[parent]; pos = parent; continue; }
      break;
    }
    h[pos] = item;
    return ret;
  }
};

struct Node {
  bool is_source = false;
  double loc = 0, scale = 0;
  LegacyRng rng;
  Branch dest;
  // server state
  int in_service = 0;
  std::vector<Ev> queue;      // FIFO (list.pop(0))
  size_t qhead = 0;
  int64_t delayed = 0;
};

struct Run {
  int dim;
  std::vector<Node> nodes;
  const int32_t* qcap;
  LegacyRng* global;
  Heap fel;
  double clock = 0.0;
  int64_t total_customers = 0;
  gdm_des_event* out;
  int64_t cap, n_out = 0;
  bool overflow = false, error = false;

  void log(double v, int64_t id, int node, int kind) {
    if (n_out < cap) out[n_out] = gdm_des_event{v, id, node, kind};
    else overflow = true;
    ++n_out;
  }
  int destination(int id) {                          // get_destination (699-743) for probability routing
    Node& n = nodes[id];
    if (!n.is_source && n.dest.sink) return -1;
    return select_child(n.dest, *global, error);
  }
  void schedule_departure(int server_id, int64_t event_id) {   // 591-612
    Node& s = nodes[server_id];
    s.in_service = 1;
    double service = 0.0;
    if (s.scale == 0.0 && s.loc <= 0.0) { error = true; return; }      // upstream: `while service_time <= 0` never ends
    while (service <= 0) service = norm_rvs(s.rng, s.loc, s.scale);
    log(service, event_id, server_id, 2);
    fel.push(Ev{2, clock + service, server_id, -1, event_id, 0.0});
  }
  void process_arrival(Ev evt) {                     // 536-588
    const int server_id = evt.server;
    log(clock, evt.id, server_id, 0);
    if (server_id < 0 || nodes[server_id].is_source) { error = true; return; }   // KeyError upstream
    Node& s = nodes[server_id];
    if (s.in_service == 0) {
      schedule_departure(server_id, evt.id);
    } else if ((int64_t)(s.queue.size() - s.qhead) + s.delayed < (int64_t)qcap[server_id]) {
      evt.arrival_time = clock;
      s.queue.push_back(evt);
    }                                                // else: the customer reneges
    if (evt.source >= 0) {
      Node& src = nodes[evt.source];
      const double dt = norm_rvs(src.rng, src.loc, src.scale);
      fel.push(Ev{1, clock + dt, server_id, evt.source, total_customers, 0.0});
      ++total_customers;
    }
  }
  void process_departure(const Ev& evt) {            // 615-677
    const int server_id = evt.server;
    log(clock, evt.id, server_id, 1);
    Node& s = nodes[server_id];
    int next = destination(server_id);
    if (error) return;
    if (next < 0) {                                  // sink-like node: first idle child that is a server (628-633)
      for (int c : s.dest.children)
        if (!nodes[c].is_source && nodes[c].in_service == 0) { next = c; break; }
    }
    if (next >= 0 || s.dest.sink) {
      if (s.queue.size() > s.qhead) {
        const Ev customer = s.queue[s.qhead++];
        if (s.qhead > 64 && s.qhead * 2 > s.queue.size()) {
          s.queue.erase(s.queue.begin(), s.queue.begin() + (long)s.qhead);
          s.qhead = 0;
        }
        schedule_departure(server_id, customer.id);
      } else {
        s.in_service = 0;
      }
      if (!s.dest.sink) process_arrival(Ev{1, clock, next, -1, evt.id, 0.0});
    } else {
      error = true;                                  // queue-type nodes (delayed departures) are not produced by the bridges
    }
  }
};

}  // namespace

extern "C" int gdm_des_run(const double* adj, int dim, const double* loc, const double* scale, const int32_t* queue_cap,
                           int64_t seed, int64_t number_of_customers, int64_t max_events, uint32_t* mt_key, int* mt_pos,
                           int* has_gauss, double* cached_gauss, gdm_des_event* out, int64_t out_capacity, int64_t* n_out,
                           int* stop_reason) {
  GDM_REQUIRE(adj && loc && scale && queue_cap && mt_key && mt_pos && has_gauss && cached_gauss && n_out && stop_reason,
              "gdm_des_run: null pointer");
  GDM_REQUIRE(dim >= 1 && dim <= 4096 && (out || out_capacity == 0) && out_capacity >= 0 && max_events >= 0,
              "gdm_des_run: bad arguments");
  GDM_REQUIRE(seed >= 0 && seed <= 0xffffffffLL, "gdm_des_run: seed must fit 32 bits (numpy's RandomState(seed))");
  GDM_REQUIRE(*mt_pos >= 0 && *mt_pos <= 624, "gdm_des_run: bad generator position");
  LegacyRng global;
  std::memcpy(global.key, mt_key, sizeof(global.key));
  global.pos = *mt_pos;
  global.has_gauss = *has_gauss;
  global.gauss = *cached_gauss;

  Run r;
  r.dim = dim;
  r.qcap = queue_cap;
  r.global = &global;
  r.out = out;
  r.cap = out_capacity;
  r.nodes.resize((size_t)dim);
  for (int i = 0; i < dim; ++i) {
    Node& n = r.nodes[(size_t)i];
    n.is_source = adj[(size_t)i * dim + i] > 0;      // sources: diagonal > 0; servers: diagonal <= 0 (363, 379)
    n.loc = loc[i];
    n.scale = scale[i];
    GDM_REQUIRE(n.scale >= 0, "gdm_des_run: Domain error in arguments (scale < 0 at node %d)", i);
    n.dest = make_branch(adj + (size_t)i * dim, dim, i);
  }
  // per-node generators (451-461): servers first, then sources, each in ascending node order
  LegacyRng seeder;
  seeder.seed((uint32_t)seed);
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < dim; ++i)
      if (r.nodes[(size_t)i].is_source == (pass == 1)) r.nodes[(size_t)i].rng.seed((uint32_t)seeder.randint(3, 9999999));

  // Initialization (518-534): one arrival per source
  for (int i = 0; i < dim && !r.error; ++i) {
    Node& src = r.nodes[(size_t)i];
    if (!src.is_source) continue;
    const double dt = norm_rvs(src.rng, src.loc, src.scale);
    const int next = r.destination(i);
    if (r.error) break;
    r.fel.push(Ev{1, r.clock + dt, next, i, r.total_customers, 0.0});
    ++r.total_customers;
  }
  int reason = 0;                                    // 0 event list empty, 1 customer count reached, 2 event cap, 3 error
  int64_t processed = 0;
  while (!r.error && !r.fel.h.empty()) {
    const Ev evt = r.fel.pop();
    if (r.total_customers > number_of_customers - 1) { reason = 1; break; }
    r.clock = evt.time;
    if (evt.type == 1) r.process_arrival(evt); else r.process_departure(evt);
    if (++processed >= max_events && max_events > 0) { reason = 2; break; }    // (the reference: wall clock, 496-499)
  }
  if (r.error) reason = 3;
  std::memcpy(mt_key, global.key, sizeof(global.key));
  *mt_pos = global.pos;
  *has_gauss = global.has_gauss;
  *cached_gauss = global.gauss;
  *n_out = r.n_out;
  *stop_reason = reason;
  if (r.error) {
    gdm_set_error("gdm_des_run: a node has no destination to select from, or a customer was routed to a source / queue "
                  "node (the reference raises here too)");
    return GDM_EINVAL;
  }
  if (r.overflow) {
    gdm_set_error("gdm_des_run: event buffer too small (%lld records needed)", (long long)r.n_out);
    return GDM_EWORKSPACE;
  }
  return GDM_OK;
}
