// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Our own harness around the REAL reference (JohnGBaker/ptmcmc) classes.  It is
// compiled only by oracle/Makefile target `ref` against the reference sources
// where they lie under /root/reference; the resulting binary lives in
// oracle/_ref/ (git-ignored, travels to the GPU box).  No reference source text
// is contained in this file -- it only #includes the reference headers at build
// time and calls the reference's public API.
//
// Sub-commands
//   golden-basic            JSON on stdout: boundary::enforce cases, prior
//                           evaluate_log tables, ladder values, swap log-Hastings.
//   golden-eigen            JSON on stdout: gaussian_prop(covar)'s eigenvector transform and
//                           sigmas for fixed covariances (as its constructor prints them).
//   golden-trace <id>       JSON on stdout: a full parallel_tempering_chains run
//                           with *scripted* proposals and *recorded* RNG tapes, so
//                           that a CPU restatement fed the same tapes must land on
//                           the same states after every PT step.
//   golden-de               JSON on stdout: differential_evolution::draw (standard, snooker, unlikely_alpha, temperature
//                           mixing) on the histories of a real ladder, with the uniforms each draw consumed.
//   golden-ess              JSON on stdout: chain::report_effective_samples on fixed AR(1) series.
//   bench <spec-file>       time parallel_tempering_chains::step() on the
//                           correlated-Gaussian problem described in spec-file;
//                           prints one JSON line (cpu_baseline, kind "reference").
//
// Reference interfaces exercised (file:line under /root/reference):
//   boundary::enforce                     states.cc:11-58
//   mixed_dist_product / evaluate_log     probability_function.cc:219-304, .hh:59
//   bayes_likelihood::basic_setup / register_evaluate_log   bayesian.hh:360-381,536-581
//   MH_chain::step / add_state            chain.cc:966-1022, 916-949
//   parallel_tempering_chains ctor/initialize/set_proposal/step
//                                         chain.cc:1163-1211,1281-1386,1393-1571
//   gaussian_prop(cov)                    proposal_distribution.hh:165-218
//   differential_evolution::draw          proposal_distribution.cc:476-801
//   chain::report_effective_samples       chain.cc:126-643

#include <chrono>
#include <new>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "bayesian.hh"
#include "chain.hh"
#include "probability_function.hh"
#include "proposal_distribution.hh"

shared_ptr<Random> globalRNG;  // the reference declares this extern (probability_function.hh:22)

// ----------------------------------------------------------------------------------------------
// tiny JSON helpers (numbers are printed with 17 significant digits => exact round trip)
// ----------------------------------------------------------------------------------------------
static std::string jnum(double v) {
  char b[64];
  if (std::isnan(v)) return "\"nan\"";
  if (std::isinf(v)) return v > 0 ? "\"inf\"" : "\"-inf\"";
  snprintf(b, sizeof b, "%.17g", v);
  return b;
}
template <class V>
static std::string jarr(const V& v) {
  std::ostringstream s;
  s << "[";
  for (size_t i = 0; i < v.size(); i++) s << (i ? "," : "") << jnum(v[i]);
  s << "]";
  return s.str();
}
static std::string jarr_i(const std::vector<int>& v) {
  std::ostringstream s;
  s << "[";
  for (size_t i = 0; i < v.size(); i++) s << (i ? "," : "") << v[i];
  s << "]";
  return s.str();
}

// deterministic helper stream for *inputs* (not part of any algorithm under test)
struct splitmix {
  unsigned long long s;
  explicit splitmix(unsigned long long seed) : s(seed) {}
  unsigned long long next() {
    unsigned long long z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
  double sym() { return 2 * uni() - 1; }
};

// silence the reference's chatter on cout while we produce JSON there
struct cout_mute {
  std::streambuf* old;
  std::ostringstream sink;
  cout_mute() { old = std::cout.rdbuf(sink.rdbuf()); }
  ~cout_mute() { std::cout.rdbuf(old); }
};

// ----------------------------------------------------------------------------------------------
// golden-basic
// ----------------------------------------------------------------------------------------------
static void golden_boundary(std::ostream& os) {
  struct cfg { int lo, hi; double xmin, xmax; };
  const int O = boundary::open, L = boundary::limit, R = boundary::reflect, W = boundary::wrap;
  std::vector<cfg> cfgs = {
      {O, O, -1, 1},   {L, L, 0, 30},   {L, O, 0, 1},   {O, L, -2, 0.5}, {R, O, 0, 1},
      {O, R, 0, 1},    {R, L, -1, 2},   {L, R, -1, 2},  {W, W, 0, 6.283185307179586},
      {W, W, -1.5, 2.25}, {R, R, 0, 1}, {R, R, -2, 3.5}, {W, W, 2, 2}, {R, R, 1, 1}, {W, O, 0, 1}};
  std::vector<double> xs = {-7.3, -2.0, -1.0, -0.25, 0.0, 0.3, 0.5, 1.0, 1.7, 2.0, 2.25, 3.1, 6.283185307179586,
                            6.5, 12.9, 29.999, 30.0, 30.5, -1e-9, 1e9 + 0.5, -1e9 - 0.25};
  os << "\"boundary\":[";
  bool first = true;
  for (auto& c : cfgs) {
    boundary b(c.lo, c.hi, c.xmin, c.xmax);
    for (double x0 : xs) {
      double x = x0;
      bool ok;
      {
        cout_mute m;
        ok = b.enforce(x);
      }
      os << (first ? "" : ",") << "\n {\"lo\":" << c.lo << ",\"hi\":" << c.hi << ",\"xmin\":" << jnum(c.xmin)
         << ",\"xmax\":" << jnum(c.xmax) << ",\"x\":" << jnum(x0) << ",\"ok\":" << (ok ? 1 : 0)
         << ",\"y\":" << jnum(x) << "}";
      first = false;
    }
  }
  os << "]";
}

static void golden_priors(std::ostream& os) {
  // several mixed_dist_product configurations, each evaluated on a batch of states
  const int uni = mixed_dist_product::uniform, gau = mixed_dist_product::gaussian, pol = mixed_dist_product::polar,
            cpol = mixed_dist_product::copolar, lg = mixed_dist_product::log;
  struct pcfg {
    std::string name;
    std::vector<int> types;
    std::vector<double> centers, halfwidths;
    std::vector<int> blo, bhi;
    std::vector<double> bmin, bmax;
  };
  const double PI = M_PI;
  const int O = boundary::open, L = boundary::limit, W = boundary::wrap, R = boundary::reflect;
  std::vector<pcfg> cfgs;
  cfgs.push_back({"lisa6", {uni, uni, pol, uni, cpol, uni}, {1.667, PI, PI / 2, PI, 0, PI / 2},
                  {1.333, PI, PI / 2, PI, PI / 2, PI / 2}, {L, W, L, W, L, W}, {L, W, L, W, L, W},
                  {0, 0, 0, 0, -PI / 2, 0}, {30, 2 * PI, PI, 2 * PI, PI / 2, PI}});
  cfgs.push_back({"mixed5", {gau, lg, uni, pol, cpol}, {0.5, 3.0, -1.0, 1.0, 0.2}, {2.0, 4.0, 2.5, 0.6, 0.7},
                  {O, L, R, O, O}, {O, O, O, O, L}, {-1e300, 0.75, -3.5, 0, 0}, {1e300, 1e300, 1e300, 0, 0.9}});
  cfgs.push_back({"gauss3", {gau, gau, gau}, {0, 1, -2}, {1, 0.5, 3}, {O, O, O}, {O, O, O}, {0, 0, 0}, {0, 0, 0}});
  {
    pcfg c;
    c.name = "unibox8";
    for (int i = 0; i < 8; i++) {
      c.types.push_back(uni);
      c.centers.push_back(0.1 * i);
      c.halfwidths.push_back(10 + 3 * i);
      c.blo.push_back(O); c.bhi.push_back(O); c.bmin.push_back(0); c.bmax.push_back(0);
    }
    cfgs.push_back(c);
  }
  {
    pcfg c;  // wide gaussian prior in 32-D: exercises the log(prod pdf) underflow
    c.name = "gauss32";
    for (int i = 0; i < 32; i++) {
      c.types.push_back(gau);
      c.centers.push_back(0);
      c.halfwidths.push_back(1.0);
      c.blo.push_back(O); c.bhi.push_back(O); c.bmin.push_back(0); c.bmax.push_back(0);
    }
    cfgs.push_back(c);
  }
  os << "\"priors\":[";
  splitmix g(0xA11CE);
  for (size_t ic = 0; ic < cfgs.size(); ic++) {
    pcfg& c = cfgs[ic];
    int D = c.types.size();
    stateSpace sp(D);
    for (int i = 0; i < D; i++) sp.set_bound(i, boundary(c.blo[i], c.bhi[i], c.bmin[i], c.bmax[i]));
    std::valarray<int> tv(D);
    std::valarray<double> cv(D), hv(D);
    for (int i = 0; i < D; i++) { tv[i] = c.types[i]; cv[i] = c.centers[i]; hv[i] = c.halfwidths[i]; }
    mixed_dist_product prior(&sp, tv, cv, hv);
    os << (ic ? "," : "") << "\n {\"name\":\"" << c.name << "\",\"types\":" << jarr_i(c.types)
       << ",\"centers\":" << jarr(c.centers) << ",\"halfwidths\":" << jarr(c.halfwidths)
       << ",\"blo\":" << jarr_i(c.blo) << ",\"bhi\":" << jarr_i(c.bhi) << ",\"bmin\":" << jarr(c.bmin)
       << ",\"bmax\":" << jarr(c.bmax) << ",\"cases\":[";
    int ncase = (c.name == "gauss32") ? 24 : 40;
    for (int k = 0; k < ncase; k++) {
      std::vector<double> x(D);
      // mostly inside the support, sometimes a bit outside (=> pdf 0 / boundary action)
      double spread = (k % 5 == 4) ? 1.6 : 0.98;
      if (c.name == "gauss32") spread = 0.3 + 0.16 * k;  // walks out to |z|~11 per dim => underflow to -inf
      for (int i = 0; i < D; i++) {
        double hw = c.halfwidths[i];
        if (c.types[i] == lg) x[i] = c.centers[i] * std::exp(g.sym() * spread * std::log(hw));
        else x[i] = c.centers[i] + g.sym() * spread * hw * (c.types[i] == gau ? 3 : 1);
      }
      state s(&sp, x);  // the constructor enforces the boundaries (states.cc:194-199)
      double lp;
      {
        cout_mute m;
        lp = prior.evaluate_log(s);
      }
      os << (k ? "," : "") << "\n  {\"x\":" << jarr(x) << ",\"valid\":" << (s.invalid() ? 0 : 1)
         << ",\"xe\":" << jarr(s.get_params_vector()) << ",\"lprior\":" << jnum(lp) << "}";
    }
    os << "]}";
  }
  os << "]";
}

static void golden_ladder(std::ostream& os) {
  // ladder as built by the parallel_tempering_chains constructor + initialize (chain.cc:1181-1183,1340)
  struct lc { int nt; double tmax; };
  std::vector<lc> ls = {{8, 1e2}, {20, 1e9}, {64, 1e4}, {256, 1e6}, {1024, 1e9}, {2, 50.0}};
  os << "\"ladders\":[";
  for (size_t k = 0; k < ls.size(); k++) {
    std::vector<double> beta;
    {
      cout_mute m;
      stateSpace sp(1);
      std::valarray<double> c(0.0, 1), h(1.0, 1);
      std::valarray<int> t(mixed_dist_product::uniform, 1);
      mixed_dist_product prior(&sp, t, c, h);
      parallel_tempering_chains ptc(ls[k].nt, ls[k].tmax, 0.1, 1);
      ptc.initialize(&prior, &prior, 1);  // likelihood := prior (flat inside the box) is enough here
      for (int i = 0; i < ls[k].nt; i++) beta.push_back(ptc.subchain(i)->invTemp());
    }
    os << (k ? "," : "") << "\n {\"ntemps\":" << ls[k].nt << ",\"tmax\":" << jnum(ls[k].tmax)
       << ",\"invtemps\":" << jarr(beta) << "}";
  }
  os << "]";
}

// ----------------------------------------------------------------------------------------------
// Gaussian target registered through the bayes_likelihood function-pointer surface
// ----------------------------------------------------------------------------------------------
struct gauss_target {
  int D;
  std::vector<double> P;  // precision, row-major
  double like0;
  long ncalls;
};
static double gauss_eval(void* obj, const state& s) {
  gauss_target* g = (gauss_target*)obj;
  g->ncalls++;
  std::valarray<double> x = s.get_params();
  double q = 0;
  for (int i = 0; i < g->D; i++) {
    double r = 0;
    for (int j = 0; j < g->D; j++) r += g->P[i * g->D + j] * x[j];
    q += x[i] * r;
  }
  return g->like0 - 0.5 * q;
}

// scripted proposal: x' = x + delta[rung][ncalls], added exactly the way gaussian_prop adds its
// offset (state::add of a space-less offset state, proposal_distribution.hh:217)
struct tape_prop : public proposal_distribution {
  const std::vector<std::vector<std::vector<double>>>* deltas;  // [rung][call][D]
  const std::vector<std::vector<double>>* hastings;             // [rung][call] scripted log-Hastings ratios, or null (0)
  int rung;
  mutable int* nextrung;
  size_t ncall;
  tape_prop(const std::vector<std::vector<std::vector<double>>>* d, int* nextrung_, const std::vector<std::vector<double>>* h = nullptr)
      : deltas(d), hastings(h), rung(-1), nextrung(nextrung_), ncall(0) {}
  state draw(state& s, chain* caller) override {
    const std::vector<double>& d = (*deltas)[rung][ncall];
    last_type = hastings ? (int)(ncall % 3) : 0;   // (a type code that varies, to check MH_chain's last_type bookkeeping)
    log_hastings = hastings ? (*hastings)[rung][ncall] : 0;
    ncall++;
    return s.add(state(nullptr, d));
  }
  tape_prop* clone() const override {
    tape_prop* c = new tape_prop(*this);
    c->rung = (*nextrung)++;  // parallel_tempering_chains::set_proposal clones in rung order (chain.cc:1368-1372)
    c->ncall = 0;
    return c;
  }
  string show() override { return "TapeProp()"; }
};

static std::vector<double> peek_tape(chain* c, int n) {
  // copy of the chain's private generator => the values the chain WILL draw, without disturbing it
  MotherOfAll* m = dynamic_cast<MotherOfAll*>(c->getPRNG().get());
  if (!m) { fprintf(stderr, "peek_tape: chain RNG is not MotherOfAll\n"); exit(2); }
  MotherOfAll copy(*m);
  std::vector<double> t(n);
  for (int i = 0; i < n; i++) t[i] = copy.Next();
  return t;
}

static int golden_trace(int id) {
  // --- problem definitions -------------------------------------------------------------------
  int D, Nt, nsteps, Ninit = 1;
  double Tmax, swap_rate, step_scale, minPrior = -30, evolve = 0, evolve_cut = -1;
  // compact fixtures (ids 7..9): the scripted offsets are not stored -- the test regenerates them from the helper stream's
  // state ("delta_state", splitmix64 as above) -- and the full states are stored every `xstride`-th step only (llike, lpost,
  // history size on every step: a state that differed would show in its llike)
  bool compact = false, with_hastings = false;
  const int xstride = 5;
  std::vector<std::string> types;
  std::vector<double> centers, scales;
  std::vector<int> blo, bhi;
  std::vector<double> bmin, bmax;
  const int O = boundary::open, L = boundary::limit, W = boundary::wrap, R = boundary::reflect;
  if (id == 1) {  // BASELINE configs[0]: 2-D correlated Gaussian, 8-rung ladder
    D = 2; Nt = 8; nsteps = 160; Tmax = 1e2; swap_rate = 0.1; step_scale = 1.2;
    types = {"uni", "uni"}; centers = {0, 0}; scales = {60, 45};
    blo = {O, O}; bhi = {O, O}; bmin = {0, 0}; bmax = {0, 0};
  } else if (id == 2) {  // many swap attempts per step (Q6 overlap cases), narrow box (out-of-prior proposals)
    D = 3; Nt = 7; nsteps = 160; Tmax = 30; swap_rate = 0.45; step_scale = 2.5;
    types = {"uni", "uni", "uni"}; centers = {0.5, 0, -0.5}; scales = {4, 3, 5};
    blo = {O, O, O}; bhi = {O, O, O}; bmin = {0, 0, 0}; bmax = {0, 0, 0};
  } else if (id == 3) {  // mixed prior + wrap / limit / reflect boundaries
    D = 5; Nt = 6; nsteps = 160; Tmax = 1e3; swap_rate = 0.3; step_scale = 1.5;
    types = {"gauss", "log", "uni", "pol", "cpol"};
    centers = {0.5, 3.0, 1.0, M_PI / 2, 0.0}; scales = {2.0, 4.0, 2.5, M_PI / 2, M_PI / 2};
    blo = {O, L, W, L, R}; bhi = {O, O, W, L, R};
    bmin = {0, 0.75, -1.5, 0, -M_PI / 2}; bmax = {0, 0, 3.5, M_PI, M_PI / 2};
  } else if (id == 4) {  // as 3, but the origin lies inside every `limit` bound (see quirk Q9: in trace 3 it does not,
                         // so there state::add() yields an invalid state and the reference never accepts a move)
    D = 5; Nt = 6; nsteps = 160; Tmax = 1e3; swap_rate = 0.3; step_scale = 0.9;
    types = {"gauss", "log", "uni", "pol", "cpol"};
    centers = {0.5, 3.0, 1.0, M_PI / 2, 0.0}; scales = {2.0, 4.0, 2.5, M_PI / 2, M_PI / 2};
    blo = {O, L, W, L, R}; bhi = {O, O, W, L, R};
    bmin = {0, -0.5, -1.5, 0, -M_PI / 2}; bmax = {0, 0, 3.5, M_PI, M_PI / 2};
  } else if (id == 5) {  // trace 1 with the ladder evolving (evolve_temps, chain.hh:302; pry_temps chain.cc:1809-1846)
    D = 2; Nt = 8; nsteps = 160; Tmax = 1e2; swap_rate = 0.1; step_scale = 1.2; evolve = 0.05;
    types = {"uni", "uni"}; centers = {0, 0}; scales = {60, 45};
    blo = {O, O}; bhi = {O, O}; bmin = {0, 0}; bmax = {0, 0};
  } else if (id == 6) {  // trace 2 (several accepted exchanges per step) with the ladder evolving at the sampler's default rate
    D = 3; Nt = 7; nsteps = 160; Tmax = 30; swap_rate = 0.45; step_scale = 2.5; evolve = 0.01;
    types = {"uni", "uni", "uni"}; centers = {0.5, 0, -0.5}; scales = {4, 3, 5};
    blo = {O, O, O}; bhi = {O, O, O}; bmin = {0, 0, 0}; bmax = {0, 0, 0};
  } else if (id == 11) {  // trace 6 with evolve_temp_lpost_cut = 0: every pry also widens the gaps whose chains' posteriors are out of order (chain.cc:1819-1827)
    D = 3; Nt = 7; nsteps = 160; Tmax = 30; swap_rate = 0.45; step_scale = 2.5; evolve = 0.01; evolve_cut = 0.0;
    types = {"uni", "uni", "uni"}; centers = {0.5, 0, -0.5}; scales = {4, 3, 5};
    blo = {O, O, O}; bhi = {O, O, O}; bmin = {0, 0, 0}; bmax = {0, 0, 0};
  } else if (id == 12) {  // ... and with a tolerance (cut = 1.5), on the mixed-prior problem of trace 4
    D = 5; Nt = 6; nsteps = 160; Tmax = 1e3; swap_rate = 0.3; step_scale = 0.9; evolve = 0.03; evolve_cut = 1.5;
    types = {"gauss", "log", "uni", "pol", "cpol"};
    centers = {0.5, 3.0, 1.0, M_PI / 2, 0.0}; scales = {2.0, 4.0, 2.5, M_PI / 2, M_PI / 2};
    blo = {O, L, W, L, R}; bhi = {O, O, W, L, R};
    bmin = {0, -0.5, -1.5, 0, -M_PI / 2}; bmax = {0, 0, 3.5, M_PI, M_PI / 2};
  } else if (id == 7) {  // BASELINE configs[1]: D=16 Gaussian, 64 temperatures (compact fixture: see below)
    D = 16; Nt = 64; nsteps = 40; Tmax = 1e4; swap_rate = 0.1; step_scale = 0.35; compact = true;
  } else if (id == 8) {  // D=32 (the headline dimension), 64 temperatures
    D = 32; Nt = 64; nsteps = 30; Tmax = 1e6; swap_rate = 0.1; step_scale = 0.25; compact = true;
  } else if (id == 9) {  // the swap phase at the headline ladder length: 1024 rungs, maxswapsperstep = 205
    D = 2; Nt = 1024; nsteps = 8; Tmax = 1e9; swap_rate = 0.1; step_scale = 1.2; compact = true;
  } else if (id == 10) {  // a proposal with a non-zero log-Hastings ratio (MH_chain::step, chain.cc:989-994)
    D = 3; Nt = 6; nsteps = 160; Tmax = 50; swap_rate = 0.3; step_scale = 1.6; with_hastings = true;
    types = {"uni", "gauss", "uni"}; centers = {0.5, 0, -0.5}; scales = {4, 1.5, 5};
    blo = {O, O, R}; bhi = {O, O, R}; bmin = {0, 0, -5.5}; bmax = {0, 0, 4.5};
  } else {
    fprintf(stderr, "unknown trace id %d\n", id);
    return 2;
  }
  if (compact) {  // uniform box prior with open boundaries, as the BASELINE workloads
    for (int i = 0; i < D; i++) {
      types.push_back("uni"); centers.push_back(0.0); scales.push_back(20.0 + (i % 5));
      blo.push_back(O); bhi.push_back(O); bmin.push_back(0); bmax.push_back(0);
    }
  }
  // target: zero-mean correlated Gaussian, precision P = inv(A^T A / D + 0.1 I) built directly as
  // P = B^T B + 0.5 I with small-integer-ish B so the fixture is self-contained
  splitmix g(0xC0FFEE + id);
  gauss_target tgt;
  tgt.D = D; tgt.ncalls = 0; tgt.P.assign(D * D, 0.0);
  {
    std::vector<double> B(D * D);
    for (auto& b : B) b = g.sym();
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++) {
        double a = 0;
        for (int k = 0; k < D; k++) a += B[k * D + i] * B[k * D + j];
        tgt.P[i * D + j] = a + (i == j ? 0.5 : 0.0);
      }
    tgt.like0 = -1.25 * D;
  }
  std::ostringstream js;
  {
    cout_mute mute;
    ProbabilityDist::setSeed(0.012556 + 0.001 * id);
    globalRNG.reset(ProbabilityDist::getPRNG());
    stateSpace space(D);
    std::vector<std::string> names;
    for (int i = 0; i < D; i++) names.push_back("x" + std::to_string(i));
    space.set_names(names);
    for (int i = 0; i < D; i++) space.set_bound(i, boundary(blo[i], bhi[i], bmin[i], bmax[i]));
    bayes_likelihood like;
    like.register_reference_object(&tgt);
    like.register_evaluate_log(gauss_eval);
    like.basic_setup(&space, types, centers, scales);
    const sampleable_probability_function* prior = like.getObjectPrior().get();

    parallel_tempering_chains ptc(Nt, Tmax, swap_rate, 1, false, false, minPrior);
    ptc.initialize(&like, prior, Ninit);
    if (evolve > 0) ptc.evolve_temps(evolve, evolve_cut);

    // scripted proposal offsets
    const unsigned long long delta_state = g.s;
    std::vector<std::vector<std::vector<double>>> deltas(Nt);
    for (int r = 0; r < Nt; r++) {
      double beta = ptc.subchain(r)->invTemp();
      double sc = step_scale / std::sqrt(std::max(beta, 0.02));
      deltas[r].resize(nsteps);
      for (int k = 0; k < nsteps; k++) {
        deltas[r][k].resize(D);
        for (int d = 0; d < D; d++) deltas[r][k][d] = g.sym() * sc;
      }
    }
    std::vector<std::vector<double>> hast(Nt);
    if (with_hastings)
      for (int r = 0; r < Nt; r++) {
        hast[r].resize(nsteps);
        for (int k = 0; k < nsteps; k++) hast[r][k] = (k % 7 == 3) ? 0.0 : 1.5 * g.sym();
      }
    int nextrung = 0;
    tape_prop prop(&deltas, &nextrung, with_hastings ? &hast : nullptr);
    ptc.set_proposal(prop);

    int maxswaps = 1 + 2 * swap_rate * Nt;  // chain.cc:1192
    std::vector<double> pt_tape = peek_tape(&ptc, 3 * maxswaps * nsteps + 8);
    std::vector<std::vector<double>> ch_tape(Nt);
    for (int r = 0; r < Nt; r++) ch_tape[r] = peek_tape(ptc.subchain(r), nsteps + 8);

    js << "{\"id\":" << id << ",\"D\":" << D << ",\"Nt\":" << Nt << ",\"nsteps\":" << nsteps << ",\"Tmax\":" << jnum(Tmax)
       << ",\"swap_rate\":" << jnum(swap_rate) << ",\"maxswaps\":" << maxswaps << ",\"minPrior\":" << jnum(minPrior)
       << ",\"add_every_N\":1,\"evolve_rate\":" << jnum(evolve) << ",\"evolve_lpost_cut\":" << jnum(evolve_cut) << ",\n\"types\":[";
    for (int i = 0; i < D; i++) js << (i ? "," : "") << "\"" << types[i] << "\"";
    js << "],\"centers\":" << jarr(centers) << ",\"scales\":" << jarr(scales) << ",\"blo\":" << jarr_i(blo)
       << ",\"bhi\":" << jarr_i(bhi) << ",\"bmin\":" << jarr(bmin) << ",\"bmax\":" << jarr(bmax)
       << ",\n\"P\":" << jarr(tgt.P) << ",\"like0\":" << jnum(tgt.like0) << ",\n\"invtemps\":[";
    for (int r = 0; r < Nt; r++) js << (r ? "," : "") << jnum(ptc.subchain(r)->invTemp());
    js << "],\n\"init\":[";
    for (int r = 0; r < Nt; r++) {
      chain* c = ptc.subchain(r);
      js << (r ? "," : "") << "\n {\"x\":" << jarr(c->getState().get_params_vector()) << ",\"llike\":" << jnum(c->getLogLike())
         << ",\"lpost\":" << jnum(c->getLogPost()) << ",\"size\":" << c->size() << "}";
    }
    js << "],\n\"pt_tape\":" << jarr(pt_tape) << ",\n\"chain_tapes\":[";
    for (int r = 0; r < Nt; r++) js << (r ? "," : "") << "\n " << jarr(ch_tape[r]);
    if (compact) {
      js << "],\n\"delta_state\":\"" << delta_state << "\",\"step_scale\":" << jnum(step_scale) << ",\"xstride\":" << xstride
         << ",\n\"steps\":[";
    } else {
      js << "],\n\"deltas\":[";
      for (int r = 0; r < Nt; r++) {
        js << (r ? "," : "") << "\n [";
        for (int k = 0; k < nsteps; k++) js << (k ? "," : "") << jarr(deltas[r][k]);
        js << "]";
      }
      if (with_hastings) {
        js << "],\n\"log_hastings\":[";
        for (int r = 0; r < Nt; r++) js << (r ? "," : "") << "\n " << jarr(hast[r]);
      }
      js << "],\n\"steps\":[";
    }
    for (int k = 0; k < nsteps; k++) {
      ptc.step();
      js << (k ? "," : "") << "\n [";
      for (int r = 0; r < Nt; r++) {
        chain* c = ptc.subchain(r);
        js << (r ? "," : "") << "{";
        if (!compact || k % xstride == xstride - 1 || k == nsteps - 1) js << "\"x\":" << jarr(c->getState().get_params_vector()) << ",";
        js << "\"llike\":" << jnum(c->getLogLike()) << ",\"lpost\":" << jnum(c->getLogPost()) << ",\"size\":" << c->size();
        if (!compact) js << ",\"invtemp\":" << jnum(c->invTemp());
        js << "}";
      }
      js << "]";
    }
    // what every rung's add_state calls pushed (MH_chain::lposts / llikes, chain.cc:935-946), raw history order
    js << "],\n\"hist_lpost\":[";
    for (int r = 0; r < Nt; r++) {
      chain* c = ptc.subchain(r);
      js << (r ? "," : "") << "\n [";
      for (int e = 0; e < c->size(); e++) js << (e ? "," : "") << jnum(c->getLogPost(e, true));
      js << "]";
    }
    js << "],\n\"hist_llike\":[";
    for (int r = 0; r < Nt; r++) {
      chain* c = ptc.subchain(r);
      js << (r ? "," : "") << "\n [";
      for (int e = 0; e < c->size(); e++) js << (e ? "," : "") << jnum(c->getLogLike(e, true));
      js << "]";
    }
    if (with_hastings) {
      // the proposal type MH_chain::add_state pushed with every row (chain.cc:943), read back from MH_chain::dumpChain's
      // rows "i lpost llike acceptance_ratio type: ..." (chain.cc:1127); rows 1.. of the raw history (row 0 = the initial state)
      js << "],\n\"hist_type\":[";
      for (int r = 0; r < Nt; r++) {
        std::ostringstream dump;
        dynamic_cast<MH_chain*>(ptc.subchain(r))->dumpChain(dump, 0, 1);
        std::istringstream in(dump.str());
        std::string line;
        js << (r ? "," : "") << "\n [";
        bool first = true;
        while (std::getline(in, line)) {
          if (line.empty() || line[0] == '#') continue;
          std::istringstream ls(line);
          std::string a, b, c2, d2, t;
          ls >> a >> b >> c2 >> d2 >> t;
          if (t.empty()) continue;
          js << (first ? "" : ",") << atoi(t.c_str());
          first = false;
        }
        js << "]";
      }
    }
    js << "],\n\"likelihood_calls\":" << tgt.ncalls << "}\n";
    globalRNG.reset();  // do not let the shared_ptr delete the master generator twice
  }
  std::cout << js.str();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// golden-eigen: gaussian_prop(covar)'s transform (proposal_distribution.hh:165-187).  diagTransform and sigmas are
// private; the constructor prints them (hh:184-186: " Eigenvalues=", " transform=") -- captured here at 17 digits.
// ----------------------------------------------------------------------------------------------
static int golden_eigen() {
  std::ostringstream js;
  js << "{\"cases\":[";
  const int dims[] = {2, 3, 5, 16, 32};
  for (int ic = 0; ic < 5; ic++) {
    const int D = dims[ic];
    splitmix g(0xE16E4 + D);
    Eigen::MatrixXd A(D, D), cov(D, D);
    for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) A(i, j) = g.sym();
    cov = A.transpose() * A / D + 0.1 * Eigen::MatrixXd::Identity(D, D);   // the synthetic covariance's shape
    for (int i = 0; i < D; i++) for (int j = 0; j < i; j++) cov(i, j) = cov(j, i);
    std::string text;
    {
      cout_mute m;
      std::cout.precision(17);
      void* mem = calloc(1, sizeof(gaussian_prop));   // (quirk Q4: the constructor copies its own uninitialised `sigmas`)
      gaussian_prop* gp = new (mem) gaussian_prop(cov);
      (void)gp;
      text = m.sink.str();
    }
    std::cout.precision(6);
    size_t pe = text.find(" Eigenvalues="), pt = text.find(" transform="), ps = text.find(" test=");
    if (pe == std::string::npos || pt == std::string::npos || ps == std::string::npos) { fprintf(stderr, "golden-eigen: constructor printout not found\n"); return 2; }
    std::vector<double> ev, tr, cv;
    { std::istringstream in(text.substr(pe + 13, pt - pe - 13)); double v; while (in >> v) ev.push_back(v); }
    { std::istringstream in(text.substr(pt + 11, ps - pt - 11)); double v; while (in >> v) tr.push_back(v); }
    if ((int)ev.size() != D || (int)tr.size() != D * D) { fprintf(stderr, "golden-eigen: parsed %zu eigenvalues, %zu transform entries for D=%d\n", ev.size(), tr.size(), D); return 2; }
    for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) cv.push_back(cov(i, j));
    // offset = diagTransform * (sigmas o z) (hh:207-213): the factor is diagTransform * diag(sqrt(eigenvalues)), row-major
    std::vector<double> fac(D * D);
    for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) fac[i * D + j] = tr[i * D + j] * std::sqrt(ev[j]);
    js << (ic ? "," : "") << "\n {\"D\":" << D << ",\"cov\":" << jarr(cv) << ",\"eigenvalues\":" << jarr(ev) << ",\"transform\":" << jarr(tr)
       << ",\"factor\":" << jarr(fac) << "}";
  }
  js << "]}\n";
  std::cout << js.str();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// bench: the reference's own classes on the correlated-Gaussian ladder, timed
// ----------------------------------------------------------------------------------------------
// One gaussian_prop(cov_r) per rung, as the GPU workload has them (cython/exampleGaussian.py:88-95,145:
// cov_r = inv(beta_r * invcov + diag((f h_i)^-2)) * 2.38^2 / D).  parallel_tempering_chains::set_proposal clones ONE proposal
// for every rung in rung order (chain.cc:1368-1372), so this wrapper's clone() builds the next rung's gaussian_prop.
struct per_rung_gaussian : public proposal_distribution {
  const std::vector<Eigen::MatrixXd>* covs;
  mutable int* nextrung;
  gaussian_prop* gp;
  per_rung_gaussian(const std::vector<Eigen::MatrixXd>* c, int* next) : covs(c), nextrung(next), gp(nullptr) {}
  state draw(state& s, chain* caller) override {
    state out = gp->draw(s, caller);
    last_type = gp->type();
    log_hastings = gp->log_hastings_ratio();
    return out;
  }
  per_rung_gaussian* clone() const override {
    per_rung_gaussian* c = new per_rung_gaussian(*this);
    Eigen::MatrixXd cov = (*covs)[(*nextrung)++];
    void* mem = calloc(1, sizeof(gaussian_prop));   // (quirk Q4, as below)
    c->gp = new (mem) gaussian_prop(cov);
    return c;
  }
  string show() override { return "PerRungGaussian()"; }
};

static int bench(const char* specfile) {
  // spec file (text): D Nt nsteps Tmax swap_rate seed fbase / then D*D covariance (row-major) / then D prior half-widths
  // fbase > 0: per-rung proposal covariances inv(beta_r invcov + diag((fbase h_i)^-2)) 2.38^2/D; 0: one gaussian_prop(cov 2.38^2/D)
  std::ifstream in(specfile);
  if (!in) { fprintf(stderr, "cannot open %s\n", specfile); return 2; }
  int D, Nt, nsteps;
  double Tmax, swap_rate, seed, fbase;
  in >> D >> Nt >> nsteps >> Tmax >> swap_rate >> seed >> fbase;
  Eigen::MatrixXd cov(D, D);
  for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) in >> cov(i, j);
  std::vector<double> hw(D);
  for (int i = 0; i < D; i++) in >> hw[i];
  if (!in) { fprintf(stderr, "short spec file\n"); return 2; }
  Eigen::MatrixXd P = cov.inverse();
  gauss_target tgt;
  tgt.D = D; tgt.ncalls = 0; tgt.P.resize(D * D);
  for (int i = 0; i < D; i++) for (int j = 0; j < D; j++) tgt.P[i * D + j] = P(i, j);
  tgt.like0 = -0.5 * (D * std::log(2 * M_PI) + std::log(cov.determinant()));
  double secs, acc0 = 0;
  long nmh = 0;
  {
    cout_mute mute;
    ProbabilityDist::setSeed(seed);
    globalRNG.reset(ProbabilityDist::getPRNG());
    stateSpace space(D);
    std::vector<std::string> names, types(D, "uni");
    std::vector<double> centers(D, 0.0);
    for (int i = 0; i < D; i++) names.push_back("x" + std::to_string(i));
    space.set_names(names);
    bayes_likelihood like;
    like.register_reference_object(&tgt);
    like.register_evaluate_log(gauss_eval);
    like.basic_setup(&space, types, centers, hw);
    const sampleable_probability_function* prior = like.getObjectPrior().get();
    Eigen::MatrixXd pcov = cov * (2.38 * 2.38 / D);
    // gaussian_prop's covariance constructor self-initialises its `sigmas` member (proposal_distribution.hh:165,
    // SURVEY quirk Q4): it copies an indeterminate valarray.  Construct it in zeroed storage so that the copy is of
    // an empty valarray, which is what the reference's own testGaussian.cc happens to get on a fresh stack.
    void* prop_mem = calloc(1, sizeof(gaussian_prop));
    gaussian_prop& prop = *new (prop_mem) gaussian_prop(pcov);
    parallel_tempering_chains ptc(Nt, Tmax, swap_rate, 100, false, false, -30);
    ptc.initialize(&like, prior, 1);
    std::vector<Eigen::MatrixXd> covs;
    int nextrung = 0;
    per_rung_gaussian rprop(&covs, &nextrung);
    if (fbase > 0) {
      Eigen::MatrixXd B = Eigen::MatrixXd::Zero(D, D);
      for (int i = 0; i < D; i++) B(i, i) = 1.0 / ((fbase * hw[i]) * (fbase * hw[i]));
      for (int r = 0; r < Nt; r++) {
        Eigen::MatrixXd S = (ptc.subchain(r)->invTemp() * P + B).inverse() * (2.38 * 2.38 / D);
        covs.push_back(0.5 * (S + S.transpose()));
      }
      ptc.set_proposal(rprop);
    } else {
      ptc.set_proposal(prop);
    }
    int warm = nsteps / 10 + 1;
    for (int k = 0; k < warm; k++) ptc.step();
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < nsteps; k++) ptc.step();
    auto t1 = std::chrono::steady_clock::now();
    secs = std::chrono::duration<double>(t1 - t0).count();
    nmh = (long)nsteps * Nt;
    globalRNG.reset();
  }
  printf("{\"kind\":\"reference\",\"D\":%d,\"ntemps\":%d,\"nsteps\":%d,\"seconds\":%.6f,\"steps_per_s\":%.6g,"
         "\"threads\":1,\"likelihood_calls\":%ld}\n",
         D, Nt, nsteps, secs, nmh / secs, tgt.ncalls);
  (void)acc0;
  return 0;
}


// ----------------------------------------------------------------------------------------------
// golden-de: the reference's differential_evolution (proposal_distribution.cc:476-801) on the history of a real ladder.
// A parallel_tempering_chains run with scripted proposals builds the histories; then, for a list of parameter sets, clones
// of differential_evolution draw for several rungs.  Recorded per draw: the uniforms the caller's generator delivered
// during the draw (peeked before, their number found afterwards from where the generator stands), the current state, and
// the proposal's outputs (proposed state, validity, log-Hastings ratio, type).  Together with the dumped histories
// (states, log-posteriors, log-likelihoods by RAW index; MAP log-posteriors; temperatures) a restatement fed the same
// uniforms must propose the same states.
// ----------------------------------------------------------------------------------------------
static int golden_de() {
  std::ostringstream js;
  js << "{\"cases\":[";
  struct lcase { int D, Nt, Ninit, nsteps; double Tmax, swap_rate, step_scale; };
  // case 0: short histories (every raw row is eligible); case 1: histories long enough for the ignored early fraction
  // ((size - 100 dim)(1 - ignore_frac) > 10 dim, proposal_distribution.cc:753-756)
  const lcase lcs[] = {{3, 4, 32, 40, 50.0, 0.3, 1.1}, {2, 3, 25, 270, 20.0, 0.25, 0.9}};
  struct vcase { double snooker, g1, bsmall, ignore, alpha, reduce; bool mix; double pmix; };
  const vcase vcs[] = {
      {0.0, 0.1, 1e-4, 0.3, 0.0, 1.0, false, 1.0},     // standard moves, the constructor's defaults
      {1.0, 0.3, 1e-4, 0.3, 0.0, 4.0, false, 1.0},     // snooker moves, reduced gamma
      {0.1, 0.3, 1e-4, 0.0, 2.0, 4.0, false, 1.0},     // the sampler's recipe (ptmcmc.cc:81-91) with unlikely_alpha on
      {0.5, 0.3, 1e-4, 0.3, 1.5, 4.0, true, 1.0},      // mixing over the ladder's rungs
      {0.5, 0.5, 1e-4, 0.2, 0.0, 2.0, true, 3.0},      // ... with mix_temperatures_more
  };
  for (size_t ic = 0; ic < sizeof lcs / sizeof lcs[0]; ic++) {
    const lcase& L = lcs[ic];
    const int D = L.D, Nt = L.Nt;
    splitmix g(0xDE0000 + ic);
    gauss_target tgt;
    tgt.D = D; tgt.ncalls = 0; tgt.P.assign(D * D, 0.0);
    {
      std::vector<double> B(D * D);
      for (auto& b : B) b = g.sym();
      for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) {
          double a = 0;
          for (int k = 0; k < D; k++) a += B[k * D + i] * B[k * D + j];
          tgt.P[i * D + j] = a + (i == j ? 0.5 : 0.0);
        }
      tgt.like0 = -1.25 * D;
    }
    cout_mute mute;
    ProbabilityDist::setSeed(0.31337 + 0.01 * ic);
    globalRNG.reset(ProbabilityDist::getPRNG());
    stateSpace space(D);
    std::vector<std::string> names, types(D, "uni");
    std::vector<double> centers(D, 0.0), scales(D);
    for (int i = 0; i < D; i++) { names.push_back("x" + std::to_string(i)); scales[i] = 6.0 + i; }
    space.set_names(names);
    bayes_likelihood like;
    like.register_reference_object(&tgt);
    like.register_evaluate_log(gauss_eval);
    like.basic_setup(&space, types, centers, scales);
    const sampleable_probability_function* prior = like.getObjectPrior().get();
    parallel_tempering_chains ptc(Nt, L.Tmax, L.swap_rate, 1, false, false, -30);
    ptc.initialize(&like, prior, L.Ninit);
    std::vector<std::vector<std::vector<double>>> deltas(Nt);
    for (int r = 0; r < Nt; r++) {
      const double sc = L.step_scale / std::sqrt(std::max(ptc.subchain(r)->invTemp(), 0.02));
      deltas[r].resize(L.nsteps);
      for (int k = 0; k < L.nsteps; k++) { deltas[r][k].resize(D); for (int d = 0; d < D; d++) deltas[r][k][d] = g.sym() * sc; }
    }
    int nextrung = 0;
    tape_prop tp(&deltas, &nextrung);
    ptc.set_proposal(tp);
    for (int k = 0; k < L.nsteps; k++) ptc.step();
    js << (ic ? "," : "") << "\n{\"D\":" << D << ",\"Nt\":" << Nt << ",\"ladder_MAPlpost\":" << jnum(ptc.getMAPlpost())
       << ",\"ladder_dim\":" << ptc.getDim() << ",\"ladder_size\":" << ptc.size() << ",\"scales\":" << jarr(scales) << ",\n\"rungs\":[";
    for (int r = 0; r < Nt; r++) {
      chain* c = ptc.subchain(r);
      js << (r ? "," : "") << "\n {\"invtemp\":" << jnum(c->invTemp()) << ",\"size\":" << c->size() << ",\"MAPlpost\":" << jnum(c->getMAPlpost())
         << ",\"dim\":" << c->getDim()
         // what an index outside the saved rows reads (MH_chain::getState / getLogPost / getLogLike, chain.cc:1056-1086): the current values
         << ",\"cur_x\":" << jarr(c->getState().get_params_vector()) << ",\"cur_lpost\":" << jnum(c->getLogPost()) << ",\"cur_llike\":" << jnum(c->getLogLike())
         << ",\"x\":[";
      for (int e = 0; e < c->size(); e++) js << (e ? "," : "") << jarr(c->getState(e, true).get_params_vector());
      js << "],\"lpost\":[";
      for (int e = 0; e < c->size(); e++) js << (e ? "," : "") << jnum(c->getLogPost(e, true));
      js << "],\"llike\":[";
      for (int e = 0; e < c->size(); e++) js << (e ? "," : "") << jnum(c->getLogLike(e, true));
      js << "]}";
    }
    js << "],\n\"draws\":[";
    bool firstdraw = true;
    for (size_t iv = 0; iv < sizeof vcs / sizeof vcs[0]; iv++) {
      const vcase& V = vcs[iv];
      for (int r = 0; r < Nt; r++) {
        differential_evolution de(V.snooker, V.g1, V.bsmall, V.ignore, V.alpha);
        de.reduce_gamma(V.reduce);
        de.support_mixing(V.mix);
        de.mix_temperatures_more(V.pmix);
        chain* me = ptc.subchain(r);
        de.set_chain(V.mix ? (chain*)&ptc : me);   // parallel_tempering_chains::set_proposal's rule (chain.cc:1373-1381)
        const int reps = V.mix ? 3 : 4;
        for (int rep = 0; rep < reps; rep++) {
          const int TAPE = 6000;
          std::vector<double> before = peek_tape(me, TAPE);
          state s = me->getState();
          state out = de.draw(s, me);
          std::vector<double> after = peek_tape(me, 2);
          int used = -1;
          for (int p = 0; p + 1 < TAPE; p++) if (before[p] == after[0] && before[p + 1] == after[1]) { used = p; break; }
          if (used < 0) { fprintf(stderr, "golden-de: a draw consumed more than %d uniforms\n", TAPE); return 2; }
          std::vector<double> ks(used);
          for (int p = 0; p < used; p++) ks[p] = before[p] * 4294967296.0 - 0.5;   // MotherOfAll::Next = (k + 0.5) / 2^32: k is exact
          js << (firstdraw ? "" : ",") << "\n {\"variant\":" << iv << ",\"rung\":" << r << ",\"snooker\":" << jnum(V.snooker) << ",\"gamma_one_frac\":" << jnum(V.g1)
             << ",\"b_small\":" << jnum(V.bsmall) << ",\"ignore_frac\":" << jnum(V.ignore) << ",\"unlikely_alpha\":" << jnum(V.alpha)
             << ",\"reduce_gamma\":" << jnum(V.reduce) << ",\"mixing\":" << (V.mix ? 1 : 0) << ",\"mix_factor\":" << jnum(V.pmix)
             << ",\"uniform_k\":" << jarr(ks) << ",\"x\":" << jarr(s.get_params_vector()) << ",\"proposed\":" << jarr(out.get_params_vector())
             << ",\"valid\":" << (out.invalid() ? 0 : 1) << ",\"log_hastings\":" << jnum(de.log_hastings_ratio()) << ",\"type\":" << de.type() << "}";
          firstdraw = false;
        }
      }
    }
    js << "]}";
    globalRNG.reset();
  }
  js << "]}\n";
  std::cout << js.str();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// golden-ess: chain::report_effective_samples (chain.cc:126-643) on fixed series.  A minimal chain subclass serves a
// synthetic history (one saved state per step): AR(1) processes x_t = phi x_(t-1) + e_t per parameter, innovations from
// the helper stream (so the test regenerates the series from the recorded parameters); expected: (ess, useful length)
// for several (width, every, esslimit) calls, the coarse-to-fine search of esslimit >= 0 included.
// ----------------------------------------------------------------------------------------------
struct series_chain : public chain {
  const stateSpace* sp;
  std::vector<std::vector<double>> rows;   // [step][dim]
  series_chain(const stateSpace* sp_, int dim_) : sp(sp_) { dim = dim_; Nsize = 0; Ninit = 0; Nearliest = 0; reporting = false; }
  void finish() { Nsize = (int)rows.size(); }
  int getStep() override { return (int)rows.size(); }
  state getState(int elem = -1, bool raw_indexing = false) override {
    if (elem < 0) elem = (int)rows.size() - 1;
    return state(sp, rows[elem]);
  }
};

static int golden_ess() {
  std::ostringstream js;
  js << "{\"cases\":[";
  struct scase { int dim, n; double phi[3]; unsigned long long seed; };
  const scase scs[] = {{2, 60000, {0.9, 0.5, 0}, 0xE55001ULL}, {3, 150000, {0.98, 0.8, 0.0}, 0xE55002ULL}, {1, 30000, {0.995, 0, 0}, 0xE55003ULL}};
  struct qcase { int width, every; double esslimit; };
  const qcase qcs[] = {{40000, 100, -1.0}, {1000, 1, -1.0}, {10000, 10, -1.0}, {1000, 1, 1000.0}, {1000, 1, 2000.0}, {2000, 2, 5000.0}, {500, 5, 700.0}, {3000, 3, 400.0}};
  for (size_t ic = 0; ic < sizeof scs / sizeof scs[0]; ic++) {
    const scase& S = scs[ic];
    std::vector<std::pair<double, int>> res;
    {
      cout_mute mute;
      ProbabilityDist::setSeed(0.5);
      stateSpace sp(S.dim);
      series_chain c(&sp, S.dim);
      splitmix g(S.seed);
      std::vector<double> x(S.dim, 0.0);
      for (int t = 0; t < S.n; t++) {
        for (int d = 0; d < S.dim; d++) {
          // innovation: sum of four uniforms on (-1, 1) (bell-shaped, exactly reproducible)
          double e = 0;
          for (int q = 0; q < 4; q++) e += g.sym();
          x[d] = S.phi[d] * x[d] + e;
        }
        c.rows.push_back(x);
      }
      c.finish();
      for (size_t iq = 0; iq < sizeof qcs / sizeof qcs[0]; iq++) res.push_back(c.report_effective_samples(-1, qcs[iq].width, qcs[iq].every, qcs[iq].esslimit));
    }
    js << (ic ? "," : "") << "\n{\"dim\":" << S.dim << ",\"n\":" << S.n << ",\"seed\":\"" << S.seed << "\",\"phi\":[";
    for (int d = 0; d < S.dim; d++) js << (d ? "," : "") << jnum(S.phi[d]);
    js << "],\"queries\":[";
    for (size_t iq = 0; iq < sizeof qcs / sizeof qcs[0]; iq++)
      js << (iq ? "," : "") << "\n {\"width\":" << qcs[iq].width << ",\"every\":" << qcs[iq].every << ",\"esslimit\":" << jnum(qcs[iq].esslimit)
         << ",\"ess\":" << jnum(res[iq].first) << ",\"length\":" << res[iq].second << "}";
    js << "]}";
  }
  js << "]}\n";
  std::cout << js.str();
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "golden-basic")) {
    ProbabilityDist::setSeed(0.224);  // chain() constructors draw their seed from the master generator (chain.hh:58-62)
    std::ostringstream os;
    os << "{";
    golden_boundary(os);
    os << ",\n";
    golden_priors(os);
    os << ",\n";
    golden_ladder(os);
    os << "}\n";
    std::cout << os.str();
    return 0;
  }
  if (argc >= 3 && !strcmp(argv[1], "golden-trace")) return golden_trace(atoi(argv[2]));
  if (argc >= 2 && !strcmp(argv[1], "golden-eigen")) return golden_eigen();
  if (argc >= 2 && !strcmp(argv[1], "golden-de")) return golden_de();
  if (argc >= 2 && !strcmp(argv[1], "golden-ess")) return golden_ess();
  if (argc >= 3 && !strcmp(argv[1], "bench")) return bench(argv[2]);
  fprintf(stderr, "usage: %s golden-basic | golden-trace <1..12> | golden-eigen | golden-de | golden-ess | bench <specfile>\n", argv[0]);
  return 2;
}
