// tests/cxx/de_ess_fixture_main.cc -- drives the facade's differential_evolution and ess_estimator with the inputs of the
// reference-generated fixtures (tests/golden/trace13.json.gz, ess.json.gz); tests/test_cxx_facade.py converts the JSON into the
// plain text read here and compares what is printed with the reference's outputs.  No engine call: runs without a GPU.
//
//   de  : stdin = D Nt, then per rung: invtemp size MAPlpost cur_lpost cur_llike / cur_x[D] / size rows of x[D] lpost llike;
//         then ndraws, and per draw: rung snooker g1 bsmall ignore alpha reduce mixing mixfac nuni / the uniforms / x[D]
//         stdout per draw: valid type log_hastings proposed[D]
//   ess : stdin = dim n nq / the series row by row / nq x (width every esslimit); stdout per query: ess length
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "ptmcmc_gpu.hh"

using namespace ptmgpu;

struct tape : public Random {   // the uniforms the reference's generator delivered during the draw, in order
  std::vector<double> u;
  size_t at;
  long overrun;
  tape() : at(0), overrun(0) {}
  double Next() override {
    if (at < u.size()) return u[at++];
    overrun++;
    return 0.5;
  }
};

// a rung of the fixture's ladder as the proposals see a chain: raw-indexed history, the current values outside it
// (MH_chain::getState / getLogPost / getLogLike, chain.cc:1056-1086)
struct fixed_rung : public chain {
  const stateSpace* sp;
  int D;
  double beta, map_lpost, cur_lpost, cur_llike;
  std::vector<double> cur_x;
  std::vector<std::vector<double> > x;
  std::vector<double> lpost, llike;
  std::shared_ptr<tape> rng;
  fixed_rung() : sp(nullptr), D(0), beta(1), map_lpost(0), cur_lpost(0), cur_llike(0), rng(new tape) {}
  void step() override {}
  bool inside(int e) const { return e >= 0 && e < (int)x.size(); }
  state getState(int e = -1, bool raw = false) override { (void)raw; return state(sp, inside(e) ? x[e] : cur_x); }
  double getLogPost(int e = -1, bool raw = false) override { (void)raw; return inside(e) ? lpost[e] : cur_lpost; }
  double getLogLike(int e = -1, bool raw = false) override { (void)raw; return inside(e) ? llike[e] : cur_llike; }
  double invTemp() override { return beta; }
  int getStep() override { return (int)x.size(); }
  int size() override { return (int)x.size(); }
  int getDim() override { return D; }
  std::shared_ptr<Random> getPRNG() override { return rng; }
  double getMAPlpost() override { return map_lpost; }
};
// the ladder itself: history, MAP and size of its coldest rung (chain.hh:278-280, chain.cc:1570-1571)
struct fixed_ladder : public chain {
  std::vector<fixed_rung>* r;
  explicit fixed_ladder(std::vector<fixed_rung>* r) : r(r) {}
  void step() override {}
  state getState(int e = -1, bool raw = false) override { return (*r)[0].getState(e, raw); }
  double getLogPost(int e = -1, bool raw = false) override { return (*r)[0].getLogPost(e, raw); }
  double getLogLike(int e = -1, bool raw = false) override { return (*r)[0].getLogLike(e, raw); }
  int getStep() override { return (*r)[0].getStep(); }
  int size() override { return (*r)[0].size(); }
  int getDim() override { return (*r)[0].D; }
  int multiplicity() override { return (int)r->size(); }
  chain* subchain(int i) override { return &(*r)[i]; }
  double getMAPlpost() override { return (*r)[0].map_lpost; }
};

static int run_de() {
  int D, Nt;
  if (!(std::cin >> D >> Nt)) return 2;
  stateSpace space(D);   // open bounds: every state is valid
  std::vector<fixed_rung> rungs(Nt);
  for (int k = 0; k < Nt; k++) {
    fixed_rung& g = rungs[k];
    int n;
    g.sp = &space; g.D = D;
    std::cin >> g.beta >> n >> g.map_lpost >> g.cur_lpost >> g.cur_llike;
    g.cur_x.resize(D);
    for (int d = 0; d < D; d++) std::cin >> g.cur_x[d];
    g.x.assign(n, std::vector<double>(D)); g.lpost.resize(n); g.llike.resize(n);
    for (int e = 0; e < n; e++) {
      for (int d = 0; d < D; d++) std::cin >> g.x[e][d];
      std::cin >> g.lpost[e] >> g.llike[e];
    }
  }
  fixed_ladder ladder(&rungs);
  int ndraws;
  std::cin >> ndraws;
  for (int k = 0; k < ndraws; k++) {
    int rung, mixing, nuni;
    double snooker, g1, bsmall, ignore, alpha, reduce, mixfac;
    std::cin >> rung >> snooker >> g1 >> bsmall >> ignore >> alpha >> reduce >> mixing >> mixfac >> nuni;
    fixed_rung& me = rungs[rung];
    me.rng->u.resize(nuni);
    for (int i = 0; i < nuni; i++) std::cin >> me.rng->u[i];
    me.rng->at = 0; me.rng->overrun = 0;
    std::vector<double> x(D);
    for (int d = 0; d < D; d++) std::cin >> x[d];
    if (!std::cin) return 2;
    differential_evolution de(snooker, g1, bsmall, ignore, alpha);
    de.reduce_gamma(reduce);
    de.support_mixing(mixing != 0);
    de.mix_temperatures_more(mixfac);
    de.set_chain(mixing ? (chain*)&ladder : (chain*)&me);   // parallel_tempering_chains::set_proposal's rule (chain.cc:1373-1381)
    state s(&space, x);
    state out = de.draw(s, &me);
    printf("%d %d %.17g", out.invalid() ? 0 : 1, de.type(), de.log_hastings_ratio());
    for (int d = 0; d < D; d++) printf(" %.17g", out.get_param(d));
    // uniforms left on the tape: the reference went on to draw the small Gaussian jump it then drops; an overrun would mean
    // this restatement asked for more uniforms than the reference did
    printf(" %ld %ld\n", (long)(me.rng->u.size() - me.rng->at), me.rng->overrun);
  }
  return 0;
}

static int run_ess() {
  int dim, n, nq;
  if (!(std::cin >> dim >> n >> nq)) return 2;
  std::vector<double> series((size_t)dim * n);
  for (size_t i = 0; i < series.size(); i++) std::cin >> series[i];
  ess_estimator est(n, dim, [&](int step, std::vector<double>& row) {
    row.assign(series.begin() + (size_t)step * dim, series.begin() + (size_t)(step + 1) * dim);
    return true;
  });
  for (int q = 0; q < nq; q++) {
    int width, every;
    double limit;
    std::cin >> width >> every >> limit;
    if (!std::cin) return 2;
    const std::pair<double, int> r = est.report(width, every, limit, n, 0);
    printf("%.17g %d\n", r.first, r.second);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "de") return run_de();
  if (argc >= 2 && std::string(argv[1]) == "ess") return run_ess();
  fprintf(stderr, "usage: %s de|ess < fixture-as-text\n", argv[0]);
  return 2;
}
