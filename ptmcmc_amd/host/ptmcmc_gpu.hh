// ptmcmc_gpu.hh -- host-side C++ mirror of ptmcmc's plug-in surface for the chain::step() path, on top of the C ABI
// (include/ptm_engine.h).  Header only, C++11, no HIP types: link against libptm_engine.so.
//
// Same class names, constructor arguments and call sequence as the reference, so that user code written against
//   states.hh / probability_function.hh / proposal_distribution.hh / bayesian.hh / chain.hh / ptmcmc.hh
// reads the same here (namespace ptmgpu; `using namespace ptmgpu;` gives the reference's global names):
//
//   stateSpace space(D); space.set_bound(i, boundary(boundary::wrap, boundary::wrap, 0, 2*M_PI));      states.hh:29-145
//   bayes_likelihood like;  like.basic_setup(&space, types, centers, scales);                        bayesian.hh:360-381
//   like.register_reference_object(obj);  like.register_evaluate_log(fn);                            bayesian.hh:536-552
//   gaussian_prop prop(cov_or_sigmas, oneDfrac);                                       proposal_distribution.hh:145-227
//   parallel_tempering_chains ptc(Ntemps, Tmax, swap_rate, add_every_N);                                chain.cc:1163
//   ptc.initialize(&like, like.getObjectPrior().get(), 1);  ptc.set_proposal(prop);                  chain.cc:1281,1367
//   for (...) ptc.step();                                                                               chain.cc:1393
//   ptc.subchain(i)->getState() / getLogPost() / getLogLike() / invTemp()                             chain.hh:89-141
//
// What runs where: these classes only DESCRIBE the problem (bounds, per-dimension prior, target, ladder, per-rung
// proposal factor) to the engine; every step() is kernels on the MI355X.  A likelihood registered as a function pointer is
// called on the host between the propose and the accept kernel (batched); `gaussian_likelihood` is evaluated on the
// device.  Errors follow the reference: print and exit(1) (chain.cc:967-971, states.cc:87-90).
#ifndef PTMCMC_GPU_HH
#define PTMCMC_GPU_HH

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <valarray>
#include <vector>
#include <sched.h>
#include <sys/stat.h>
#include <sys/types.h>

#include "ptm_engine.h"

namespace ptmgpu {

inline void ptm_check(int rc, const char* what) {
  if (rc != PTM_OK) {
    std::cout << what << ": " << ptm_last_error() << std::endl;
    exit(1);
  }
}

// ---- states.hh --------------------------------------------------------------------------------------------------
class boundary {  // states.hh:29-48
  int lowertype, uppertype;
  double xmin, xmax;

 public:
  static const int open = 0, limit = 1, reflect = 2, wrap = 3;
  boundary(int lowertype = open, int uppertype = open, double min = -INFINITY, double max = INFINITY)
      : lowertype(lowertype), uppertype(uppertype), xmin(min), xmax(max) {}
  void getDomainLimits(double& xmin_, double& xmax_) const { xmin_ = xmin; xmax_ = xmax; }
  bool isWrapped() const { return lowertype == wrap && uppertype == wrap; }
  // boundary::enforce (states.cc:11-58), on the host for states the HOST builds (proposals drawn on the host side); the
  // device enforces every proposal again with the same rules (ptm_kernels.hpp boundary_enforce)
  bool enforce(double& x) const {
    if ((lowertype == wrap) != (uppertype == wrap)) return false;
    if (lowertype == wrap) {
      const double width = xmax - xmin;
      if (width <= 0) return false;
      double xt = std::fmod(x - xmin, width);
      if (xt < 0) xt += width;
      x = xmin + xt;
      return true;
    }
    if (lowertype == reflect && uppertype == reflect) {
      const double halfwidth = xmax - xmin;
      if (halfwidth <= 0) return false;
      const double width = 2 * halfwidth;
      double xt = std::fmod(x - xmin, width);
      if (xt < 0) xt += width;
      if (xt >= halfwidth) xt = halfwidth - xt;   // as the reference folds it (states.cc:41)
      x = xmin + xt;
      return true;
    }
    if (lowertype == reflect && x < xmin) x = xmin + (xmin - x);
    else if (uppertype == reflect && x > xmax) x = xmax - (x - xmax);
    if (lowertype == limit && x < xmin) return false;
    if (uppertype == limit && x > xmax) return false;
    return true;
  }
  int lower() const { return lowertype; }
  int upper() const { return uppertype; }
  std::string show() const {
    std::ostringstream s;
    if (lowertype == wrap) s << "w[" << xmin << "," << xmax << ")w";
    else s << (lowertype == reflect ? "R[" : lowertype == limit ? "[" : "(") << xmin << "," << xmax
           << (uppertype == reflect ? "]R" : uppertype == limit ? "]" : ")");
    return s.str();
  }
};

class stateSpace {  // states.hh:60-145 (names, bounds; symmetries are out of scope)
  int dim;
  std::vector<boundary> bounds;
  std::vector<std::string> names;
  std::map<std::string, int> index;
  bool have_names;

 public:
  stateSpace(int dim = 0) : dim(dim), bounds(dim), have_names(false) {}
  int size() const { return dim; }
  void set_bound(int i, const boundary& b) {
    if (i < dim) bounds[i] = b;
    else { std::cout << "stateSpace::set_bound: Index out of range, " << i << ">=" << dim << "." << std::endl; exit(1); }
  }
  boundary get_bound(int i) const {
    if (i < 0 || i >= dim) { std::cout << "stateSpace::set_bound: Index out of range, " << i << ">=" << dim << "." << std::endl; exit(1); }
    return bounds[i];
  }
  void set_names(const std::vector<std::string>& n) {
    if ((int)n.size() < dim) { std::cout << "stateSpace::set_names: Vector of param names is too short. Quitting." << std::endl; exit(-1); }
    names.assign(n.begin(), n.begin() + dim);
    for (int i = 0; i < dim; i++) index[names[i]] = i;
    have_names = true;
  }
  void set_names(const std::string n[]) { set_names(std::vector<std::string>(n, n + dim)); }
  std::string get_name(int i) const { return have_names && i < dim ? names[i] : "[unnamed]"; }
  bool enforce(std::valarray<double>& params) const {   // states.cc:86-102
    if ((int)params.size() != dim) { std::cout << "stateSpace::enforce:  Dimension error.  Expected " << dim << " params, but given " << params.size() << "." << std::endl; exit(1); }
    for (int i = 0; i < dim; i++)
      if (!bounds[i].enforce(params[i])) return false;
    return true;
  }
  int get_index(const std::string& name) const { return have_names && index.count(name) ? index.at(name) : -1; }
  int requireIndex(const std::string& name) const {
    int i = get_index(name);
    if (i < 0) { std::cout << "stateSpace::checkNames(): Name '" << name << "' not found in state space." << std::endl; exit(1); }
    return i;
  }
  std::string show() const {
    std::ostringstream s;
    s << "StateSpace:(dim=" << dim << ")\n";
    for (int i = 0; i < dim; i++) s << "  " << get_name(i) << " in " << bounds[i].show() << "\n";
    return s.str();
  }
};

class state {  // states.hh:147-234, states.cc:161-253
  const stateSpace* space;
  std::valarray<double> params;
  bool valid;

 public:
  // state(space, n): the zero vector, ENFORCED (states.cc:168-176) -- quirk Q9: if zero violates a `limit` bound the state is
  // born invalid, and every state::add() result with it
  state(const stateSpace* space = nullptr, int n = 0) : space(space), params(0.0, n), valid(space != nullptr) { if (space) enforce(); }
  state(const stateSpace* sp, const std::valarray<double>& a) : space(sp), params(a), valid(sp != nullptr) { enforce(); }
  state(const stateSpace* sp, const std::vector<double>& a) : space(sp), params(a.data(), a.size()), valid(sp != nullptr) { enforce(); }
  // values the engine hands back: enforced on the device already
  static state from_engine(const stateSpace* sp, const double* x, int n) { state s; s.space = sp; s.params = std::valarray<double>(x, n); s.valid = sp != nullptr; return s; }
  void enforce() {   // states.cc:161-166
    if (!space) valid = false;
    if (!valid) return;
    valid = space->enforce(params);
  }
  int size() const { return params.size(); }
  double get_param(int i) const { return params[i]; }
  double get_param(const std::string& name) const { return params[space->requireIndex(name)]; }
  void set_param(int i, double v) { params[i] = v; }
  std::valarray<double> get_params() const { return params; }
  std::vector<double> get_params_vector() const { return std::vector<double>(std::begin(params), std::end(params)); }
  const stateSpace* getSpace() const { return space; }
  bool invalid() const { return !valid; }
  // The state a chain of state::scalar_mult / state::add operations starting from this one ends in (states.cc:183-214): those
  // build on the ENFORCED ORIGIN of the space -- valid iff the origin is (quirk Q9) -- and do not enforce their result.
  state moved_to(const std::vector<double>& x) const {
    state r(space, size());
    for (int i = 0; i < size() && i < (int)x.size(); i++) r.params[i] = x[i];
    return r;
  }
  // the vector-space operations some proposals rely on (states.cc:194-253; extra_enforcement is off in the reference)
  state add(const state& other) const {
    state result(space, size());
    if (other.size() != size()) { std::cout << "state::add: Sizes mismatch. (" << size() << "!=" << other.size() << ")\n"; exit(1); }
    for (int i = 0; i < size(); i++) result.params[i] = params[i] + other.params[i];
    return result;
  }
  state scalar_mult(double x) const {
    state result(space, size());
    for (int i = 0; i < size(); i++) result.params[i] = params[i] * x;
    return result;
  }
  double innerprod(const state& other, bool constrained = false) const {
    if (constrained && !(valid && other.valid)) return NAN;
    if (other.size() != size()) { std::cout << "state::innerprod: sizes mismatch.\n"; exit(1); }
    double result = 0;
    for (int i = 0; i < size(); i++) result += params[i] * other.params[i];
    return result;
  }
  std::vector<int> projection_indices_by_name(const stateSpace* subspace) const {   // states.cc:256-264
    std::vector<int> idx;
    for (int isub = 0; isub < subspace->size(); isub++) idx.push_back(space ? space->get_index(subspace->get_name(isub)) : -1);
    return idx;
  }
  std::string get_string(int prec = -1) const {
    std::ostringstream s;
    if (prec > 0) s.precision(prec);
    for (int i = 0; i < size(); i++) s << (i ? ", " : "") << params[i];
    return s.str();
  }
};

// ---- ProbabilityDist / newran: what the plug-in surface sees of the random numbers ----------------------------------------
// The reference hands every chain its own newran generator and proposals pull uniforms with caller->getPRNG()->Next()
// (chain.hh:44-75).  Here a chain's host-side generator is one more counter-based Philox4x32-10 stream -- keyed like the
// device streams by (seed, walker, rung, step), in a tag domain of its own -- so host-side proposals are as reproducible and
// as independent of the launch geometry as the device's.
class Random {
 public:
  virtual ~Random() {}
  virtual double Next() = 0;   // uniform in (0, 1)
};

namespace detail {
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {   // Salmon et al., SC'11
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
}  // namespace detail

class philox_random : public Random {   // counter: (block, stream, step[31:0], step[55:32] | tag << 24), tag 3 = host proposals
  uint64_t seed, step;
  uint32_t stream, block, buf[4];
  int have;

 public:
  philox_random() : seed(0), step(0), stream(0), block(0), have(0) {}
  void reseat(uint64_t seed_, uint32_t stream_, uint64_t step_) { seed = seed_; stream = stream_; step = step_; block = 0; have = 0; }
  double Next() override {
    if (!have) {
      const uint32_t ctr[4] = {block++, stream, (uint32_t)step, ((uint32_t)(step >> 32) & 0x00FFFFFFu) | (3u << 24)};
      const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
      detail::philox4x32_10(ctr, key, buf);
      have = 4;
    }
    return ((double)buf[4 - have--] + 0.5) * (1.0 / 4294967296.0);   // newran's open-interval map (newran1.cxx:432)
  }
  // a standard normal (Box-Muller on two uniforms; the device has its own table-driven form for its own draws)
  double Normal() {
    const double u1 = Next(), u2 = Next();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

class ProbabilityDist {   // ProbabilityDist.h:37-48: the master seed every chain's generator descends from
 public:
  static double& seed_ref() { static double s = -1; return s; }
  static void setSeed(double seed) { seed_ref() = seed; }
  static double getSeed() { return seed_ref(); }
  // the engine's Philox key for a seed in [0, 1) (< 0: the fixed default)
  static uint64_t engineSeed() { const double s = seed_ref(); return s >= 0 ? (uint64_t)(s * 4294967296.0) : 0x5EED0001ull; }
  // The reference seeds every chain from the master generator in construction order (chain.hh:58-62), so the ladders a
  // program builds one after the other (its loop over clone() / initialize() / run()) differ.  Here: the n-th ladder of the
  // process gets the master key with n folded into the upper word (the first one the master key itself).
  static int& ladders_made() { static int n = 0; return n; }
  static uint64_t nextLadderSeed() { const uint64_t n = (uint64_t)ladders_made()++; return engineSeed() ^ (n * 0x9E3779B97F4A7C15ull << 32); }
};

// ---- command-line flags ("--name=value", bare "--name" = "true") ------------------------------------------------------------
// The reference's programs declare their flags through three small classes (options.hh): every component registers
// Option(name, help, default) objects in one shared Options table (Optioned::addOptions / addOption), main() adds its own and
// calls Options::parse(argc, argv), and the components read their values back by name (optValue / optSet).  Only that calling
// surface -- the class and method names and the flag syntax -- is kept; the table below is this build's own: entries in one
// vector in order of declaration, a name index beside it, and argv compacted in a single pass.
class Options;
class Option {
  friend class Options;
  enum origin { none, preset, command_line };
  std::string key, help, text;
  origin from;

 public:
  Option() : from(none) {}
  Option(const std::string& name, const std::string& info, const std::string& vdefault = "<no default>")
      : key(name), help(info), text(vdefault == "<no default>" ? std::string() : vdefault), from(vdefault == "<no default>" ? none : preset) {}
  // one line that identifies the declaration (two declarations of a flag must agree in it)
  std::string describe() const { return key + "('" + help + "')=" + (from == none ? std::string("<no value>") : text); }
};

class Options {
  std::vector<Option> table;
  std::map<std::string, size_t> where;
  std::string lead;   // "--" or "-"

  const Option* find(const std::string& name) const {
    std::map<std::string, size_t>::const_iterator it = where.find(name);
    return it == where.end() ? nullptr : &table[it->second];
  }
  // is `word` a flag of this table's syntax?  If so its name and, after '=', its value ("true" for a bare flag)
  bool as_flag(const std::string& word, std::string& name, std::string& val) const {
    if (word.size() <= lead.size() || word.compare(0, lead.size(), lead) != 0) return false;
    const size_t eq = word.find('=', lead.size());
    name = word.substr(lead.size(), eq == std::string::npos ? std::string::npos : eq - lead.size());
    val = eq == std::string::npos ? std::string("true") : word.substr(eq + 1);
    return true;
  }
  // records a recognised flag; false (and a complaint, if wanted) for one nobody declared
  bool take(const std::string& name, const std::string& val, bool verbose) {
    std::map<std::string, size_t>::iterator it = where.find(name);
    if (it == where.end()) {
      if (verbose) std::cerr << "Option '" << name << "' not recognized." << std::endl;
      return false;
    }
    table[it->second].text = val;
    table[it->second].from = Option::command_line;
    return true;
  }

 public:
  Options(bool dash_dash = true) : lead(dash_dash ? "--" : "-") {}
  void add(const Option& opt) {
    if (const Option* have = find(opt.key)) {   // the first declaration stands; a different second one is worth a warning
      if (have->describe() != opt.describe())
        std::cout << "Options::add: flag '" << opt.key << "' is declared twice with different help / default; keeping " << have->describe() << std::endl;
      return;
    }
    where[opt.key] = table.size();
    table.push_back(opt);
  }
  bool exists(const std::string& name) const { return find(name) != nullptr; }
  // true if the flag has a value (given, or a default); the value goes to return_value
  bool set(const std::string& name, std::string& return_value) const {
    const Option* o = find(name);
    if (!o) { std::cerr << "Options: Error no option '" << name << "'." << std::endl; return false; }
    if (o->from == Option::none) return false;
    return_value = o->text;
    return true;
  }
  bool set(const std::string& name) const { std::string unused; return set(name, unused); }
  std::string value(const std::string& name) const { std::string v; set(name, v); return v; }
  std::string print_usage() const {
    std::ostringstream os;
    os << "Options:\n";
    for (std::map<std::string, size_t>::const_iterator it = where.begin(); it != where.end(); ++it) {   // by name
      std::string head = "  " + lead + it->first;
      if (head.size() < 26) head.resize(26, ' ');
      os << head << "  " << table[it->second].help << "\n";
    }
    return os.str();
  }
  // Flags this table knows are recorded and taken out of argv (argc shrinks); every other word stays where it is, in order.
  // Returns true if some word looked like a flag but was not declared (the caller may have a second table for those).
  bool parse(int& argc, char* argv[], bool verbose = true) {
    bool unknown = false;
    int kept = argc > 0 ? 1 : 0;
    for (int i = kept; i < argc; i++) {
      std::string name, val;
      if (as_flag(argv[i], name, val)) {
        if (take(name, val, verbose)) continue;
        unknown = true;
      }
      argv[kept++] = argv[i];
    }
    argc = kept;
    return unknown;
  }
  // the same on a vector of words that does not start with the program name
  bool parse(std::vector<std::string>& words, bool verbose = true) {
    bool unknown = false;
    std::vector<std::string> rest;
    for (size_t i = 0; i < words.size(); i++) {
      std::string name, val;
      if (as_flag(words[i], name, val)) {
        if (take(name, val, verbose)) continue;
        unknown = true;
      }
      rest.push_back(words[i]);
    }
    words.swap(rest);
    return unknown;
  }
  std::string report() const {
    std::ostringstream s;
    for (std::map<std::string, size_t>::const_iterator it = where.begin(); it != where.end(); ++it) {
      const Option& o = table[it->second];
      s << " " << it->first << ':' << (o.from == Option::command_line ? o.text : std::string("(not set)")) << '\n';
    }
    return s.str();
  }
};

// mix-in of every component that declares flags: addOptions(table, prefix) binds it to the program's table, addOption declares,
// optValue / optGetValue / optSet read back
class Optioned {
  Options* bound;
  std::string scope;   // prefix of this component's flag names

  Options& options() const {
    if (!bound) { std::cout << "Optioned: addOptions() has not been called for this object; it has no flags to read." << std::endl; exit(1); }
    return *bound;
  }

 protected:
  void copyOptioned(const Optioned& other) { bound = other.bound; scope = other.scope; }
  void addOption(const std::string& name, const std::string& info, const std::string& vdefault = "<no default>") { options().add(Option(scope + name, info, vdefault)); }

 public:
  Optioned() : bound(nullptr) {}
  virtual ~Optioned() {}
  virtual void addOptions(Options& opts, const std::string& prefix_ = "") { bound = &opts; scope = prefix_; }
  bool haveOptions() const { return bound != nullptr; }
  std::unique_ptr<std::istringstream> optValue(const std::string& name) { return std::unique_ptr<std::istringstream>(new std::istringstream(options().value(scope + name))); }
  template <class T> void optGetValue(const std::string& name, T& val) { *optValue(name) >> val; }
  bool optSet(const std::string& name) { return options().set(scope + name); }
  std::string reportOptions() { return options().report(); }
};

// ---- probability_function.hh --------------------------------------------------------------------------------------
class probability_function {  // probability_function.hh:31-44
 protected:
  const stateSpace* space;

 public:
  virtual ~probability_function() {}
  probability_function(const stateSpace* space) : space(space) {}
  virtual double evaluate_log(state& s) { return 0; }
  const stateSpace* get_space() const { return space; }
};

class sampleable_probability_function : public probability_function {  // probability_function.hh:48-83
 protected:
  unsigned int dim;

 public:
  sampleable_probability_function(const stateSpace* space) : probability_function(space), dim(0) {}
  virtual int getDim() const { return dim; }
  virtual void getScales(std::valarray<double>& out) const {}
  virtual void getScales(std::vector<double>& out) const { std::valarray<double> v; getScales(v); out.assign(std::begin(v), std::end(v)); }
  // host-side evaluation and sampling (probability_function.hh:48-83), for proposals that draw from the prior and for the
  // likelihood's best-posterior bookkeeping; the chain's own prior values are computed on the device
  virtual double evaluate(state& s) const { return 1; }
  virtual double evaluate_log(state& s) const { return std::log(evaluate(s)); }   // probability_function.hh:59
  virtual state drawSample(Random& rng) const { std::cout << "sampleable_probability_function::drawSample: not defined for this prior" << std::endl; exit(1); }
  virtual std::string show() const { return "UnspecifiedSampleableProbabilityFunction()"; }
  // engine description: per-dimension (type, center, halfwidth), types as mixed_dist_product::{uniform,...}; true if the prior
  // IS such a product -- it is then evaluated and drawn from on the device.  Any other prior (false, the default: a subclass
  // need only give evaluate / evaluate_log and drawSample, as in the reference) is evaluated on the host, through the engine's
  // prior callback (ptm_set_prior_callback), and the chains' start states are drawn with its drawSample.
  virtual bool describe(std::vector<int>& types, std::vector<double>& centers, std::vector<double>& halfwidths) const { return false; }
};

class mixed_dist_product : public sampleable_probability_function {  // probability_function.hh:141-170, .cc:219-262
 protected:
  std::valarray<int> types;
  std::valarray<double> centers, halfwidths;

 public:
  static const int uniform = 1, gaussian = 2, polar = 3, copolar = 4, log = 5;
  mixed_dist_product(const stateSpace* space, const std::valarray<int>& types, const std::valarray<double>& centers,
                     const std::valarray<double>& halfwidths)
      : sampleable_probability_function(space), types(types), centers(centers), halfwidths(halfwidths) {
    dim = centers.size();
    if (dim != halfwidths.size() || dim != types.size() || (space && (unsigned)space->size() > dim)) {
      std::cout << "mixed_dist_product(constructor): Array sizes mismatch.\n";
      exit(1);
    }
    for (unsigned i = 0; i < dim; i++)
      if (types[i] == log && (centers[i] <= 0 || halfwidths[i] <= 1)) {
        std::cout << "mixed_dist_product(constructor): Need centers>0 and halfwidths>1 for log-type dimension [" << i << "]." << std::endl;
        exit(1);
      }
  }
  void getScales(std::valarray<double>& out) const override { out = halfwidths; }
  // mixed_dist_product::evaluate (probability_function.cc:281-304): the product of the 1-D pdfs of ProbabilityDist.h:76-257;
  // an invalid state has probability 0
  double evaluate(state& s) const override {
    if (s.invalid()) return 0;
    double result = 1;
    for (unsigned i = 0; i < dim; i++) result *= pdf1(i, s.get_param(i));
    return result;
  }
  // ... and its drawSample (probability_function.cc:264-279): one draw per dimension from the 1-D distribution's inverse cdf
  // (uniform / log / polar / copolar) or a normal (gaussian)
  state drawSample(Random& rng) const override {
    std::valarray<double> x(dim);
    for (unsigned i = 0; i < dim; i++) {
      const double c = centers[i], h = halfwidths[i];
      switch (types[i]) {
        case uniform: x[i] = rng.Next() * (2 * h) + (c - h); break;
        case gaussian: { const double u1 = rng.Next(), u2 = rng.Next(); x[i] = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2) * h + c; break; }
        case polar: { double a = c - h, b = c + h; if (a < 0) a = 0; if (b > M_PI) b = M_PI; const double ca = std::cos(a), cb = std::cos(b); x[i] = std::acos(ca - rng.Next() * (ca - cb)); break; }
        case copolar: { double a = c - h, b = c + h; if (a < -M_PI / 2) a = -M_PI / 2; if (b > M_PI / 2) b = M_PI / 2; const double sa = std::sin(a), sb = std::sin(b); x[i] = std::asin(sa + rng.Next() * (sb - sa)); break; }
        case log: { const double l0 = std::log(c / h), l1 = std::log(c * h); x[i] = std::exp(rng.Next() * (l1 - l0) + l0); break; }
        default: x[i] = NAN;
      }
    }
    return state(space, x);
  }
  std::string show() const override {
    std::ostringstream ss;
    ss << "MixedDistProduct(dim=" << dim << ")";
    return ss.str();
  }
  double pdf1(unsigned i, double x) const {
    const double c = centers[i], h = halfwidths[i];
    switch (types[i]) {
      case uniform: return (x < c - h || x > c + h) ? 0 : 1 / (2 * h);
      case gaussian: { const double xn = (x - c) / h; return std::exp(-xn * xn / 2) / std::sqrt(2 * M_PI) / h; }
      case polar: { double a = c - h, b = c + h; if (x < a || x > b) return 0; if (a < 0) a = 0; if (b > M_PI) b = M_PI; return std::sin(x) / (-std::cos(b) + std::cos(a)); }
      case copolar: { double a = c - h, b = c + h; if (x < a || x > b) return 0; if (a < -M_PI / 2) a = -M_PI / 2; if (b > M_PI / 2) b = M_PI / 2; return std::cos(x) / (std::sin(b) - std::sin(a)); }
      case log: { const double a = c / h, b = c * h; if (x < a || x > b) return 0; return 1 / (std::log(b) - std::log(a)) / x; }
    }
    return NAN;
  }
  bool describe(std::vector<int>& t, std::vector<double>& c, std::vector<double>& h) const override {
    t.assign(std::begin(types), std::end(types));
    c.assign(std::begin(centers), std::end(centers));
    h.assign(std::begin(halfwidths), std::end(halfwidths));
    return true;
  }
};

class uniform_dist_product : public mixed_dist_product {  // probability_function.hh:113-125
  static std::valarray<double> mid(const std::valarray<double>& a, const std::valarray<double>& b) { return (a + b) / 2.0; }
  static std::valarray<double> half(const std::valarray<double>& a, const std::valarray<double>& b) { return (b - a) / 2.0; }

 public:
  uniform_dist_product(const stateSpace* space, const std::valarray<double>& min_corner, const std::valarray<double>& max_corner)
      : mixed_dist_product(space, std::valarray<int>(uniform, min_corner.size()), mid(min_corner, max_corner), half(min_corner, max_corner)) {}
};

class gaussian_dist_product : public mixed_dist_product {  // probability_function.hh:92-108
 public:
  gaussian_dist_product(const stateSpace* space, const std::valarray<double>& x0s, const std::valarray<double>& sigmas)
      : mixed_dist_product(space, std::valarray<int>(gaussian, x0s.size()), x0s, sigmas) {}
};

// A prior on a direct product of independent state spaces (probability_function.hh:181-215, .cc:346-452): the product-space
// parameters are matched to the subspaces BY NAME; evaluate() is the product of the subspace priors on the projected states,
// drawSample() the direct product of subspace samples.  As in the reference the j-th parameter handed to subspace k is the
// j-th product-space parameter found in it, in PRODUCT-space order (.cc:387-405,443-447) -- name the subspaces' parameters in the
// order the product space lists them.  If every factor is a per-dimension product the whole is one and lives on the device.
class independent_dist_product : public sampleable_probability_function {
  std::vector<const sampleable_probability_function*> ss_dists;
  std::vector<const stateSpace*> ss;
  std::vector<int> index_ss, index_ss_index;      // product index -> subspace, index within it
  std::vector<std::vector<int> > ss_indices;      // subspace -> product indices found in it, in product order

 public:
  independent_dist_product(const stateSpace* product_space, const std::vector<const sampleable_probability_function*>& subspace_dists)
      : sampleable_probability_function(product_space) {
    dim = product_space->size();
    int dim_count = 0;
    for (size_t i = 0; i < subspace_dists.size(); i++)
      if (subspace_dists[i]->getDim() > 0) {   // empty subspaces are skipped
        ss_dists.push_back(subspace_dists[i]);
        ss.push_back(subspace_dists[i]->get_space());
        dim_count += subspace_dists[i]->getDim();
      }
    if ((int)dim != dim_count) {
      std::cout << "independent_dist_product(constructor): Total dimension of subspaces does not match product space dimension:" << std::endl;
      exit(1);
    }
    ss_indices.resize(ss.size());
    index_ss.resize(dim); index_ss_index.resize(dim);
    for (unsigned i = 0; i < dim; i++) {
      const std::string name = product_space->get_name(i);
      bool found = false;
      for (size_t j = 0; j < ss.size(); j++) {
        const int idx = ss[j]->get_index(name);
        if (idx < 0) continue;
        if (found) { std::cout << "independent_dist_product(constructor): Found name '" << name << "' in multiple spaces." << std::endl; exit(1); }
        found = true;
        index_ss[i] = (int)j; index_ss_index[i] = idx;
        ss_indices[j].push_back((int)i);
      }
      if (!found) { std::cout << "independent_dist_product(constructor): Did not find name '" << name << "' among subspace names." << std::endl; exit(1); }
    }
  }
  independent_dist_product(const stateSpace* product_space, const sampleable_probability_function* d1, const sampleable_probability_function* d2)
      : independent_dist_product(product_space, std::vector<const sampleable_probability_function*>{d1, d2}) {}
  independent_dist_product(const stateSpace* product_space, const sampleable_probability_function* d1, const sampleable_probability_function* d2,
                           const sampleable_probability_function* d3)
      : independent_dist_product(product_space, std::vector<const sampleable_probability_function*>{d1, d2, d3}) {}
  independent_dist_product(const stateSpace* product_space, const sampleable_probability_function* d1, const sampleable_probability_function* d2,
                           const sampleable_probability_function* d3, const sampleable_probability_function* d4)
      : independent_dist_product(product_space, std::vector<const sampleable_probability_function*>{d1, d2, d3, d4}) {}
  state drawSample(Random& rng) const override {
    std::vector<state> sub;
    for (size_t k = 0; k < ss.size(); k++) sub.push_back(ss_dists[k]->drawSample(rng));
    std::valarray<double> v(dim);
    for (unsigned i = 0; i < dim; i++) v[i] = sub[index_ss[i]].get_param(index_ss_index[i]);
    return state(space, v);
  }
  double evaluate(state& s) const override {
    if (s.invalid()) return 0;
    if ((int)dim != s.size()) { std::cout << "independent_dist_product:evaluate: State size mismatch.\n"; exit(1); }
    double result = 1;
    for (size_t k = 0; k < ss.size(); k++) {
      std::valarray<double> v(ss[k]->size());
      for (int j = 0; j < ss[k]->size(); j++) v[j] = s.get_param(ss_indices[k][j]);
      state sub(ss[k], v);
      result *= ss_dists[k]->evaluate(sub);
    }
    return result;
  }
  void getScales(std::valarray<double>& out) const override {
    out.resize(dim);
    int count = 0;
    for (size_t k = 0; k < ss.size(); k++) {
      std::valarray<double> sc;
      ss_dists[k]->getScales(sc);
      for (size_t j = 0; j < sc.size(); j++) out[count++] = sc[j];
    }
  }
  std::string show() const override {
    std::ostringstream o;
    o << "IndependentDistProduct(";
    for (size_t k = 0; k < ss.size(); k++) o << (k ? ", " : "") << ss_dists[k]->show();
    o << ")";
    return o.str();
  }
  bool describe(std::vector<int>& t, std::vector<double>& c, std::vector<double>& h) const override {
    t.assign(dim, 0); c.assign(dim, 0.0); h.assign(dim, 0.0);
    for (size_t k = 0; k < ss.size(); k++) {
      std::vector<int> tk;
      std::vector<double> ck, hk;
      if (!ss_dists[k]->describe(tk, ck, hk)) return false;
      for (size_t j = 0; j < tk.size() && j < ss_indices[k].size(); j++) {
        t[ss_indices[k][j]] = tk[j]; c[ss_indices[k][j]] = ck[j]; h[ss_indices[k][j]] = hk[j];
      }
    }
    return true;
  }
};

// ---- chain.hh: the base interface (what proposals and the driver see of a chain) -------------------------------------------
class proposal_distribution;
class chain {  // chain.hh:34-141
 public:
  virtual ~chain() {}
  virtual void step() = 0;
  virtual state getState(int elem = -1, bool raw_indexing = false) = 0;
  virtual double getLogPost(int elem = -1, bool raw_indexing = false) = 0;
  virtual double getLogLike(int elem = -1, bool raw_indexing = false) = 0;
  virtual double invTemp() { return 1.0; }
  virtual int multiplicity() { return 1; }
  virtual chain* subchain(int index) { return this; }
  virtual int getStep() = 0;
  virtual int size() { return getStep() + 1; }                       // saved history rows (chain.hh:86)
  virtual int getDim() { return 0; }
  virtual int get_id() { return 0; }
  virtual std::shared_ptr<Random> getPRNG() { return std::shared_ptr<Random>(); }   // chain.hh:75
  virtual double getMAPlpost() { return -1e200; }                    // chain.hh:116-117
  virtual state getMAPstate() { return getState(); }
  virtual std::string status() { return ""; }
  virtual std::string report_prop(int style = 0) { return ""; }
};

// ---- proposal_distribution.hh ---------------------------------------------------------------------------------------
// The whole plug-in surface of proposal_distribution.hh:38-88: draw(state&, chain*), log_hastings_ratio(), type(), accept() /
// reject(), clone(), set_chain(), is_ready(), support_mixing().  A proposal that can describe itself as a Gaussian with a fixed
// factor (device_describe) is drawn ON the device, fused into the sweep kernel; every other proposal -- differential
// evolution, sets of mixed members, user proposals with callbacks -- runs through the engine's host-proposal step
// (ptm_set_proposal_callback): draw() is called here, once per moving chain and step, and the device does the rest of
// MH_chain::step (enforce, prior, likelihood, Metropolis test).
class proposal_distribution {
 protected:
  int id;
  double log_hastings;
  int last_type, accept_count, reject_count;
  chain* ch;
  void* user_parent_object;
  void* user_instance_object;
  void* (*new_user_instance_object_function)(void* object, int id);
  static int& idcount() { static int n = 0; return n; }
  void set_instance() {   // proposal_distribution.hh:49-56
    if (user_parent_object && new_user_instance_object_function) user_instance_object = new_user_instance_object_function(user_parent_object, id);
    else user_instance_object = nullptr;
  }

 public:
  virtual ~proposal_distribution() {}
  proposal_distribution(void* user_parent_object = nullptr, void* (*new_user_instance_object_function)(void* object, int id) = nullptr)
      : log_hastings(0), last_type(0), accept_count(0), reject_count(0), ch(nullptr), user_parent_object(user_parent_object),
        user_instance_object(nullptr), new_user_instance_object_function(new_user_instance_object_function) {
    id = idcount()++;
    set_instance();
  }
  virtual double log_hastings_ratio() { return log_hastings; }   // proposal part of the Hastings ratio of the most recent draw
  virtual void set_chain(chain* c) { ch = c; }
  virtual state draw(state& s, chain* caller) { return s; }      // the base class is not useful
  virtual bool is_ready() { return true; }
  virtual proposal_distribution* clone() const { return new proposal_distribution(*this); }
  virtual std::string show() { return "UnspecifiedProposal()"; }
  virtual int type() { return last_type; }
  virtual bool support_mixing() { return false; }
  virtual void accept() { accept_count++; }
  virtual void reject() { reject_count++; }
  virtual void accept(int count) { accept_count = count; }
  virtual void reject(int count) { reject_count = count; }
  virtual void checkpoint(std::string path) {}
  virtual void restart(std::string path) {}
  virtual std::string report(int style = 0) {
    std::ostringstream ss;
    if (style == 0) ss << accept_count * 1.0 / (accept_count + reject_count) << "(" << accept_count << ")";
    return ss.str();
  }
  // ---- device description (this build's addition): a Gaussian with a fixed factor is drawn by the sweep kernel itself
  virtual bool device_describe(int dim, int& kind, std::vector<double>& factor, double& oneDfrac) const { return false; }
  // a set of Gaussian members that are scalar multiples of one factor: cumulative shares, scales, oneDfracs (else false)
  virtual bool device_describe_mixture(int dim, std::vector<double>& cum, std::vector<double>& scales, std::vector<double>& odfs) const { return false; }
  // differential evolution the device can draw itself (ptm_set_proposal_de): no temperature mixing, unlikely_alpha = 0; a set answers
  // for its differential-evolution member
  virtual bool device_describe_de(ptm_de_params& q) const { return false; }
};

namespace detail {
// a standard normal from a chain's generator (the reference: newran's Normal, newran2.cxx:164-217; replaced, not reproduced)
inline double normal_from(Random& rng) {
  const double u1 = rng.Next(), u2 = rng.Next();
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}
// symmetric eigen-decomposition by cyclic Jacobi (the reference uses Eigen::SelfAdjointEigenSolver): eigenvalues ascending in
// `lambda`, eigenvectors in the columns of V (row-major n x n).  Pinned against the reference's gaussian_prop(covar) by
// tests/golden/eigen.json.gz.
inline void jacobi_eigen(std::vector<double> A, int n, std::vector<double>& lambda, std::vector<double>& V) {
  std::vector<double> Q(n * n, 0.0);
  for (int i = 0; i < n; i++) Q[i * n + i] = 1;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        if (std::fabs(A[p * n + q]) < 1e-300) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2 * A[p * n + q]);
        double t = (theta >= 0 ? 1 : -1) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), sn = t * c;
        for (int k = 0; k < n; k++) { double a = A[k * n + p], b = A[k * n + q]; A[k * n + p] = c * a - sn * b; A[k * n + q] = sn * a + c * b; }
        for (int k = 0; k < n; k++) { double a = A[p * n + k], b = A[q * n + k]; A[p * n + k] = c * a - sn * b; A[q * n + k] = sn * a + c * b; }
        for (int k = 0; k < n; k++) { double a = Q[k * n + p], b = Q[k * n + q]; Q[k * n + p] = c * a - sn * b; Q[k * n + q] = sn * a + c * b; }
      }
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]]) std::swap(order[i], order[j]);
  lambda.assign(n, 0.0); V.assign(n * n, 0.0);
  for (int j = 0; j < n; j++) {
    lambda[j] = A[order[j] * n + order[j]];
    for (int i = 0; i < n; i++) V[i * n + j] = Q[i * n + order[j]];
  }
}
}  // namespace detail

class gaussian_prop : public proposal_distribution {  // proposal_distribution.hh:145-227
  bool identity_trans;
  std::vector<double> factor;  // DIAG: sigmas; else dense D x D row-major V*diag(sqrt(lambda))
  int ndim;
  double oneDfrac;

  // factor = V * diag(sqrt(lambda)) so that offset = factor * z as in hh:207-213 (eigenvalues ascending, hh:173-176)
  static void eigen_factor(const std::vector<double>& A, int n, std::vector<double>& F) {
    std::vector<double> lam, V;
    detail::jacobi_eigen(A, n, lam, V);
    F.assign(n * n, 0.0);
    for (int j = 0; j < n; j++) {
      const double sg = std::sqrt(lam[j] > 0 ? lam[j] : 0.0);
      for (int i = 0; i < n; i++) F[i * n + j] = V[i * n + j] * sg;
    }
  }

 public:
  gaussian_prop(const std::valarray<double>& sigmas, double oneDfrac = 0.0, bool scaleWithTemp = false)
      : identity_trans(true), factor(std::begin(sigmas), std::end(sigmas)), ndim(sigmas.size()), oneDfrac(oneDfrac) { check(); }
  gaussian_prop(const std::vector<double>& sigmas, double oneDfrac = 0.0, bool scaleWithTemp = false)
      : identity_trans(true), factor(sigmas), ndim(sigmas.size()), oneDfrac(oneDfrac) { check(); }
  // covariance, row-major ndim x ndim (the reference takes an Eigen::MatrixXd, hh:165)
  gaussian_prop(const std::vector<double>& covar, int ndim, double oneDfrac = 0.0, bool scaleWithTemp = false)
      : identity_trans(false), ndim(ndim), oneDfrac(oneDfrac) {
    if ((int)covar.size() != ndim * ndim) { std::cout << "gaussian_prop(constructor II): covar must be a square matrix!" << std::endl; exit(-1); }
    eigen_factor(covar, ndim, factor);
    check();
  }
  void check() const {
    if (oneDfrac < 0 || oneDfrac > 1) { std::cout << "gaussian_prop(constructor): We require 0<=oneDfrac<=1. " << std::endl; exit(1); }
  }
  gaussian_prop* clone() const override { return new gaussian_prop(*this); }
  std::string show() override {
    std::ostringstream ss;
    ss << "StepBy" << (identity_trans ? "" : "Covar") << "[dim=" << ndim << "](1Dfrac=" << oneDfrac << ")";
    return ss.str();
  }
  // gaussian_prop::draw on the host (hh:194-218) -- used when this proposal is a member of a set that cannot go to the device
  // as a whole: ndim normals first, then the optional one-dimensional move, then the transform (quirk Q4: scaleWithTemp has no
  // effect in the reference)
  state draw(state& s, chain* caller) override {
    Random& rng = *caller->getPRNG();
    std::vector<double> z(ndim);
    for (int i = 0; i < ndim; i++) z[i] = detail::normal_from(rng) * (identity_trans ? factor[i] : 1.0);
    double x = 1;
    if (oneDfrac > 0) x = rng.Next();
    if (oneDfrac > 0 && x < oneDfrac) {
      const int i = (int)(ndim * rng.Next());
      for (int j = 0; j < ndim; j++) if (j != i) z[j] = 0;
      last_type = 1;
    } else last_type = 0;
    std::vector<double> off(ndim, 0.0);
    if (identity_trans) off = z;
    else
      for (int i = 0; i < ndim; i++) { double a = 0; for (int j = 0; j < ndim; j++) a += factor[i * ndim + j] * z[j]; off[i] = a; }
    log_hastings = 0;
    return s.add(state(nullptr, off));
  }
  bool device_describe(int dim, int& kind, std::vector<double>& f, double& odf) const override {
    if (dim != ndim) { std::cout << "gaussian_prop: dimension mismatch with the chain (" << ndim << " vs " << dim << ")" << std::endl; exit(1); }
    kind = identity_trans ? PTM_PROP_DIAG : PTM_PROP_DENSE;
    f = factor;
    odf = oneDfrac;
    return true;
  }
};

// draw_from_dist (proposal_distribution.hh:119-132): an independence proposal from a sampleable distribution (the prior)
class draw_from_dist : public proposal_distribution {
  const sampleable_probability_function& dist;

 public:
  draw_from_dist(const sampleable_probability_function& dist) : dist(dist) {}
  state draw(state& s, chain* caller) override {
    state newstate = dist.drawSample(*caller->getPRNG());
    log_hastings = dist.evaluate_log(s) - dist.evaluate_log(newstate);   // likelier to draw the new state than the old one
    return newstate;
  }
  draw_from_dist* clone() const override { return new draw_from_dist(*this); }
  std::string show() override { return "DrawFrom[" + dist.show() + "]()"; }
};

// user_gaussian_prop (proposal_distribution.hh:236-304, .cc:259-474): a Gaussian step on a named sub-space whose covariance a
// user callback may replace before any draw.  The covariance comes as a vector: ndim variances, or the ndim (ndim+1)/2 entries of
// the upper triangle row by row; it is rescaled to the correlation matrix, diagonalised, and the step is
// diag(sigma) V (sqrt(lambda) o z) (reset_dist, .cc:340-403; negative eigenvalues are set to zero).
class user_gaussian_prop : public proposal_distribution {
 public:
  typedef bool (*check_update_prototype)(const void* parent_object, void* instance_object, const state& s, double invtemp,
                                         const std::vector<double>& randoms, std::vector<double>& covarvec);
  typedef void (*checkpoint_restart_prototype)(const void* parent_object, void* instance_object, const std::string path);
  typedef void (*accept_reject_prototype)(const void* parent_object, void* instance_object);

 private:
  std::vector<double> diagTransform;   // ndim x ndim row-major: diag(sqrt(cov_ii)) * eigenvectors of the correlation matrix
  std::vector<double> sigmas, covar_vec;
  int ndim;
  std::string label;
  check_update_prototype user_check_update;
  bool check_update_registered;
  checkpoint_restart_prototype user_checkpoint, user_restart;
  bool checkpoint_restart_registered;
  accept_reject_prototype user_accept, user_reject;
  bool accept_reject_registered;
  std::vector<int> idx_map;
  int nrand;
  bool have_dist, isverbose;

 protected:
  stateSpace domainSpace;
  void reset_dist(const std::vector<double>& covarvec) {   // .cc:340-403
    covar_vec = covarvec;
    std::vector<double> cov((size_t)ndim * ndim, 0.0);
    const int ULsize = ndim * (ndim + 1) / 2;
    if ((int)covarvec.size() == ndim) {
      for (int i = 0; i < ndim; i++) cov[i * ndim + i] = covarvec[i];
    } else if ((int)covarvec.size() == ULsize) {
      int ic = 0;
      for (int i = 0; i < ndim; i++)
        for (int j = i; j < ndim; j++) { cov[i * ndim + j] = cov[j * ndim + i] = covarvec[ic++]; }
    } else {
      std::cout << "gaussian_prop:reset_dist Covar vector has unexpeced size,ndim=" << ndim << ", UL size=" << ULsize << " but got " << covarvec.size() << " " << std::endl;
      if (have_dist) { std::cout << " Skipping update!" << std::endl; return; }
      std::cout << " Setting to identity matrix!" << std::endl;
      for (int i = 0; i < ndim; i++) cov[i * ndim + i] = 1;
    }
    std::vector<double> corr((size_t)ndim * ndim), lam, V;
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) corr[i * ndim + j] = cov[i * ndim + j] / std::sqrt(cov[i * ndim + i]) / std::sqrt(cov[j * ndim + j]);
    detail::jacobi_eigen(corr, ndim, lam, V);
    diagTransform.assign((size_t)ndim * ndim, 0.0);
    sigmas.assign(ndim, 0.0);
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) diagTransform[i * ndim + j] = std::sqrt(cov[i * ndim + i]) * V[i * ndim + j];
    for (int i = 0; i < ndim; i++) {
      if (lam[i] < 0) {
        std::cout << "user_gaussian_prop[" + label + "]: Warning. Negative covariance eigenvalue[" << i << "]=" << lam[i] << " set to zero." << std::endl;
        lam[i] = 0;
      }
      sigmas[i] = std::sqrt(lam[i]);
    }
    have_dist = true;
  }
  bool check_update(const state& s, chain* caller) {   // .cc:406-441
    if (!check_update_registered) return false;
    std::vector<double> randoms(nrand);
    for (auto& x : randoms) x = caller->getPRNG()->Next();
    std::vector<double> covarvec;
    const double beta = caller->invTemp();
    if (!user_check_update(user_parent_object, user_instance_object, s, beta, randoms, covarvec)) return false;
    reset_dist(covarvec);
    return true;
  }

 public:
  user_gaussian_prop(const stateSpace& sp, const std::vector<double>& covarvec = std::vector<double>(), int nrand = 0, const std::string label = "",
                     void* user_parent_object = nullptr, void* (*new_user_instance_object_function)(void* object, int id) = nullptr)
      : proposal_distribution(user_parent_object, new_user_instance_object_function), ndim(sp.size()), label(label), user_check_update(nullptr),
        check_update_registered(false), user_checkpoint(nullptr), user_restart(nullptr), checkpoint_restart_registered(false),
        user_accept(nullptr), user_reject(nullptr), accept_reject_registered(false), nrand(nrand), have_dist(false), isverbose(false), domainSpace(sp) {
    reset_dist(covarvec);
  }
  user_gaussian_prop(void* user_parent_object, check_update_prototype function, const stateSpace& sp, const std::vector<double>& covarvec = std::vector<double>(),
                     int nrand = 0, const std::string label = "", void* (*new_user_instance_object_function)(void* object, int id) = nullptr)
      : user_gaussian_prop(sp, covarvec, nrand, label, user_parent_object, new_user_instance_object_function) { register_check_update(function); }
  user_gaussian_prop* clone() const override {   // .cc:278-287: a clone asks the parent object for its own instance object
    user_gaussian_prop* c = new user_gaussian_prop(*this);
    c->id = idcount()++;
    c->set_instance();
    return c;
  }
  std::string get_label() const { return label; }
  void verbose(bool set_to = false) { isverbose = set_to; }
  void register_check_update(check_update_prototype function) { user_check_update = function; check_update_registered = true; }
  void register_checkpoint_restart(checkpoint_restart_prototype checkpointfn, checkpoint_restart_prototype restartfn) {
    user_checkpoint = checkpointfn; user_restart = restartfn; checkpoint_restart_registered = true;
  }
  void register_accept_reject(accept_reject_prototype acceptfn, accept_reject_prototype rejectfn) {
    user_accept = acceptfn; user_reject = rejectfn; accept_reject_registered = true;
  }
  state draw(state& s, chain* caller) override {   // .cc:289-338
    if (idx_map.empty()) idx_map = s.projection_indices_by_name(&domainSpace);
    std::vector<double> sparams(domainSpace.size(), 0.0);
    for (int i = 0; i < ndim; i++) if (idx_map[i] >= 0) sparams[i] = s.get_param(idx_map[i]);
    state ss = state(&domainSpace, sparams);
    last_type = check_update(ss, caller);
    Random& rng = *caller->getPRNG();
    std::vector<double> z(ndim), vec(ndim, 0.0);
    for (int i = 0; i < ndim; i++) z[i] = detail::normal_from(rng) * sigmas[i];
    for (int i = 0; i < ndim; i++) { double a = 0; for (int j = 0; j < ndim; j++) a += diagTransform[i * ndim + j] * z[j]; vec[i] = a; }
    state newstate = s.scalar_mult(0);
    for (int i = 0; i < ndim; i++) if (idx_map[i] >= 0) newstate.set_param(idx_map[i], vec[i]);
    log_hastings = 0;
    return s.add(newstate);
  }
  std::string show() override { return "StepByUserCovar[" + label + "]"; }
  void accept() override { if (accept_reject_registered) user_accept(user_parent_object, user_instance_object); accept_count++; }
  void reject() override { if (accept_reject_registered) user_reject(user_parent_object, user_instance_object); reject_count++; }
  void checkpoint(std::string path) override {   // .cc:443-458 (the covariance vector; the user's own data through its callback)
    std::ostringstream ss;
    ss << path << "user_gaussian_proposal=" << id << ".cp/";
    mkdir(ss.str().c_str(), 0777);
    std::ofstream os((ss.str() + "core_data.cp").c_str(), std::ios::binary);
    const size_t n = covar_vec.size();
    os.write((const char*)&n, sizeof n);
    os.write((const char*)covar_vec.data(), (std::streamsize)(n * 8));
    if (checkpoint_restart_registered) user_checkpoint(user_parent_object, user_instance_object, ss.str());
  }
  void restart(std::string path) override {
    std::ostringstream ss;
    ss << path << "user_gaussian_proposal=" << id << ".cp/";
    std::ifstream is((ss.str() + "core_data.cp").c_str(), std::ios::binary);
    size_t n = 0;
    is.read((char*)&n, sizeof n);
    std::vector<double> cv(n);
    is.read((char*)cv.data(), (std::streamsize)(n * 8));
    if (is) reset_dist(cv);
    if (checkpoint_restart_registered) user_restart(user_parent_object, user_instance_object, ss.str());
  }
  // without a callback the step is one fixed Gaussian: the device can draw it (factor embedded in the chain's space by
  // parameter names; rows / columns of dimensions outside the sub-space stay zero)
  bool device_describe_in(const stateSpace* full, int dim, int& kind, std::vector<double>& f, double& odf) const {
    if (check_update_registered || !full) return false;
    std::vector<int> map;
    for (int i = 0; i < ndim; i++) map.push_back(full->get_index(domainSpace.get_name(i)));
    kind = PTM_PROP_DENSE;
    f.assign((size_t)dim * dim, 0.0);
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++)
        if (map[i] >= 0 && map[j] >= 0) f[(size_t)map[i] * dim + map[j]] = diagTransform[i * ndim + j] * sigmas[j];
    odf = 0;
    return true;
  }
};

// proposal_distribution_set (the interface of proposal_distribution.hh:306-349): a weighted mixture of proposals.  One uniform
// of the caller's stream picks a member; its draw, log-Hastings ratio and type (reported as member + 10 x type) become the
// set's.  The weights may adapt to the members' acceptance (adapt_rate: a member that repeats its last outcome -- two accepts
// or two rejects in a row -- loses a little weight, which steers every member towards a mid-range acceptance rate; the pick
// thresholds are rebuilt every 10 x members outcomes) and may depend on the chain's temperature (Tpow, hot_shares: weight
// share + (hot_share - share)(1 - beta^Tpow)).
// On the device: members that are all gaussian_props and scalar multiples of the first one (the sampler's default Gaussian
// recipe is exactly that) -- the rung keeps ONE factor and a table of scales; anything else is drawn here, on the host.
class proposal_distribution_set : public proposal_distribution {
  struct slot {
    proposal_distribution* prop;
    double weight;       // cold share, kept normalised
    double hot_weight;   // share in the hot limit (thermal sets only)
    bool repeated;       // outcome of this member's previous proposal (accepted?)
  };
  std::vector<slot> slots;
  std::vector<double> upper;   // pick thresholds: member i is taken by a uniform below upper[i]; upper.back() == 1
  double adapt_rate, thermal_power;
  int outcomes_seen, rebuild_after, current;
  bool owner;

  // weights -> thresholds.  The weights are renormalised in place; a thermal set moves each threshold towards the hot share by
  // the chain's (1 - beta^Tpow) and then rescales so that the last threshold is exactly 1.
  void rebuild_thresholds() {
    double warm = 0;
    if (thermal_power > 0) {
      if (ch) warm = 1 - std::pow(ch->invTemp(), thermal_power);
      else std::cout << "proposal_distribution_set: temperature-dependent shares need the chain (set_chain) -- using the cold shares." << std::endl;
    }
    double total = 0;
    for (size_t i = 0; i < slots.size(); i++) total += slots[i].weight;
    double edge = 0;
    for (size_t i = 0; i < slots.size(); i++) {
      slot& m = slots[i];
      m.weight /= total;
      edge = edge + m.weight;
      if (thermal_power > 0) edge += (m.hot_weight - m.weight) * warm;
      upper[i] = edge;
    }
    const double top = upper.back();
    for (size_t i = 0; i < upper.size(); i++) upper[i] /= top;
  }
  // bookkeeping shared by accept() and reject()
  void outcome(bool accepted) {
    slot& m = slots[current];
    if (adapt_rate != 0) {
      if (m.repeated == accepted) m.weight *= 1 - adapt_rate * 0.25;
      m.repeated = accepted;
      if (++outcomes_seen >= rebuild_after) rebuild_thresholds();
    }
    if (accepted) m.prop->accept(); else m.prop->reject();
  }
  std::vector<double> weights(bool hot) const {
    std::vector<double> w(slots.size());
    for (size_t i = 0; i < slots.size(); i++) w[i] = hot ? slots[i].hot_weight : slots[i].weight;
    return w;
  }

 public:
  // takes the pointers (take_pointers = true, the reference's default: the set deletes its members); clone() deep-copies
  proposal_distribution_set(const std::vector<proposal_distribution*>& props, const std::vector<double>& shares_, double adapt_rate = 0, double Tpow = 0,
                            std::vector<double> hot_shares_ = std::vector<double>(), bool take_pointers = true)
      : adapt_rate(adapt_rate), thermal_power(Tpow), outcomes_seen(0), rebuild_after(10 * (int)props.size()), current(0), owner(take_pointers) {
    if (props.empty() || props.size() != shares_.size()) { std::cout << "proposal_distribution_set: " << props.size() << " proposals but " << shares_.size() << " shares." << std::endl; exit(1); }
    bool hot_ok = false;
    double hot_total = 0;
    if (Tpow > 0) {
      for (size_t i = 0; i < hot_shares_.size(); i++) hot_total += hot_shares_[i];
      hot_ok = hot_shares_.size() == props.size() && hot_total > 0;
      if (!hot_ok) std::cout << "proposal_distribution_set: Tpow > 0 needs one hot share per member with a positive sum -- using the cold shares." << std::endl;
    }
    for (size_t i = 0; i < props.size(); i++) {
      slot m;
      m.prop = props[i]; m.weight = shares_[i]; m.repeated = true;
      m.hot_weight = Tpow > 0 ? (hot_ok ? hot_shares_[i] / hot_total : shares_[i]) : 0.0;
      slots.push_back(m);
    }
    upper.assign(slots.size(), 0.0);
    rebuild_thresholds();
    last_type = 0;
  }
  ~proposal_distribution_set() { if (owner) for (size_t i = 0; i < slots.size(); i++) delete slots[i].prop; }
  proposal_distribution_set(const proposal_distribution_set&) = delete;
  proposal_distribution_set* clone() const override {
    std::vector<proposal_distribution*> copies;
    for (size_t i = 0; i < slots.size(); i++) copies.push_back(slots[i].prop->clone());
    proposal_distribution_set* c = new proposal_distribution_set(copies, weights(false), adapt_rate, thermal_power, weights(true), true);
    for (size_t i = 0; i < slots.size(); i++) c->slots[i].repeated = slots[i].repeated;
    c->outcomes_seen = outcomes_seen;
    return c;
  }
  void set_chain(chain* c) override {
    ch = c;
    for (size_t i = 0; i < slots.size(); i++) slots[i].prop->set_chain(c);
    rebuild_thresholds();   // (a thermal set now knows its temperature)
  }
  bool support_mixing() override {
    for (size_t i = 0; i < slots.size(); i++) if (slots[i].prop->support_mixing()) return true;
    return false;
  }
  // The pick: a set of one member draws no uniform (so that it IS its member, random stream included); otherwise the first
  // member that is ready and whose threshold lies above the uniform.  If the uniform falls to members that are not ready
  // (differential evolution before its history is long enough) another uniform is drawn -- a hundred times at most.
  state draw(state& s, chain* caller) override {
    Random& rng = *caller->getPRNG();
    for (int attempt = 0; attempt <= 100; attempt++) {
      const double u = slots.size() > 1 ? rng.Next() : 0.0;
      for (size_t i = 0; i < slots.size(); i++) {
        proposal_distribution* p = slots[i].prop;
        if (!(u < upper[i]) || !p->is_ready()) continue;
        state proposed = p->draw(s, caller);
        log_hastings = p->log_hastings_ratio();
        last_type = (int)i + 10 * p->type();
        current = (int)i;
        return proposed;
      }
    }
    std::cout << "proposal_distribution_set::draw: no member was ready to draw in 100 attempts." << std::endl;
    exit(1);
  }
  void accept() override { proposal_distribution::accept(); outcome(true); }
  void reject() override { proposal_distribution::reject(); outcome(false); }
  void checkpoint(std::string path) override { for (size_t i = 0; i < slots.size(); i++) slots[i].prop->checkpoint(path); }
  void restart(std::string path) override { for (size_t i = 0; i < slots.size(); i++) slots[i].prop->restart(path); }
  std::string show() override {
    std::ostringstream ss;
    ss << "ChooseFrom(";
    for (size_t i = 0; i < slots.size(); i++) ss << "  " << (upper[i] - (i ? upper[i - 1] : 0.0)) * 100. << "% : " << slots[i].prop->show() << "\n";
    ss << ")\n";
    return ss.str();
  }
  // style 0: acceptance of the set and of every member; style 1: the current pick probabilities
  std::string report(int style = 0) override {
    std::ostringstream ss;
    if (style == 0) {
      ss << proposal_distribution::report(0) << ":[";
      for (size_t i = 0; i < slots.size(); i++) ss << (i ? "," : "") << slots[i].prop->report(0);
      ss << "]";
    } else if (style == 1) {
      ss << "shares=[";
      for (size_t i = 0; i < slots.size(); i++) {
        ss << (i ? "," : "") << upper[i] - (i ? upper[i - 1] : 0.0);
        const std::string inner = slots[i].prop->report(1);
        if (!inner.empty()) ss << ":" << inner;
      }
      ss << "]";
    }
    return ss.str();
  }
  std::vector<proposal_distribution*> members() const {
    std::vector<proposal_distribution*> v;
    for (size_t i = 0; i < slots.size(); i++) v.push_back(slots[i].prop);
    return v;
  }
  // the member that is a differential evolution the device can draw (at most one, and not the last: a member that is not ready is
  // passed over for the NEXT one, proposal_distribution.cc:111), or -1
  int device_de_member() const {
    int at = -1;
    for (size_t i = 0; i < slots.size(); i++) {
      ptm_de_params q;
      if (!slots[i].prop->device_describe_de(q)) continue;
      if (at >= 0 || i + 1 == slots.size()) return -2;   // two of them, or nothing behind it: the host's business
      at = (int)i;
    }
    return at;
  }
  int first_gaussian_member() const {
    const int de = device_de_member();
    for (size_t i = 0; i < slots.size(); i++) if ((int)i != de) return (int)i;
    return -1;
  }
  bool device_describe_de(ptm_de_params& q) const override {
    const int de = device_de_member();
    return de >= 0 && adapt_rate == 0 && !(thermal_power > 0) && slots[de].prop->device_describe_de(q);
  }
  bool device_describe(int dim, int& kind, std::vector<double>& f, double& odf) const override {
    if (adapt_rate != 0 || thermal_power > 0) return false;   // shares that move are the host's business
    const int g0 = first_gaussian_member();
    if (device_de_member() == -2 || g0 < 0 || !slots[g0].prop->device_describe(dim, kind, f, odf)) return false;
    std::vector<double> cum, sc, od;
    if (!device_describe_mixture(dim, cum, sc, od)) return false;
    odf = 0;   // the members' oneDfracs live in the mixture table
    return true;
  }
  // Gaussian members that are scalar multiples of one factor, and at most one differential evolution (scale -1: ptm_set_proposal_de)
  bool device_describe_mixture(int dim, std::vector<double>& cum, std::vector<double>& scales, std::vector<double>& odfs) const override {
    if (adapt_rate != 0 || thermal_power > 0) return false;
    const int de = device_de_member(), g0 = first_gaussian_member();
    if (de == -2 || g0 < 0) return false;
    int kind0; double odf0; std::vector<double> f0;
    if (!slots[g0].prop->device_describe(dim, kind0, f0, odf0)) return false;
    cum.clear(); scales.clear(); odfs.clear();
    for (size_t i = 0; i < slots.size(); i++) {
      cum.push_back(i + 1 == slots.size() ? 1.0 : upper[i]);
      if ((int)i == de) { scales.push_back(-1.0); odfs.push_back(0.0); continue; }
      int kind; double odf; std::vector<double> f;
      if (!slots[i].prop->device_describe(dim, kind, f, odf) || kind != kind0 || f.size() != f0.size()) return false;
      double sc = 0;
      for (size_t k = 0; k < f.size(); k++) if (f0[k] != 0) { sc = f[k] / f0[k]; break; }
      for (size_t k = 0; k < f.size(); k++)
        if (std::fabs(f[k] - sc * f0[k]) > 1e-12 * (std::fabs(f[k]) + std::fabs(sc * f0[k])) + 1e-300) return false;   // not a multiple
      scales.push_back(sc);
      odfs.push_back(odf);
    }
    return true;
  }
};

// differential_evolution (the interface of proposal_distribution.hh:351-414): DE-MC with sampling from the past, ter Braak &
// Vrugt, Stat. Comput. 18 (2008) 435 -- the jump is a multiple of the difference of two states drawn from the saved history.
//   parallel-direction move (eq. 2):  x* = x + gamma (z1 - z2),  gamma = 1.68 / sqrt(d) / reduce_gamma, or 1 with probability
//                                     gamma_one_frac (unit jumps hop between modes); symmetric, log-Hastings 0, type 0
//   snooker move (eq. 3-4):           z drawn from the history (until z != x), x* = x + gamma ((z1 - z2) . e) e / |e|^2 with
//                                     e = x - z, gamma uniform on (1.2, 2.2) / reduce_gamma; log-Hastings
//                                     (d - 1)/2 (ln |x* - z|^2 - ln |x - z|^2), type 1
// How the reference's implementation behaves (proposal_distribution.cc:476-801), and this one with it -- pinned draw by draw
// against the reference by tests/golden/trace13.json.gz:
//   * rows are taken uniformly from the RAW history [start, size): start > 0 only once (size - 100 d)(1 - ignore_frac) > 10 d,
//     then start = (size - 100 d) ignore_frac; the proposal is ready when the history holds 10 d rows;
//   * unlikely_alpha > 0: a row whose log-posterior lies below (MAP - d) is kept with probability exp(alpha (lpost - MAP + d)),
//     alpha shrinking by 0.9 per refusal.  The MAP and the log-posteriors are those of the proposal's OWN chain (set_chain),
//     also when the row index was drawn for the length of another rung's history;
//   * support_mixing: each history state comes from a rung of the ladder picked with weight exp(min(0, -(L0 - a Lmed) / mix)),
//     a = beta_caller - beta_rung, L0 = ln mean exp(a l) over ten rows (log-likelihoods read from the CALLER's history at
//     indices drawn for the rung's length, non-finite ones redrawn), Lmed the median (6th of 10, sorted) of ten of the rung's
//     own log-likelihoods;
//   * the small Gaussian jump of eq. 2 (b_small) is drawn by the reference and then dropped (its sum is never assigned), so no
//     proposal depends on it: it is not drawn here.
// Needs the chain history -- on this build the ladder keeps a host mirror of what MH_chain::add_state saves when a host-side
// proposal is in use (parallel_tempering_chains::sync_history).
class differential_evolution : public proposal_distribution {
  double p_snooker, p_unit_gamma, jump_noise, skip_early, discount, gamma_divisor, mix_strength;
  bool mixing, bound_to_chain;
  int d;

  int rows_to_start() const { return 10 * d; }
  int rows_before_skipping() const { return 100 * d; }
  void must_be_ready() {
    if (is_ready()) return;
    if (bound_to_chain) std::cout << "differential_evolution: the history holds " << ch->size() << " rows, " << rows_to_start() << " are needed" << std::endl;
    else std::cout << "differential_evolution: no chain was set" << std::endl;
    std::cout << "differential_evolution: drawing before is_ready() -- check is_ready() first." << std::endl;
    exit(1);
  }
  // a raw row index of `source`'s history, by the uniforms of `who`
  int pick_row(chain* who, chain* source) {
    must_be_ready();
    Random& u = *who->getPRNG();
    const int rows = source->size();
    const int spare = rows - rows_before_skipping();
    const int first = spare * (1 - skip_early) > rows_to_start() ? (int)(spare * skip_early) : 0;
    const double floor_lpost = ch->getMAPlpost() - ch->getDim();
    for (double strictness = discount;; strictness *= 0.9) {
      const int row = (int)(first + (rows - first) * u.Next());
      if (!(strictness > 0)) return row;
      const double lpost = ch->getLogPost(row, true);
      if (!(floor_lpost > lpost)) return row;
      if (u.Next() < std::exp(strictness * (lpost - floor_lpost))) return row;
    }
  }
  // weight of rung `rung` as a source of history for a caller at inverse temperature beta_caller
  double rung_weight(chain* who, chain* rung, double beta_caller, bool& same_temperature) {
    const int nprobe = 10;
    double probe[nprobe], lo = 1e100, hi = -1e100;
    for (int got = 0; got < nprobe;) {
      const double l = who->getLogLike(pick_row(who, rung), true);
      if (!std::isfinite(l)) continue;
      if (l > hi) hi = l;
      if (l < lo) lo = l;
      probe[got++] = l;
    }
    const double a = -(rung->invTemp() - beta_caller);
    same_temperature = a == 0;
    const double pivot = a < 0 ? lo : hi;   // keeps every exponent below <= 0
    double acc = 0;
    for (int i = 0; i < nprobe; i++) acc += std::exp((probe[i] - pivot) * a);
    const double log_mean = std::log(acc / nprobe) + pivot * a;
    std::vector<double> own(nprobe);
    for (int i = 0; i < nprobe; i++) own[i] = rung->getLogLike(pick_row(who, rung), true);
    std::sort(own.begin(), own.end());
    double lw = -(log_mean - own[nprobe / 2] * a);
    lw /= mix_strength;
    return std::exp(lw > 0 ? 0.0 : lw);
  }
  // one state of the history, as a parameter vector
  std::vector<double> pick_state(chain* who) {
    must_be_ready();
    const int nrungs = ch->multiplicity();
    if (nrungs == 1 || !mixing) return ch->getState(pick_row(who, ch), true).get_params_vector();
    std::vector<double> reach(nrungs + 1, 0.0);
    const double beta = who->invTemp();
    int own_rung = 0;
    for (int r = 0; r < nrungs; r++) {
      bool same = false;
      reach[r + 1] = reach[r] + rung_weight(who, ch->subchain(r), beta, same);
      if (same) own_rung = r;
    }
    const double mark = who->getPRNG()->Next() * reach[nrungs];
    int from = own_rung;
    for (int r = 0; r < nrungs; r++) if (mark <= reach[r + 1]) { from = r; break; }
    chain* rung = ch->subchain(from);
    return rung->getState(pick_row(who, rung), true).get_params_vector();
  }
  static double dot(const std::vector<double>& a, const std::vector<double>& b) {
    double t = 0;
    for (size_t i = 0; i < a.size(); i++) t += a[i] * b[i];
    return t;
  }
  // v + c w, elementwise (the product rounded before the sum, as state::scalar_mult followed by state::add does it)
  static std::vector<double> plus_scaled(const std::vector<double>& v, const std::vector<double>& w, double c) {
    std::vector<double> r(v.size());
    for (size_t i = 0; i < v.size(); i++) { const double t = w[i] * c; r[i] = v[i] + t; }
    return r;
  }
  state parallel_move(state& s, chain* who) {
    Random& u = *who->getPRNG();
    const double gamma = u.Next() < p_unit_gamma ? 1.0 : 1.68 / std::sqrt((double)d) / gamma_divisor;
    const std::vector<double> z1 = pick_state(who), z2 = pick_state(who);
    std::vector<double> x = plus_scaled(s.get_params_vector(), z1, gamma);
    x = plus_scaled(x, z2, -gamma);
    log_hastings = 0;
    last_type = 0;
    return s.moved_to(x);
  }
  state snooker_move(state& s, chain* who) {
    const double gamma = (1.2 + who->getPRNG()->Next()) / gamma_divisor;
    const std::vector<double> x = s.get_params_vector();
    std::vector<double> z, axis;
    double axis2 = 0;
    for (int tries = 0; axis2 == 0; tries++) {   // the history repeats states: the projection needs a z that is not x itself
      if (tries > 1000) { std::cout << "differential_evolution: a thousand history states in a row equal the current state; giving up." << std::endl; exit(1); }
      z = pick_state(who);
      axis = plus_scaled(x, z, -1.0);
      axis2 = dot(axis, axis);
    }
    const std::vector<double> z1 = pick_state(who), z2 = pick_state(who);
    std::vector<double> diff(x.size());
    for (size_t i = 0; i < diff.size(); i++) { const double a = z1[i] * gamma, b = z2[i] * (-gamma); diff[i] = a + b; }
    const std::vector<double> y = plus_scaled(x, axis, dot(diff, axis) / axis2);
    const std::vector<double> from_z = plus_scaled(y, z, -1.0);
    log_hastings = (std::log(dot(from_z, from_z)) - std::log(axis2)) * (d - 1) / 2.0;
    last_type = 1;
    return s.moved_to(y);
  }

 public:
  differential_evolution(double snooker = 0.0, double gamma_one_frac = 0.1, double b_small = 0.0001, double ignore_frac = 0.3, double unlikely_alpha = 0)
      : p_snooker(snooker), p_unit_gamma(gamma_one_frac), jump_noise(b_small), skip_early(ignore_frac), discount(unlikely_alpha), gamma_divisor(1),
        mix_strength(1), mixing(false), bound_to_chain(false), d(0) {}
  void reduce_gamma(double factor) { gamma_divisor = factor; }
  void mix_temperatures_more(double factor) { mix_strength = factor; }
  void set_chain(chain* c) override { ch = c; bound_to_chain = true; d = c->getDim(); }
  bool is_ready() override { return bound_to_chain && ch->size() >= rows_to_start(); }
  state draw(state& s, chain* caller) override { return p_snooker > caller->getPRNG()->Next() ? snooker_move(s, caller) : parallel_move(s, caller); }
  differential_evolution* clone() const override { return new differential_evolution(*this); }
  std::string show() override {
    std::ostringstream ss;
    ss << "DifferentialEvolution(snooker=" << p_snooker << ", gamma_one_frac=" << p_unit_gamma << ", b_small=" << jump_noise << ", ignore_frac=" << skip_early << ")\n";
    return ss.str();
  }
  bool support_mixing(bool do_it) { mixing = do_it; return do_it; }
  bool support_mixing() override { return mixing; }
  // what the engine draws on the device from its own history (ptm_set_proposal_de): everything but temperature mixing and
  // unlikely_alpha, which stay with this class on the host
  bool device_describe_de(ptm_de_params& q) const override {
    static const bool off = [] { const char* v = getenv("PTM_HOST_DE"); return v && *v && *v != '0'; }();   // (A/B timing: keep the host path)
    if (off || mixing || discount != 0) return false;
    q.snooker = p_snooker; q.gamma_one_frac = p_unit_gamma; q.reduce_gamma = gamma_divisor; q.ignore_frac = skip_early;
    return true;
  }
};

// ---- effective sample size of a saved series ---------------------------------------------------------------------------------
// The estimator behind chain::report_effective_samples (chain.cc:126-643) and the sampler's --chain_ess_stop, restated on a
// plain series so that it can be checked on its own (tests/golden/ess.json.gz holds the reference's answers for fixed AR(1)
// series).  In the manner of Geyer (1992), "Practical Markov chain Monte Carlo", sec. 3:
//   * the last part of the series is cut into windows of `width` steps that end at the newest step, each sampled every
//     `every` steps; `burn` more windows in front of them only feed the lagged samples;
//   * per window and lag L (0, then every, 2 every, ... spaced by factors of 1.1 up to burn windows): the mean m of the
//     samples and their L-lagged partners taken together, and the lagged covariance about m;
//   * for the newest n windows together: rho(L) = sum count (cov_L + (M - m_L)^2) / sum count (cov_0 + (M - m_0)^2), M the mean
//     of the window means; autocorrelation length 1 + 2 sum (L_k - L_(k-1)) rho(L_k), stopped -- and the last term taken back --
//     at the second negative rho in a row (the initially positive sequence);
//   * ess(n) = n width / length (a length below the sampling stride is not believed: 3 strides are assumed instead), the
//     minimum over the features; the answer is the best n: early windows lengthen the correlation more than they add samples.
// report() chooses width / stride like the reference: windows doubled until at most 20 + 2 of them cover the series, or, given
// a limit on the ess worth resolving, a search from coarse to fine strides.
class ess_estimator {
 public:
  // features of the state saved for nominal step `step` (false: none, the sample is skipped)
  typedef std::function<bool(int step, std::vector<double>& features)> reader;

 private:
  int steps, nfeat;
  reader read;
  bool sample(int step, std::vector<double>& v) const { return step >= 0 && step < steps && read(step, v) && (int)v.size() >= nfeat; }

  struct cell { double mean, cov; int count; };

 public:
  ess_estimator(int steps, int nfeat, reader read) : steps(steps), nfeat(nfeat), read(read) {}

  // one pass with fixed windows: ess and the number of windows that gave it
  void windowed(int width, int every, int burn, double& ess_out, int& nwin_out) const {
    ess_out = 0; nwin_out = 0;
    if (width < 2) width = 2;
    if (every < 1) every = 1;
    if (burn < 1) burn = 1;
    const int per_window = width / every, span = per_window * every;   // samples per window, steps they cover
    if (per_window < 1) return;
    int nwin = steps / span - burn;
    if (nwin < 1) return;
    const int origin = steps - nwin * span;
    std::vector<int> lags(1, 0);
    {
      double grow = 1;
      for (int k = 1; k < burn * per_window;) {
        lags.push_back(every * k);
        const int was = k;
        while (k == was) { grow *= 1.1; k = (int)grow; }
      }
    }
    const int nlag = (int)lags.size();
    // table[f][w][l]
    std::vector<std::vector<std::vector<cell> > > table(nfeat, std::vector<std::vector<cell> >(nwin, std::vector<cell>(nlag)));
    std::vector<double> now, then;
    std::vector<std::vector<double> > base(per_window);
    std::vector<char> have(per_window);
    for (int w = 0; w < nwin; w++) {
      const int w0 = origin + w * span;
      for (int i = 0; i < per_window; i++) have[i] = sample(w0 + i * every, base[i]) ? 1 : 0;
      for (int l = 0; l < nlag; l++) {
        std::vector<double> s1(nfeat, 0.0), s2(nfeat, 0.0);
        int n = 0;
        for (int i = 0; i < per_window; i++) {
          if (!have[i]) continue;
          if (l == 0) {
            for (int f = 0; f < nfeat; f++) { s1[f] += base[i][f]; s2[f] += base[i][f] * base[i][f]; }
          } else {
            if (!sample(w0 + i * every - lags[l], then)) continue;
            for (int f = 0; f < nfeat; f++) { s1[f] += then[f] + base[i][f]; s2[f] += then[f] * base[i][f]; }
          }
          n++;
        }
        for (int f = 0; f < nfeat; f++) {
          cell& c = table[f][w][l];
          c.count = n;
          c.mean = l == 0 ? s1[f] / n : s1[f] / n / 2;
          c.cov = s2[f] / n - c.mean * c.mean;
        }
      }
    }
    for (int n = 1; n <= nwin; n++) {
      double worst = 1e100;
      for (int f = 0; f < nfeat; f++) {
        double msum = 0;
        for (int w = nwin - n; w < nwin; w++) msum += table[f][w][0].mean;
        const double M = msum / n;
        double length = 1.0, last_term = 0, previous = 1;
        int last_lag = 0;
        for (int l = 1; l < nlag; l++) {
          double top = 0, bottom = 0;
          for (int w = nwin - n; w < nwin; w++) {
            const cell &c = table[f][w][l], &c0 = table[f][w][0];
            const double dm = M - c.mean, dm0 = M - c0.mean;
            const double cv = c.cov + dm * dm, var = c0.cov + dm0 * dm0;
            top += cv * c.count;
            bottom += var * c.count;
          }
          const double rho = top / bottom;
          if (previous < 0 && rho < 0) { length -= last_term; break; }
          previous = rho;
          last_term = 2.0 * (lags[l] - last_lag) * rho;
          length += last_term;
          last_lag = lags[l];
        }
        double e = n * width / length;
        if (length < every) e = n * width / 3.0 / every;
        if (e < worst) worst = e;
      }
      if (worst > ess_out) { ess_out = worst; nwin_out = n; }
    }
  }

  // (ess, useful length of the series in steps).  width: first guess of the window; every: sampling stride (< 0: the stride the
  // series was saved with, from its rows and its initial rows); esslimit < 0: no limit
  std::pair<double, int> report(int width, int every, double esslimit, int rows = 0, int initial_rows = 0) const {
    const int min_burn = 2, min_per_window = 1000, max_windows = 20;
    while (width < steps * 0.05) width *= 2;
    if (every < 0) every = rows > initial_rows ? (int)(0.5 + ((double)steps - initial_rows) / (rows - initial_rows)) : 1;
    if (every < 1) every = 1;
    double ess = 0, best_width = 0;
    int nwin = 0;
    if (esslimit < 0) {
      if (width < 0) width = every * min_per_window;
      while (width * (max_windows + min_burn) < steps) width *= 2;
      windowed(width, every, min_burn, ess, nwin);
      best_width = width;
    } else {
      // An ess above the limit need not be resolved, so a long series is first looked at with a coarse stride (cheap) and the
      // stride refined only while the limit, not the series' length, decides the window: a window should hold ~1000 samples,
      // the newest <= 20 windows are used, and limit x 3 strides is the longest stretch whose ess could still matter.
      const double length = steps, reach = esslimit * 3.0;
      for (bool last_round = false; !last_round; every *= 2) {
        int windows = (int)(length / (min_per_window * every));
        if (windows > max_windows) windows = max_windows;
        if (windows < 1) break;
        width = (int)(length / windows);
        if (width * (windows - 1) > reach * every) {
          windows = (int)(reach / min_per_window + 1);
          if (windows > max_windows) windows = max_windows;
          if (windows > 1) width = (int)((reach * every) / (windows - 1));
          else { windows = 1; width = min_per_window * every; }
        } else last_round = true;
        if ((length - length / (max_windows + min_burn)) * 0.5 < windows * width) {
          double e; int n;
          windowed(width, every, (int)(length / width - windows), e, n);
          if (e > ess) { ess = e; nwin = n; best_width = width; }
        }
      }
    }
    return std::make_pair(ess, (int)best_width * nwin);
  }
};

// A small persistent worker pool for the likelihood batches: starting and joining threads for every batch costs more than
// a cheap plug-in's whole batch.  run(n, chunk, f) calls f(k0, k1) over [0, n) in chunks, on the workers and the caller.
class eval_pool {
  // Workers sleep on a condition variable between batches.  PTM_EVAL_SPIN_US=<us> lets them SPIN that long for the next batch first (a
  // running sampler hands one over every ~50-100 us; a sleeping thread takes ~50 us to wake, the whole batch of a 128-chain ladder) --
  // an OpenMP runtime's active wait policy.  Not the default: on a shared host it is a gamble (the LISA example at 128 temperatures,
  // spinning / one thread, on two boxes of the pool: 76 / 102 and 117 / 84 us per step), and spinning workers eat a container's CPU quota.
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  std::function<void(int, int)> job;
  std::atomic<int> next{0}, gen{0}, busy{0}, sleepers{0};
  std::atomic<bool> stop{false};
  int n = 0, chunk = 1;
  long long spin_ns = 0;
  static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  static void relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void drain() {
    for (;;) {
      const int k0 = next.fetch_add(chunk);
      if (k0 >= n) return;
      job(k0, k0 + chunk < n ? k0 + chunk : n);
    }
  }
  void loop(int seen) {   // `seen`: the generation current when this worker was created (resize() holds the mutex)
    for (;;) {
      bool got = false;
      for (const long long t0 = now_ns(); spin_ns > 0 && now_ns() - t0 < spin_ns;) {
        if (stop.load(std::memory_order_acquire)) return;
        if (gen.load(std::memory_order_acquire) != seen) { got = true; break; }
        for (int k = 0; k < 32; k++) relax();
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(m);
        sleepers.fetch_add(1);
        cv_go.wait(lk, [&] { return stop.load() || gen.load() != seen; });
        sleepers.fetch_sub(1);
        if (stop.load()) return;
      }
      seen = gen.load(std::memory_order_acquire);
      drain();
      if (busy.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(m); cv_done.notify_one(); }
    }
  }

 public:
  eval_pool() { if (const char* v = getenv("PTM_EVAL_SPIN_US")) spin_ns = (long long)(atof(v) * 1e3); }
  ~eval_pool() {
    { std::lock_guard<std::mutex> lk(m); stop.store(true); }
    cv_go.notify_all();
    for (auto& t : workers) t.join();
  }
  int size() const { return (int)workers.size(); }
  void resize(int nworkers) {
    std::lock_guard<std::mutex> lk(m);   // a new worker starts from the current generation: it waits for the NEXT run
    while ((int)workers.size() < nworkers) {
      const int g = gen.load();
      workers.emplace_back([this, g] { loop(g); });
    }
  }
  void run(int n_, int chunk_, const std::function<void(int, int)>& f) {
    // (every worker has left the batch before: run() returns only when `busy` is back to zero)
    job = f; n = n_; chunk = chunk_ < 1 ? 1 : chunk_;
    next.store(0); busy.store((int)workers.size());
    gen.fetch_add(1);   // (sequentially consistent, like the workers' `sleepers` count: one of the two sides always sees the other's write)
    if (sleepers.load() > 0) { std::lock_guard<std::mutex> lk(m); cv_go.notify_all(); }   // (a worker on its way to sleep checks `gen` under the mutex)
    drain();
    for (const long long t0 = now_ns(); busy.load(std::memory_order_acquire) != 0;) {
      if (now_ns() - t0 > 2000000) {   // a worker that sleeps or was descheduled: wait for it without burning this thread
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return busy.load() == 0; });
        break;
      }
      relax();
    }
  }
};

// ---- bayesian.hh: the likelihood plug-in -----------------------------------------------------------------------------
class bayes_likelihood : public probability_function, public Optioned {  // bayesian.hh:307-581 (minimal interface)
 protected:
  stateSpace nativeSpace;
  std::shared_ptr<const sampleable_probability_function> nativePrior;
  double (*user_evaluate_log)(void* object, const state& s);
  void* user_object;
  bool evaluate_log_registered;
  double best_post;
  std::atomic<double> best_seen;   // best_post as the evaluating threads may read it without the mutex (it only ever grows)
  state best;
  std::mutex best_mutex;
  std::vector<double> likelyScales;
  bool have_scales;
  std::vector<proposal_distribution*> proposals;   // likelihood-associated proposals (bayesian.hh:784-790), owned
  std::vector<double> prop_shares;

 public:
  bool check_posterior;
  bayes_likelihood() : probability_function(nullptr), user_evaluate_log(nullptr), user_object(nullptr),
                       evaluate_log_registered(false), best_post(-INFINITY), best_seen(-INFINITY), have_scales(false), check_posterior(true) {}
  ~bayes_likelihood() { for (auto p : proposals) delete p; }
  bayes_likelihood(const bayes_likelihood&) = delete;
  void addOptions(Options& opt, const std::string& prefix = "") override { Optioned::addOptions(opt, prefix); }
  virtual void setup() {}   // the minimal interface has nothing to set up after basic_setup (bayesian.hh:394-405 is for data + signal)
  virtual void reset() {    // bayesian.hh:407-412
    best_post = -INFINITY;
    best_seen.store(-INFINITY);
    if (space) best = state(space, space->size()).scalar_mult(0);
  }
  virtual state bestState() { return best; }
  virtual double bestPost() { return best_post; }
  virtual void getScales(std::vector<double>& scales) {   // bayesian.hh:384-391
    if (have_scales) scales = likelyScales;
    else getObjectPrior()->getScales(scales);
  }
  void addProposal(const proposal_distribution* proposal, double share = 1) { proposals.push_back(proposal->clone()); prop_shares.push_back(share); }
  std::vector<proposal_distribution*> get_proposals() const { return proposals; }
  std::vector<double> get_prop_shares() const { return prop_shares; }
  void basic_setup(const stateSpace* sp, sampleable_probability_function* prior) {  // bayesian.hh:345-358
    nativeSpace = *sp;
    nativePrior.reset(prior);
    space = &nativeSpace;
    best = state(space, space->size());
    reset();
  }
  void basic_setup(const stateSpace* sp, const std::vector<std::string>& types, const std::vector<double>& centers,
                   const std::vector<double>& priorScales) {  // bayesian.hh:360-381
    std::valarray<int> t(types.size());
    for (size_t i = 0; i < types.size(); i++) {
      const std::string& s = types[i];
      if (s == "uni" || s == "uniform") t[i] = mixed_dist_product::uniform;
      else if (s == "gauss" || s == "gaussian") t[i] = mixed_dist_product::gaussian;
      else if (s == "pol" || s == "polar") t[i] = mixed_dist_product::polar;
      else if (s == "cpol" || s == "copol") t[i] = mixed_dist_product::copolar;
      else if (s == "log") t[i] = mixed_dist_product::log;
      else { std::cout << "bayes_likelihood::basic_setup: unknown prior type '" << s << "'" << std::endl; exit(1); }
    }
    nativeSpace = *sp;
    basic_setup(sp, new mixed_dist_product(&nativeSpace, t, std::valarray<double>(centers.data(), centers.size()),
                                           std::valarray<double>(priorScales.data(), priorScales.size())));
  }
  void register_reference_object(void* object) { user_object = object; }
  void register_evaluate_log(double (*function)(void* object, const state& s)) {
    user_evaluate_log = function;
    evaluate_log_registered = true;
  }
  std::shared_ptr<const sampleable_probability_function> getObjectPrior() const {
    if (!nativePrior) { std::cout << "bayes_component::getObjectPrior: No prior is defined for this object!" << std::endl; exit(1); }
    return nativePrior;
  }
  const stateSpace* getObjectStateSpace() const { return &nativeSpace; }
  // host evaluation of the plug-in (bayesian.hh:553-581), with the reference's best-posterior bookkeeping (quirk Q8: the
  // prior is evaluated again here, on the host, and the best log-posterior seen so far and its state are kept under a lock;
  // a posterior that is not finite turns the likelihood into -inf -- the reference also prints a warning for every such
  // state, this build only for NaN / +inf: a -inf likelihood outside the support is ordinary)
  double evaluate_log(state& s) override {
    if (!evaluate_log_registered) { std::cout << "bayes_component::panic!\nNo evaluate_log function is registered" << std::endl; exit(1); }
    double result = (*user_evaluate_log)(user_object, s);
    if (check_posterior) {
      const double lprior = nativePrior ? nativePrior->evaluate_log(s) : 0.0;
      const double post = result + lprior;
      // (the mutex only for a state that may be a new best: sixteen threads taking it for every evaluation evaluate one after the other)
      if (!(post <= best_seen.load(std::memory_order_relaxed))) {
        std::lock_guard<std::mutex> lk(best_mutex);
        if (!(post <= best_post)) { best_post = post; best = s; best_seen.store(post, std::memory_order_relaxed); }   // (a NaN sticks, as in the unguarded form)
      }
      if (!std::isfinite(post)) {
        if (!(post < 0)) std::cout << "Logpost is NAN!\n  params=" << s.get_string() << "\n  like=" << result << "  post=" << post << std::endl;
        result = -INFINITY;
      }
    }
    return result;
  }
  // device-resident targets override this and return true after describing themselves to the engine
  virtual bool describe_device_target(ptm_engine* e) { return false; }
  // C-ABI trampoline: the engine hands over the gated proposals of one sweep
  // The batch is spread over host threads -- the reference evaluates the rungs of one step under `omp parallel for`
  // (chain.cc:1544-1557) and so requires a thread-safe evaluate_log already; set_eval_threads(1) serialises.
  void set_eval_threads(int n) { eval_threads = n; }
  static void batch_trampoline(void* self, const double* X, int n, int dim, double* out) {
    bayes_likelihood* l = (bayes_likelihood*)self;
    auto work = [l, X, dim, out](int k0, int k1) {
      for (int k = k0; k < k1; k++) {
        state s = state::from_engine(l->getObjectStateSpace(), X + (size_t)k * dim, dim);   // (enforced on the device already)
        out[k] = l->evaluate_log(s);
      }
    };
    // (half of the usable processors: this thread and the HIP runtime's own need some, and a container's CPU quota throttles EVERY
    //  thread of the process once it is used up)
    int nt = l->eval_threads > 0 ? l->eval_threads : (usable_cpus() + 1) / 2;
    if (nt > n / 16) nt = n / 16;    // at least 16 states per thread
    // waking sleeping workers costs ~50 us: a batch that the measured cost per evaluation prices below ~4 such units stays on this
    // thread -- 30 us if the workers spin for their batches (PTM_EVAL_SPIN_US) (eval_threads == 0 only; an explicit thread count is obeyed)
    static const double pool_from_ns = [] {
      const char* v = getenv("PTM_EVAL_POOL_NS");
      if (v && *v) return atof(v);
      const char* sp = getenv("PTM_EVAL_SPIN_US");
      return (sp && atof(sp) > 0) ? 30e3 : 200e3;
    }();
    if (l->eval_threads <= 0 && l->eval_ns >= 0 && (l->eval_ns * n < pool_from_ns || l->pool_off)) nt = 1;
    if (nt <= 1) {
      const auto t0 = std::chrono::steady_clock::now();
      work(0, n);
      const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
      if (n > 0) l->eval_ns = l->eval_ns < 0 ? ns / n : 0.8 * l->eval_ns + 0.2 * ns / n;
      return;
    }
    if (!l->pool) l->pool.reset(new eval_pool);
    l->pool->resize(nt - 1);   // the calling thread works too
    const auto t0 = std::chrono::steady_clock::now();
    l->pool->run(n, (n + 4 * nt - 1) / (4 * nt), work);
    // does the pool pay on this machine?  (a container may show more hardware threads than it lets the process use; a likelihood may
    // serialise itself.)  Three batches in a row that cost more per evaluation than 0.7 of this thread alone: no pool from then on.
    if (l->eval_threads <= 0 && l->eval_ns > 0 && n > 0) {
      const double per = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / n;
      if (per > 0.7 * l->eval_ns) { if (++l->pool_bad >= 3) l->pool_off = true; }
      else l->pool_bad = 0;
    }
  }
  // the processors this process may really use: hardware threads, cut by its affinity mask and by a cgroup CPU quota
  // (PTM_EVAL_THREADS overrides)
  static int usable_cpus() {
    static const int n = [] {
      int c = (int)std::thread::hardware_concurrency();
      if (c < 1) c = 1;
#ifdef __linux__
      cpu_set_t set;
      if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0 && k < c) c = k; }
      {
        std::ifstream f2("/sys/fs/cgroup/cpu.max");   // cgroup v2: "<quota|max> <period>"
        std::string q;
        long long per = 0;
        if (f2 >> q >> per && q != "max" && per > 0) { const long long k = (atoll(q.c_str()) + per - 1) / per; if (k >= 1 && k < c) c = (int)k; }
        std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // cgroup v1
        long long q1 = 0, p1 = 0;
        if (fq >> q1 && fp >> p1 && q1 > 0 && p1 > 0) { const long long k = (q1 + p1 - 1) / p1; if (k >= 1 && k < c) c = (int)k; }
      }
#endif
      if (const char* v = getenv("PTM_EVAL_THREADS")) { const int k = atoi(v); if (k >= 1) c = k; }
      return c;
    }();
    return n;
  }

 private:
  int eval_threads = 0;   // 0: all hardware threads
  double eval_ns = 0;     // running estimate of one evaluation's cost (the first batch runs serially and measures it)
  int pool_bad = 0;       // batches in a row that the pool made no faster
  bool pool_off = false;  // ... three of them: this likelihood stays on the calling thread
  std::unique_ptr<eval_pool> pool;

 public:
};

// correlated Gaussian target evaluated ON the device: like0 - 1/2 (x-mean)^T P (x-mean) (cython/exampleGaussian.py:46-109)
class gaussian_likelihood : public bayes_likelihood {
  std::vector<double> mean, precision;
  double like0;

 public:
  gaussian_likelihood(const std::vector<double>& precision_row_major, double like0, const std::vector<double>& mean = std::vector<double>())
      : mean(mean), precision(precision_row_major), like0(like0) {}
  bool describe_device_target(ptm_engine* e) override {
    ptm_check(ptm_set_target_gaussian(e, mean.empty() ? nullptr : mean.data(), precision.data(), like0), "gaussian_likelihood");
    return true;
  }
  double evaluate_log(state& s) override {
    const int D = s.size();
    double q = 0;
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++)
        q += (s.get_param(i) - (mean.empty() ? 0 : mean[i])) * precision[i * D + j] * (s.get_param(j) - (mean.empty() ? 0 : mean[j]));
    return like0 - 0.5 * q;
  }
};

class parallel_tempering_chains : public chain {  // chain.hh:214-330, chain.cc:1163-1571
  const int Ntemps, add_every_N;
  const double Tmax, swap_rate, dpriormin;
  ptm_engine* eng;
  const stateSpace* sp;
  int dim, nstep, hist_rows, hist_rungs = 0;   // hist_rungs: the coldest rungs whose saved states are kept (0: all)
  int Woff = 0, eng_device = -1;   // first GLOBAL replica of this engine, its device (set_replica_range)
  int W;   // independent replicas of the ladder run side by side (the reference runs its Nchain repeats one after the
           // other, ptmcmc.cc main loop): chain (rung i, replica w) sits at index i*W + w of every engine array
  std::vector<double> temps, X, llike, lpost;
  bool fresh, hist_fresh = false;
  size_t hist_have = 0;       // history chains [0, hist_have) are in hx / hl / hp / hmeta / hb (when hist_fresh)
  int hist_read_rungs = 0;    // > 0: a read-back fetches this many rungs (limit_history_reads), 0: all that keep history
  // replica-exchange diagnostics of chain.cc:1346-1356,1448-1451,1495-1498 (directions / instances / ups / downs), kept
  // on the host by replaying each step's candidate log (replica 0); on after track_exchanges(true)
  bool tracking = false;
  std::vector<int> directions, instances;
  std::vector<long> ups, downs;
  std::vector<int32_t> log_pairs, log_acc;
  void fetch_swaps() {
    const int ms = ptm_max_swaps_per_step(eng);
    log_pairs.resize((size_t)W * ms); log_acc.resize((size_t)W * ms);
    ptm_check(ptm_get_last_swaps(eng, log_pairs.data(), log_acc.data()), "replay_step");
  }
  void replay_swaps() {   // the loop of chain.cc:1436-1537, bookkeeping part, in pick order
    const int ms = ptm_max_swaps_per_step(eng);
    for (int j = 0; j < ms; j++) {
      const int i = log_pairs[j];
      if (i < 0) continue;
      if (i > 0) {                                   // chain.cc:1448-1451
        if (directions[i] > 0) ups[i]++;
        if (directions[i] < 0) downs[i]++;
      }
      if (log_acc[j]) {                              // chain.cc:1495-1498
        std::swap(directions[i], directions[i + 1]);
        std::swap(instances[i], instances[i + 1]);
        if (i == 0) directions[i] = 1;
        if (i + 1 == Ntemps - 1) directions[i + 1] = -1;
      }
    }
  }
  std::vector<double> hx, hl, hp;       // host copy of the history ring (dumpChain)
  std::vector<int32_t> hmeta;
  std::vector<int64_t> hnhist;
  std::vector<double> hb;
  std::vector<proposal_distribution*> props;

  // ---- host-side proposals (anything set_proposal cannot describe to the device as a fixed Gaussian) --------------------
  // The engine calls back once per sweep with the current states of every chain that moves (ptm_set_proposal_callback);
  // each chain's own clone of the proposal draws for it, with the chain's view as `caller`.  A host MIRROR of what
  // MH_chain::add_state saves (chain.cc:935-946) is kept for the proposals that read the chain history (differential
  // evolution): all Ninit initial draws, then every saved row, pulled from the device's short history ring after each step.
  bool host_mode = false, want_host = false;
  bool want_de = false, de_built = false;   // differential evolution drawn on the device from the device's history
  int ring_rows = 0, ring_rungs = 0;   // the device history ring as the engine was created with it
  uint64_t eng_seed = 0;
  struct mirror_t {   // one chain's saved history, raw indexing as MH_chain::states / lposts / llikes (chain.hh:155-163)
    std::vector<double> x, llike, lprior, beta;
    std::vector<int32_t> meta;   // {Naccept, Ntries, type} per row
    size_t rows() const { return llike.size(); }
  };
  std::vector<mirror_t> mirror;        // [i*W + w]
  std::vector<int64_t> mirror_seen;    // device ring rows already copied (row numbers < this)
  int Ninit_rows = 1, Ninit_asked = 1;   // initial rows in the mirror / initial draws initialize() was asked for
  std::vector<std::vector<double> > init_x;            // the extra initial draws of initialize(n > 1): [k][chain*dim]
  std::vector<std::vector<double> > init_ll, init_lp;  // [k][chain]

  class rung_view : public chain {
    parallel_tempering_chains* p;
    int i, w;
    std::shared_ptr<philox_random> rng;

   public:
    rung_view(parallel_tempering_chains* p, int i, int w = 0) : p(p), i(i), w(w), rng(new philox_random) {}
    void step() override { std::cout << "rung_view::step: step the ladder, not a rung" << std::endl; exit(1); }
    size_t at() const { return (size_t)i * p->W + w; }
    // elem < 0 (or out of range): the current state; raw indexing: row `elem` of the saved history (host mirror, kept when
    // host-side proposals are in use); nominal indexing: step `elem` -> row Ninit + elem / add_every_N (chain.cc:1041-1050)
    bool hist_row(int elem, bool raw, size_t& row) {
      if (!p->host_mode || elem < 0) return false;
      const mirror_t& m = p->mirror[at()];
      const long r = raw ? elem : p->Ninit_rows + elem / p->add_every_N;
      if (r < 0 || (size_t)r >= m.rows()) return false;
      row = (size_t)r;
      return true;
    }
    state getState(int elem = -1, bool raw = false) override {
      size_t r;
      if (hist_row(elem, raw, r)) return state::from_engine(p->sp, p->mirror[at()].x.data() + r * p->dim, p->dim);
      p->refresh();
      return state::from_engine(p->sp, p->X.data() + at() * p->dim, p->dim);
    }
    double getLogPost(int elem = -1, bool raw = false) override {
      size_t r;
      if (hist_row(elem, raw, r)) { const mirror_t& m = p->mirror[at()]; const double t = m.beta[r] * m.llike[r]; return m.lprior[r] + t; }
      p->refresh();
      return p->lpost[at()];
    }
    double getLogLike(int elem = -1, bool raw = false) override {
      size_t r;
      if (hist_row(elem, raw, r)) return p->mirror[at()].llike[r];
      p->refresh();
      return p->llike[at()];
    }
    double invTemp() override { return p->cur_beta(i, w); }
    int getStep() override { return p->nstep; }
    int size() override { return p->host_mode ? (int)p->mirror[at()].rows() : p->nstep + 1; }
    int getDim() override { return p->dim; }
    int get_id() override { return (int)(w * p->Ntemps + i); }
    std::shared_ptr<Random> getPRNG() override { return rng; }
    void reseat(uint64_t seed, uint64_t step) { rng->reseat(seed, (uint32_t)((uint64_t)(w + p->Woff) * p->Ntemps + i), step); }
    double getMAPlpost() override { p->refresh_map(); return p->mlpost[at()]; }
    state getMAPstate() override { p->refresh_map(); return state::from_engine(p->sp, p->mX.data() + at() * p->dim, p->dim); }
  };
  std::vector<rung_view> views;
  // one replica's ladder as a chain: what a proposal that mixes the parallel chains' histories is given (chain.cc:1376-1378)
  class ladder_view : public chain {
    parallel_tempering_chains* p;
    int w;

   public:
    ladder_view(parallel_tempering_chains* p, int w) : p(p), w(w) {}
    void step() override { std::cout << "ladder_view::step: step the ladder itself" << std::endl; exit(1); }
    chain* cold() { return &p->views[(size_t)w * p->Ntemps]; }
    state getState(int elem = -1, bool raw = false) override { return cold()->getState(elem, raw); }
    double getLogPost(int elem = -1, bool raw = false) override { return cold()->getLogPost(elem, raw); }
    double getLogLike(int elem = -1, bool raw = false) override { return cold()->getLogLike(elem, raw); }
    int getStep() override { return p->nstep; }
    int size() override { return cold()->size(); }
    int getDim() override { return p->dim; }
    int multiplicity() override { return p->Ntemps; }
    chain* subchain(int index) override { return &p->views[(size_t)w * p->Ntemps + index]; }
    double getMAPlpost() override { return cold()->getMAPlpost(); }
    state getMAPstate() override { return cold()->getMAPstate(); }
  };
  std::vector<ladder_view> ladders;

  // the engine's callbacks (ptm_propose_batch_fn / ptm_proposal_result_fn)
  static void propose_trampoline(void* self, int n, int dim, const double* X_cur, const int32_t* rung, const int32_t* walker, uint64_t step,
                                 double* X_prop, double* log_hastings, int32_t* type, int32_t* valid) {
    parallel_tempering_chains* p = (parallel_tempering_chains*)self;
    p->fresh = false;   // (the temperatures an evolving ladder shows its proposals are this step's)
    for (int k = 0; k < n; k++) {
      const size_t v = (size_t)(walker[k] - p->Woff) * p->Ntemps + rung[k];   // (the engine names the GLOBAL walker)
      rung_view& view = p->views[v];
      view.reseat(p->eng_seed, step);
      state s = state::from_engine(p->sp, X_cur + (size_t)k * dim, dim);
      proposal_distribution* prop = p->props[v];
      state out = prop->draw(s, &view);
      for (int d = 0; d < dim; d++) X_prop[(size_t)k * dim + d] = out.get_param(d);
      log_hastings[k] = prop->log_hastings_ratio();
      type[k] = prop->type();
      valid[k] = out.invalid() ? 0 : 1;
    }
  }
  static void result_trampoline(void* self, int n, const int32_t* rung, const int32_t* walker, const int32_t* accepted) {
    parallel_tempering_chains* p = (parallel_tempering_chains*)self;
    for (int k = 0; k < n; k++) {
      proposal_distribution* prop = p->props[(size_t)(walker[k] - p->Woff) * p->Ntemps + rung[k]];
      if (accepted[k]) prop->accept(); else prop->reject();   // chain.cc:1009,1015
    }
  }
  // append what the last step(s) saved to the host mirror (rows the ring has not lost yet: call after every step)
  std::vector<double> rx, rl, rp, rb;
  std::vector<int32_t> rmeta;
  std::vector<int64_t> rnsize;
  void sync_history() { fetch_history(); append_history(); }
  void fetch_history() {
    const size_t HC = (size_t)Ntemps * W, cap = ring_rows;
    rx.resize(cap * HC * dim); rl.resize(cap * HC); rp.resize(cap * HC); rb.resize(cap * HC); rmeta.resize(cap * HC * 4); rnsize.resize(HC);
    ptm_check(ptm_get_array(eng, PTM_ARR_NSIZE, rnsize.data()), "sync_history");
    ptm_check(ptm_get_history(eng, rx.data(), rl.data(), rp.data(), rmeta.data()), "sync_history");
    ptm_check(ptm_get_history_invtemps(eng, rb.data()), "sync_history");
  }
  void append_history() {
    const size_t HC = (size_t)Ntemps * W, cap = ring_rows;
    for (size_t c = 0; c < HC; c++) {
      mirror_t& m = mirror[c];
      for (int64_t row = mirror_seen[c]; row < rnsize[c]; row++) {
        const size_t o = (size_t)(row % (int64_t)cap) * HC + c;
        if (rmeta[4 * o + 3] != (int32_t)row) { std::cout << "parallel_tempering_chains: the history ring lost rows before they were mirrored (step() n > ring / 2?)" << std::endl; exit(1); }
        m.x.insert(m.x.end(), rx.begin() + o * dim, rx.begin() + (o + 1) * dim);
        m.llike.push_back(rl[o]); m.lprior.push_back(rp[o]); m.beta.push_back(rb[o]);
        m.meta.push_back(rmeta[4 * o]); m.meta.push_back(rmeta[4 * o + 1]); m.meta.push_back(rmeta[4 * o + 2]);
      }
      mirror_seen[c] = rnsize[c];
    }
  }

  std::vector<double> mX, mlpost;   // host copy of the MAP states (chain.cc:931-934), all rungs / replicas
  bool map_fresh = false;
  void refresh_map() {
    if (map_fresh) return;
    mX.resize((size_t)Ntemps * W * dim); mlpost.resize((size_t)Ntemps * W);
    ptm_check(ptm_get_map(eng, mX.data(), mlpost.data(), nullptr, nullptr), "parallel_tempering_chains");
    map_fresh = true;
  }
  // evolve_temps (chain.hh:302-307): each replica's ladder owns its temperatures; betas [W][Ntemps] is their host copy
  double ev_rate = 0, ev_cut = -1;
  std::vector<double> betas;
  double cur_beta(int i, int w) {
    if (!(ev_rate > 0)) return 1 / temps[i];
    refresh();
    return betas[(size_t)w * Ntemps + i];
  }
  void refresh() {
    if (fresh) return;
    ptm_check(ptm_batch_begin(eng), "parallel_tempering_chains");
    if (ev_rate > 0) {
      betas.resize((size_t)W * Ntemps);
      ptm_check(ptm_get_invtemps(eng, betas.data()), "parallel_tempering_chains");
    }
    ptm_check(ptm_get_states(eng, X.data()), "parallel_tempering_chains");
    ptm_check(ptm_get_array(eng, PTM_ARR_LLIKE, llike.data()), "parallel_tempering_chains");
    ptm_check(ptm_get_array(eng, PTM_ARR_LPOST, lpost.data()), "parallel_tempering_chains");
    ptm_check(ptm_batch_end(eng), "parallel_tempering_chains");
    fresh = true;
  }

 public:
  parallel_tempering_chains(int Ntemps, double Tmax, double swap_rate = 0.01, int add_every_N = 1, bool do_evid = false,
                            bool verbose_evid = true, double dpriormin = -30)
      : Ntemps(Ntemps), add_every_N(add_every_N), Tmax(Tmax), swap_rate(swap_rate), dpriormin(dpriormin), eng(nullptr),
        sp(nullptr), dim(0), nstep(0), hist_rows(0), W(1), temps(Ntemps, 1.0), fresh(false) {
    // geometric ladder, chain.cc:1181-1183
    double tratio = Ntemps > 1 ? std::exp(std::log(Tmax) / (Ntemps - 1)) : 1.0;
    for (int i = 1; i < Ntemps; i++) temps[i] = temps[i - 1] * tratio;
  }
  ~parallel_tempering_chains() {
    if (eng) ptm_engine_destroy(eng);
    for (auto p : props) delete p;
  }
  // Keep what MH_chain::add_state saves (every add_every_N-th state of every rung, chain.cc:935-946) on the device, in a
  // ring of `rows_per_chain` rows per rung; call before initialize().  The reference keeps the whole history in host
  // vectors; dumpChain() below writes the same file from the ring.
  // `coldest_rungs` > 0 keeps the history of that many rungs from the cold end only (what the sampler's pt_dump_n asks
  // for): less device memory, smaller read-backs.
  void keep_history(int rows_per_chain, int coldest_rungs = 0) {
    hist_rows = rows_per_chain;
    hist_rungs = coldest_rungs > 0 && coldest_rungs < Ntemps ? coldest_rungs : 0;
  }
  int history_rungs() const { return hist_rows > 0 ? ((hist_rungs > 0 && !de_built) ? hist_rungs : Ntemps) : 0; }   // (differential evolution on the device: every rung's ring)
  // Run `n` independent replicas of the ladder in one engine (call before initialize()).  Replica w uses the random
  // streams of walker w; every accessor below takes the replica as an optional last argument (default 0).  Multiples
  // of 64 fill whole wavefronts and take the fast kernels.
  void set_replicas(int n) { W = n < 1 ? 1 : n; }
  // Several GPUs: one process per GPU, each with its own share of the replicas -- replicas `begin .. begin + n - 1` of the
  // population (call before initialize(); `device` < 0: the process's current device, e.g. through HIP_VISIBLE_DEVICES).
  // Replicas are independent ladders, so there is nothing to exchange between the processes, and a replica's chains are the
  // very ones a single engine holding the whole population gives it (ptm_config.walker_begin: its random streams are those of
  // the GLOBAL replica index) -- evolving ladders and host-side proposals included.
  void set_replica_range(int begin, int n, int device = -1) { Woff = begin < 0 ? 0 : begin; W = n < 1 ? 1 : n; eng_device = device; }
  int replica_begin() const { return Woff; }
  // Keep the reference's exchange diagnostics (which instance sits on which rung, round-trip directions, ups / downs)
  // for replica 0: costs one read-back of the step's candidate log per step, so it is off by default.
  void track_exchanges(bool on) {
    tracking = on;
    directions.assign(Ntemps, 0); instances.resize(Ntemps); ups.assign(Ntemps, 0); downs.assign(Ntemps, 0);
    for (int i = 0; i < Ntemps; i++) instances[i] = i;                       // chain.cc:1346-1356
    if (Ntemps > 0) { directions[0] = -1; directions[Ntemps - 1] = 1; }
    if (Ntemps == 1) directions[0] = -1;
  }
  // parallel_tempering_chains::evolve_temps (chain.hh:302-307): after every accepted exchange the gap between the two
  // temperatures is pried apart by (1 + rate) and the ladder renormalised (pry_temps, chain.cc:1501-1518,1809-1846).
  // Before or after initialize().  lpost_cut >= 0: the posterior-ordering cut (chain.cc:1819-1827).
  bool evolve_temps(double rate = 0.01, double lpost_cut = -1) {
    ev_rate = rate; ev_cut = lpost_cut;
    if (eng) { ptm_check(ptm_set_evolve_temps(eng, ev_rate, ev_cut), "evolve_temps"); fresh = false; }
    return true;
  }
  const std::vector<int>& getInstances() const { return instances; }
  const std::vector<int>& getDirections() const { return directions; }
  // parallel_tempering_chains::dumpTempStats (chain.cc:2025-2040): T, up fraction, swap acceptance of the pair above
  void dumpTempStats(std::ostream& os) {
    std::vector<int64_t> tries, acc;
    swap_counts(tries, acc);
    os << "#T0 up_frac0 up-swap_ratio0-1" << std::endl;
    for (int i = 0; i < Ntemps; i++) {
      double up_frac;
      if (i == 0) up_frac = 1;
      else if (i == Ntemps - 1) up_frac = 0;
      else up_frac = tracking ? ups[i] / (double)(ups[i] + downs[i]) : 0.0 / 0.0;
      os << 1 / cur_beta(i, 0) << " " << up_frac << " ";
      if (i < Ntemps - 1) os << acc[i] / (double)tries[i] << ": ";
      os << std::endl;
    }
    os << "\n" << std::endl;
  }
  int replicas() const { return W; }
  // chain::checkpoint / restart (chain.cc:1213-1290): everything the ladder's future depends on, into <path>chain0-cp/
  // PTchain.cp.  NOT interchangeable with the reference's checkpoint files: those hold its newran generators and growing
  // history vectors; this one holds the engine's arrays (states, llikes, MH_chain counters, step count = position of every
  // random stream, exchange counters, evolving temperatures, history ring, MAPs, the exchange diagnostics, and -- with
  // host-side proposals -- the host history mirror; the proposals checkpoint themselves beside it as in the reference) --
  // the continued run is the uninterrupted run, bit for bit.  The header names the format version and every parameter a
  // restart must share (sizes, seed, swap rate, add_every_N, ring shape, evolution, proposal placement); the whole file is
  // read and checked BEFORE anything is applied to the engine.
  struct cp_header {
    char magic[8];
    int32_t version, Ntemps, W, dim, add_every_N, ring_rows, ring_rungs, evolving, tracking, host_mode, Ninit_rows, nstep;
    uint64_t seed, engine_step;
    double swap_rate, Tmax;
  };
  void checkpoint(const std::string& path) {
    const std::string dir = path + "chain0-cp/";
    mkdir(dir.c_str(), 0777);
    std::ofstream os((dir + "PTchain.cp").c_str(), std::ios::binary);
    if (!os) { std::cout << "parallel_tempering_chains::checkpoint: cannot write " << dir << "PTchain.cp" << std::endl; exit(1); }
    const size_t N = (size_t)Ntemps * W, np = (size_t)W * (Ntemps > 1 ? Ntemps - 1 : 1);
    auto wr = [&](const void* ptr, size_t bytes) { os.write((const char*)ptr, (std::streamsize)bytes); };
    cp_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "PTMGPU02", 8);
    h.version = 2; h.Ntemps = Ntemps; h.W = W; h.dim = dim; h.add_every_N = add_every_N; h.ring_rows = ring_rows; h.ring_rungs = ring_rungs;
    h.evolving = ev_rate > 0 ? 1 : 0; h.tracking = tracking ? 1 : 0; h.host_mode = host_mode ? 1 : 0; h.Ninit_rows = Ninit_rows; h.nstep = nstep;
    h.seed = eng_seed; h.engine_step = ptm_step_count(eng); h.swap_rate = swap_rate; h.Tmax = Tmax;
    wr(&h, sizeof h);
    std::vector<double> x(N * dim), ll(N);
    std::vector<int32_t> nt(N), na(N), ty(N);
    std::vector<int64_t> nh(N), st(np), sa(np);
    ptm_check(ptm_get_states(eng, x.data()), "checkpoint");
    ptm_check(ptm_get_array(eng, PTM_ARR_LLIKE, ll.data()), "checkpoint");
    ptm_check(ptm_get_array(eng, PTM_ARR_NTRIES, nt.data()), "checkpoint");
    ptm_check(ptm_get_array(eng, PTM_ARR_NACCEPT, na.data()), "checkpoint");
    ptm_check(ptm_get_array(eng, PTM_ARR_LAST_TYPE, ty.data()), "checkpoint");
    ptm_check(ptm_get_array(eng, PTM_ARR_NHIST, nh.data()), "checkpoint");
    ptm_check(ptm_get_swap_counts(eng, st.data(), sa.data()), "checkpoint");
    wr(x.data(), x.size() * 8); wr(ll.data(), N * 8); wr(nt.data(), N * 4); wr(na.data(), N * 4); wr(ty.data(), N * 4);
    wr(nh.data(), N * 8); wr(st.data(), np * 8); wr(sa.data(), np * 8);
    if (ev_rate > 0) {
      std::vector<double> b((size_t)W * Ntemps);
      ptm_check(ptm_get_invtemps(eng, b.data()), "checkpoint");
      wr(b.data(), b.size() * 8);
    }
    if (ring_rows > 0) {
      const size_t n = (size_t)ring_rows * ring_rungs * W;
      std::vector<double> gx(n * dim), gl(n), gp(n), gb(n);
      std::vector<int32_t> gm(n * 4);
      ptm_check(ptm_get_history(eng, gx.data(), gl.data(), gp.data(), gm.data()), "checkpoint");
      ptm_check(ptm_get_history_invtemps(eng, gb.data()), "checkpoint");
      wr(gx.data(), gx.size() * 8); wr(gl.data(), n * 8); wr(gp.data(), n * 8); wr(gm.data(), gm.size() * 4); wr(gb.data(), n * 8);
    }
    {
      std::vector<double> mx(N * dim), mp(N), ml(N), mr(N);
      ptm_check(ptm_get_map(eng, mx.data(), mp.data(), ml.data(), mr.data()), "checkpoint");
      wr(mx.data(), mx.size() * 8); wr(mp.data(), N * 8); wr(ml.data(), N * 8); wr(mr.data(), N * 8);
    }
    if (tracking) {
      std::vector<int64_t> u(ups.begin(), ups.end()), d(downs.begin(), downs.end());
      wr(directions.data(), (size_t)Ntemps * 4); wr(instances.data(), (size_t)Ntemps * 4); wr(u.data(), (size_t)Ntemps * 8); wr(d.data(), (size_t)Ntemps * 8);
    }
    if (host_mode) {
      for (size_t c = 0; c < N; c++) {
        const mirror_t& m = mirror[c];
        const uint64_t rows = m.rows();
        wr(&rows, 8); wr(&mirror_seen[c], 8);
        wr(m.x.data(), m.x.size() * 8); wr(m.llike.data(), rows * 8); wr(m.lprior.data(), rows * 8); wr(m.beta.data(), rows * 8); wr(m.meta.data(), m.meta.size() * 4);
      }
      for (auto pp : props) pp->checkpoint(dir);
    }
    if (!os) { std::cout << "parallel_tempering_chains::checkpoint: write failed" << std::endl; exit(1); }
  }
  // into a ladder set up exactly like the one that was saved (same sizes, seed, likelihood, prior, proposal, evolve_temps)
  void restart(const std::string& path) {
    const std::string dir = path + "chain0-cp/", fn = dir + "PTchain.cp";
    std::ifstream is(fn.c_str(), std::ios::binary);
    if (!is) { std::cout << "parallel_tempering_chains::restart: cannot read " << fn << std::endl; exit(1); }
    std::vector<char> buf((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    bool ok = true;
    auto rd = [&](void* ptr, size_t bytes) { if (pos + bytes > buf.size()) { ok = false; return; } memcpy(ptr, buf.data() + pos, bytes); pos += bytes; };
    cp_header h;
    rd(&h, sizeof h);
    if (!ok || std::string(h.magic, 8) != "PTMGPU02" || h.version != 2) { std::cout << "parallel_tempering_chains::restart: " << fn << " is not a checkpoint of this build (format 2)" << std::endl; exit(1); }
    if (h.Ntemps != Ntemps || h.W != W || h.dim != dim || h.add_every_N != add_every_N || h.ring_rows != ring_rows || h.ring_rungs != ring_rungs ||
        h.evolving != (ev_rate > 0 ? 1 : 0) || h.host_mode != (host_mode ? 1 : 0) || h.seed != eng_seed || h.swap_rate != swap_rate || h.Tmax != Tmax) {
      std::cout << "parallel_tempering_chains::restart: " << fn << " was written by a differently configured ladder (sizes, seed, swap rate, Tmax, "
                << "history ring, evolution and proposal placement must all match)" << std::endl;
      exit(1);
    }
    const size_t N = (size_t)Ntemps * W, np = (size_t)W * (Ntemps > 1 ? Ntemps - 1 : 1);
    std::vector<double> x(N * dim), ll(N), b, gx, gl, gp, gb, mx(N * dim), mp(N), ml(N), mr(N);
    std::vector<int32_t> nt(N), na(N), ty(N), gm, dir_(Ntemps), ins_(Ntemps);
    std::vector<int64_t> nh(N), st(np), sa(np), u(Ntemps), d(Ntemps);
    rd(x.data(), x.size() * 8); rd(ll.data(), N * 8); rd(nt.data(), N * 4); rd(na.data(), N * 4); rd(ty.data(), N * 4);
    rd(nh.data(), N * 8); rd(st.data(), np * 8); rd(sa.data(), np * 8);
    if (ev_rate > 0) { b.resize((size_t)W * Ntemps); rd(b.data(), b.size() * 8); }
    const size_t nring = (size_t)ring_rows * ring_rungs * W;
    if (ring_rows > 0) {
      gx.resize(nring * dim); gl.resize(nring); gp.resize(nring); gb.resize(nring); gm.resize(nring * 4);
      rd(gx.data(), gx.size() * 8); rd(gl.data(), nring * 8); rd(gp.data(), nring * 8); rd(gm.data(), gm.size() * 4); rd(gb.data(), nring * 8);
    }
    rd(mx.data(), mx.size() * 8); rd(mp.data(), N * 8); rd(ml.data(), N * 8); rd(mr.data(), N * 8);
    if (h.tracking) { rd(dir_.data(), (size_t)Ntemps * 4); rd(ins_.data(), (size_t)Ntemps * 4); rd(u.data(), (size_t)Ntemps * 8); rd(d.data(), (size_t)Ntemps * 8); }
    std::vector<mirror_t> mir;
    std::vector<int64_t> seen;
    if (host_mode) {
      mir.resize(N); seen.resize(N);
      for (size_t c = 0; c < N && ok; c++) {
        uint64_t rows = 0;
        rd(&rows, 8); rd(&seen[c], 8);
        if (!ok || rows > buf.size()) { ok = false; break; }
        mirror_t& m = mir[c];
        m.x.resize(rows * dim); m.llike.resize(rows); m.lprior.resize(rows); m.beta.resize(rows); m.meta.resize(rows * 3);
        rd(m.x.data(), m.x.size() * 8); rd(m.llike.data(), rows * 8); rd(m.lprior.data(), rows * 8); rd(m.beta.data(), rows * 8); rd(m.meta.data(), m.meta.size() * 4);
      }
    }
    if (!ok || pos != buf.size()) { std::cout << "parallel_tempering_chains::restart: " << fn << " is truncated or has trailing bytes; nothing was applied" << std::endl; exit(1); }
    // ---- the file is whole and matches: apply
    ptm_check(ptm_restore(eng, x.data(), ll.data(), nt.data(), na.data(), ty.data(), nh.data(), h.engine_step, st.data(), sa.data()), "restart");
    if (ev_rate > 0) ptm_check(ptm_set_invtemps(eng, b.data()), "restart");
    if (ring_rows > 0) ptm_check(ptm_set_history(eng, gx.data(), gl.data(), gp.data(), gm.data(), gb.data()), "restart");
    ptm_check(ptm_set_map(eng, mx.data(), mp.data(), ml.data(), mr.data()), "restart");
    if (h.tracking) {
      track_exchanges(true);
      directions.assign(dir_.begin(), dir_.end()); instances.assign(ins_.begin(), ins_.end());
      ups.assign(u.begin(), u.end()); downs.assign(d.begin(), d.end());
    }
    if (host_mode) {
      mirror.swap(mir); mirror_seen.swap(seen); Ninit_rows = h.Ninit_rows;
      for (auto pp : props) pp->restart(dir);
    }
    nstep = h.nstep;
    fresh = hist_fresh = map_fresh = false;
  }
  // chain.cc:1281-1365: n prior draws per rung (the device draws them: any prior type but the improper flat one); the last one
  // is the chain's start, the others seed the history that history-reading proposals draw from (ptmcmc.cc:86: de_ni * Npar)
  bayes_likelihood* init_like = nullptr;
  const sampleable_probability_function* init_prior = nullptr;
  bool prior_on_host = false;   // the prior is not a per-dimension product: evaluated through ptm_set_prior_callback
  // C-ABI trampoline of a host-evaluated prior: probability_function::evaluate_log on every (valid) state of the batch
  static void prior_trampoline(void* self, const double* X, int n, int dim, double* out) {
    const sampleable_probability_function* pr = (const sampleable_probability_function*)self;
    for (int k = 0; k < n; k++) {
      state s = state::from_engine(pr->get_space(), X + (size_t)k * dim, dim);   // (enforced on the device already)
      out[k] = pr->evaluate_log(s);
    }
  }
  std::vector<double> init_start;
  bool have_start = false;
  // tell the ladder BEFORE initialize() that its proposal will be drawn on the host (saves building the engine twice)
  void use_host_proposals(bool on = true) { want_host = on; }
  // ... or that its set holds a differential evolution the DEVICE draws (ptm_set_proposal_de): the engine then records every rung's
  // history (keep_history must hold the whole run) and keeps the n - 1 extra draws of initialize(n) on the device
  void use_device_de(bool on = true) { want_de = on; }
  bool draws_de_on_device() const { return de_built; }
  void initialize(bayes_likelihood* log_likelihood, const sampleable_probability_function* log_prior, int n = 1, uint64_t seed = 0x5EED0001ull,
                  const std::vector<double>* start_states = nullptr) {
    init_like = log_likelihood; init_prior = log_prior; Ninit_asked = n < 1 ? 1 : n; eng_seed = seed;
    have_start = start_states != nullptr;
    if (have_start) init_start = *start_states;
    build_engine(want_host);
  }

 private:
  void build_engine(bool host) {
    if (eng) { ptm_engine_destroy(eng); eng = nullptr; }
    bayes_likelihood* log_likelihood = init_like;
    const sampleable_probability_function* log_prior = init_prior;
    sp = log_prior->get_space();
    dim = log_prior->getDim();
    host_mode = host;
    Ninit_rows = host ? Ninit_asked : 1;
    // differential evolution on the device: every rung's history stays there, in a ring as long as the run (keep_history), with the
    // extra draws of initialize(n) beside it; the rows keep the ring's numbering (row 0 = the start state)
    de_built = !host && want_de && hist_rows > 0;
    const int draws = (host || de_built) ? Ninit_asked : 1;
    // host-side proposals: every rung's saved rows pass through a SHORT device ring into the host mirror after each step
    ring_rows = host ? 8 : hist_rows;
    ring_rungs = (host || de_built) ? Ntemps : history_rungs();
    ptm_config cfg = ptm_config();   // (zeroed: fields a newer ABI adds default to 0)
    cfg.struct_size = sizeof cfg;
    cfg.dim = dim; cfg.n_rungs = Ntemps; cfg.rung_begin = 0; cfg.rung_count = Ntemps; cfg.n_walkers = W; cfg.seed = eng_seed;
    cfg.swap_rate = swap_rate; cfg.add_every_n = add_every_N; cfg.min_prior = dpriormin; cfg.device = eng_device; cfg.stream = nullptr;
    cfg.walker_begin = Woff;
    cfg.time_kernels = 0; cfg.swap_log_steps = 0; cfg.exchange_row_capacity = 0; cfg.history_rungs = ring_rungs; cfg.history_capacity = ring_rows; cfg.map_rungs = Ntemps;
    ptm_check(ptm_engine_create(&cfg, &eng), "parallel_tempering_chains::initialize");
    std::vector<int> lo(dim), hi(dim), types;
    std::vector<double> xmin(dim), xmax(dim), centers, halfwidths;
    for (int i = 0; i < dim; i++) {
      boundary b = sp ? sp->get_bound(i) : boundary();
      lo[i] = b.lower(); hi[i] = b.upper();
      b.getDomainLimits(xmin[i], xmax[i]);
      if (lo[i] == boundary::open && hi[i] == boundary::open) xmin[i] = xmax[i] = 0;
    }
    ptm_check(ptm_set_bounds(eng, lo.data(), hi.data(), xmin.data(), xmax.data()), "set_bounds");
    prior_on_host = !log_prior->describe(types, centers, halfwidths);
    if (!prior_on_host) {
      ptm_check(ptm_set_prior(eng, types.data(), centers.data(), halfwidths.data()), "set_prior");
      if (!log_likelihood->describe_device_target(eng))
        ptm_check(ptm_set_target_callback(eng, &bayes_likelihood::batch_trampoline, log_likelihood), "set_target_callback");
    } else {
      // a prior that is not a per-dimension product: evaluated on the host between the propose and the accept kernel, and so
      // is the likelihood then (whatever it is: its evaluate_log)
      ptm_check(ptm_set_target_callback(eng, &bayes_likelihood::batch_trampoline, log_likelihood), "set_target_callback");
      ptm_check(ptm_set_prior_callback(eng, &parallel_tempering_chains::prior_trampoline, (void*)log_prior), "set_prior_callback");
    }
    std::vector<double> beta(Ntemps);
    for (int i = 0; i < Ntemps; i++) beta[i] = 1 / temps[i];  // chain.cc:1340
    ptm_check(ptm_set_ladder(eng, beta.data()), "set_ladder");
    if (ev_rate > 0) ptm_check(ptm_set_evolve_temps(eng, ev_rate, ev_cut), "evolve_temps");
    const size_t N = (size_t)Ntemps * W;
    X.assign(N * dim, 0.0); llike.assign(N, 0.0); lpost.assign(N, 0.0);
    init_x.clear(); init_ll.clear(); init_lp.clear();
    if (have_start) {
      ptm_check(ptm_set_states(eng, init_start.data(), nullptr), "set_states");
      Ninit_rows = 1;
    } else if (prior_on_host) {
      // MH_chain::initialize (chain.cc:846-876) on the host: every start state is a draw of the prior's own drawSample with the
      // chain's generator, redrawn while the likelihood is below -1e100 (:856-869); draws 1 .. n-1 are kept for the mirror
      std::vector<double> x0(N * dim);
      for (int k = draws - 1; k >= 0; k--) {
        std::vector<double> x(N * dim), ll(N), lp(N);
        for (size_t c = 0; c < N; c++) {
          philox_random rng;
          rng.reseat(eng_seed, (uint32_t)((c % W + Woff) * Ntemps + c / W), ((uint64_t)1 << 40) + (uint64_t)k);   // (a step no sweep reaches)
          for (int attempt = 0;; attempt++) {
            state s = log_prior->drawSample(rng);
            const double l = s.invalid() ? -INFINITY : log_likelihood->evaluate_log(s);
            if (!(l < -1e100)) {
              for (int d = 0; d < dim; d++) x[c * dim + d] = s.get_param(d);
              ll[c] = l; lp[c] = log_prior->evaluate_log(s);
              break;
            }
            if (attempt >= 100000) { std::cout << "parallel_tempering_chains::initialize: no valid start state drawn from the prior" << std::endl; exit(1); }
          }
        }
        if (k > 0) { init_x.push_back(x); init_ll.push_back(ll); init_lp.push_back(lp); }
        else x0 = x;
      }
      ptm_check(ptm_set_states(eng, x0.data(), nullptr), "set_states");
    } else {
      // draws 1 .. n-1 first (kept for the mirror; all of them in ONE call, the order of the rows as before: n-1 first), draw 0 last:
      // it is the chain's start, whatever n is
      if (draws > 1) {
        const size_t nd = (size_t)draws - 1;
        std::vector<double> ax(nd * N * dim), al(nd * N), ap(nd * N);
        int rc = ptm_draw_prior_rows(eng, 1, (int)nd, ax.data(), al.data(), ap.data());
        if (rc == PTM_ERR_UNSUPPORTED) {
          std::cout << "parallel_tempering_chains::initialize: " << ptm_last_error() << "; pass start states" << std::endl;
          exit(1);
        }
        ptm_check(rc, "draw_prior_rows");
        for (size_t k = nd; k >= 1; k--) {
          const size_t o = k - 1;   // (row of draw k)
          init_x.push_back(std::vector<double>(ax.begin() + o * N * dim, ax.begin() + (o + 1) * N * dim));
          init_ll.push_back(std::vector<double>(al.begin() + o * N, al.begin() + (o + 1) * N));
          init_lp.push_back(std::vector<double>(ap.begin() + o * N, ap.begin() + (o + 1) * N));
        }
      }
      int rc = ptm_init_from_prior_k(eng, 0);
      if (rc == PTM_ERR_UNSUPPORTED) {
        std::cout << "parallel_tempering_chains::initialize: " << ptm_last_error() << "; pass start states" << std::endl;
        exit(1);
      }
      ptm_check(rc, "init_from_prior");
    }
    views.clear(); ladders.clear();
    for (int w = 0; w < W; w++) {
      for (int i = 0; i < Ntemps; i++) views.push_back(rung_view(this, i, w));   // views[w*Ntemps + i]
      ladders.push_back(ladder_view(this, w));
    }
    fresh = hist_fresh = map_fresh = false;
    if (host) {
      // the mirror starts with the initial draws (MH_chain::initialize adds every one of them, chain.cc:846-876)
      mirror.assign(N, mirror_t());
      mirror_seen.assign(N, 0);
      for (size_t c = 0; c < N; c++) {
        const int i = (int)(c / W);
        for (size_t k = 0; k < init_x.size(); k++) {
          mirror_t& m = mirror[c];
          m.x.insert(m.x.end(), init_x[k].begin() + c * dim, init_x[k].begin() + (c + 1) * dim);
          m.llike.push_back(init_ll[k][c]); m.lprior.push_back(init_lp[k][c]); m.beta.push_back(beta[i]);
          m.meta.push_back(1); m.meta.push_back(1); m.meta.push_back(-1);
        }
      }
      sync_history();   // row 0 of the ring: the start state
    }
  }

 public:
  // chain.cc:1367-1386: one clone per rung (and replica); a proposal that supports mixing is given the whole ladder as its
  // chain, any other its own rung (set_chain).  Gaussians with a fixed factor -- and sets of scalar multiples of one -- are
  // drawn on the device; everything else through the host-proposal step.
  void set_proposal(proposal_distribution& proposal) {
    for (auto p : props) delete p;
    props.clear();
    int kind = 0;
    double odf = 0;
    std::vector<double> f, all, odfs(Ntemps);
    bool on_device = true;
    for (int i = 0; i < Ntemps && on_device; i++) {
      const user_gaussian_prop* ug = dynamic_cast<const user_gaussian_prop*>(&proposal);
      if (!(ug ? ug->device_describe_in(sp, dim, kind, f, odf) : proposal.device_describe(dim, kind, f, odf))) { on_device = false; break; }
      all.insert(all.end(), f.begin(), f.end());
      odfs[i] = odf;
    }
    {
      ptm_de_params q;
      if (on_device && proposal.device_describe_de(q)) {
        // a set with a differential evolution for the device: possible if the ring holds the run and the dimensions fit the kernel
        // that draws it (ptm_set_proposal_de); else the whole set is drawn on the host
        if (hist_rows <= 0 || dim > 32) on_device = false;
        else if (!de_built) { want_de = true; build_engine(false); }   // (initialize() did not know: set up again, with the same draws)
      }
    }
    if (on_device) {
      if (host_mode) build_engine(false);
      for (int i = 0; i < Ntemps; i++) props.push_back(proposal.clone());
      ptm_check(ptm_set_proposals(eng, kind, all.data(), odfs.data()), "set_proposals");
      std::vector<double> cum, sc, od;
      if (proposal.device_describe_mixture(dim, cum, sc, od)) {   // a proposal_distribution_set of scaled Gaussians
        const int K = (int)cum.size();
        std::vector<double> C((size_t)Ntemps * K), S(C.size()), O(C.size());
        for (int i = 0; i < Ntemps; i++)
          for (int k = 0; k < K; k++) { C[(size_t)i * K + k] = cum[k]; S[(size_t)i * K + k] = sc[k]; O[(size_t)i * K + k] = od[k]; }
        ptm_check(ptm_set_proposal_mixture(eng, K, C.data(), S.data(), O.data()), "set_proposal_mixture");
      } else {
        ptm_check(ptm_set_proposal_mixture(eng, 0, nullptr, nullptr, nullptr), "set_proposal_mixture");
      }
      ptm_check(ptm_set_proposal_callback(eng, nullptr, nullptr, nullptr), "set_proposal_callback");
      ptm_de_params q;
      if (proposal.device_describe_de(q)) {
        // the set's differential evolution is drawn on the device from the device's own history: the extra draws of initialize(n),
        // oldest first (init_x), then the ring
        const size_t N = (size_t)Ntemps * W;
        std::vector<double> rows(init_x.size() * N * dim);
        for (size_t k = 0; k < init_x.size(); k++) std::copy(init_x[k].begin(), init_x[k].end(), rows.begin() + k * N * dim);
        ptm_check(ptm_set_proposal_de(eng, &q, (int)init_x.size(), rows.empty() ? nullptr : rows.data()), "set_proposal_de");
      } else {
        ptm_check(ptm_set_proposal_de(eng, nullptr, 0, nullptr), "set_proposal_de");
      }
      return;
    }
    if (!host_mode) build_engine(true);   // (initialize() did not know: the engine is set up again, with the same draws)
    for (int w = 0; w < W; w++)
      for (int i = 0; i < Ntemps; i++) {
        proposal_distribution* c = proposal.clone();
        if (c->support_mixing()) c->set_chain(&ladders[w]);
        else c->set_chain(&views[(size_t)w * Ntemps + i]);
        props.push_back(c);   // props[w*Ntemps + i]
      }
    ptm_check(ptm_set_proposal_callback(eng, &parallel_tempering_chains::propose_trampoline, &parallel_tempering_chains::result_trampoline, this),
              "set_proposal_callback");
  }
  bool proposals_on_host() const { return host_mode; }
  bool prior_evaluated_on_host() const { return prior_on_host; }   // the prior is not a per-dimension product (describe() == false)
  // chain::report_prop (chain.cc:2096-2109 flavour): every rung's proposal report, replica 0
  std::string report_prop(int style = 0) override {
    std::ostringstream ss;
    for (int i = 0; i < Ntemps && i < (int)props.size(); i++) ss << "  T=" << 1 / cur_beta(i, 0) << ": " << props[i]->report(style) << "\n";
    return ss.str();
  }
  // per-rung factors for proposals that differ by rung (what user_gaussian_prop's check_update achieves in the reference)
  void set_proposal_factors(int kind, const std::vector<double>& factors, const std::vector<double>& oneDfracs = std::vector<double>()) {
    ptm_check(ptm_set_proposals(eng, kind, factors.data(), oneDfracs.empty() ? nullptr : oneDfracs.data()), "set_proposals");
  }
  void step() override {
    ptm_check(ptm_step(eng, 1), "parallel_tempering_chains::step");
    nstep++;
    fresh = hist_fresh = map_fresh = false;
    if (host_mode || tracking) {   // what the host needs of this step, read with ONE wait on the device
      ptm_check(ptm_batch_begin(eng), "parallel_tempering_chains::step");
      if (host_mode) fetch_history();
      if (tracking) fetch_swaps();
      ptm_check(ptm_batch_end(eng), "parallel_tempering_chains::step");
      if (host_mode) append_history();
      if (tracking) replay_swaps();
    }
  }
  // n steps in ONE engine call when nothing on the host has to look in between (no host-side proposals, no exchange tracking):
  // small ladders then run many steps per kernel launch (ptm_fused_kernel.hpp) instead of paying the launch and the tables'
  // staging for every step.  Same chains as n calls of step().
  void step_n(int n) {
    if (n <= 0) return;
    if (host_mode || tracking) { for (int k = 0; k < n; k++) step(); return; }
    ptm_check(ptm_step(eng, n), "parallel_tempering_chains::step_n");
    nstep += n;
    fresh = hist_fresh = map_fresh = false;
  }
  void step(int n) {
    if (tracking || host_mode) { for (int k = 0; k < n; k++) step(); return; }
    ptm_check(ptm_step(eng, n), "parallel_tempering_chains::step");
    nstep += n;
    fresh = hist_fresh = map_fresh = false;
  }
  double getMAPlpost() override { return views[0].getMAPlpost(); }   // the cold rung's (chain.cc:1570-1571)
  state getMAPstate() override { return views[0].getMAPstate(); }
  state getState(int = -1, bool = false) override { return views[0].getState(); }
  double getLogPost(int = -1, bool = false) override { return views[0].getLogPost(); }
  double getLogLike(int = -1, bool = false) override { return views[0].getLogLike(); }
  int getStep() override { return nstep; }
  int multiplicity() override { return Ntemps; }
  chain* subchain(int index) override { return subchain(index, 0); }
  chain* subchain(int index, int replica) {
    if (index >= 0 && index < Ntemps && replica >= 0 && replica < W) return &views[(size_t)replica * Ntemps + index];
    std::cout << "parallel_tempering_chains::subchain:index out of range. (" << index << " of " << Ntemps << ")" << std::endl;
    exit(1);
  }
  ptm_engine* engine() { return eng; }
  // MH_chain::dumpChain (chain.cc:1112-1135) for rung `ichain`: one line per saved step i = Nburn, Nburn+ievery, ... :
  //   i lpost llike acceptance_ratio prop_type: p0 ... pD-1 invtemp
  // Rows that have already left the ring are skipped (the ring holds the newest rows_per_chain saved states).
  // The device's history ring -> hx / hl / hp / hmeta / hb (the ring's own layout [row][history chain]), the chains a dump or an
  // effective-sample estimate reads: rungs [0, hist_read_rungs) if the driver has said which it writes out (limit_history_reads:
  // ptmcmc_sampler::run dumps pt_dump_n rungs and estimates on the cold one), every history rung as soon as anybody asks beyond.  With
  // differential evolution on the device the ring holds EVERY rung for the whole run: all of it was 0.1 s per output event at 128
  // temperatures x 4000 rows of 32 parameters, most of a run that reports every 500 steps.
  void read_history(size_t at, const char* who) {
    const size_t HC = (size_t)history_rungs() * W, cap = hist_rows;
    size_t want = hist_read_rungs > 0 ? (size_t)std::min(hist_read_rungs, history_rungs()) * W : HC;
    if (at >= want) want = HC;
    if (hist_fresh && hist_have >= want) return;
    hx.resize(cap * HC * dim); hl.resize(cap * HC); hp.resize(cap * HC); hmeta.resize(cap * HC * 4); hnhist.resize((size_t)Ntemps * W); hb.resize(cap * HC);
    ptm_check(ptm_get_history_chains(eng, 0, (int)want, hx.data(), hl.data(), hp.data(), hmeta.data(), hb.data()), who);   // (hb: the temperature each row was saved at)
    ptm_check(ptm_get_array(eng, PTM_ARR_NHIST, hnhist.data()), who);
    hist_fresh = true;
    hist_have = want;
  }
  void limit_history_reads(int rungs) { hist_read_rungs = rungs; }
  void dumpChain(int ichain, std::ostream& os, int Nburn = 0, int ievery = 1, int replica = 0) {
    if (host_mode) {   // host-side proposals: the whole saved history is in the host mirror
      const size_t at = (size_t)ichain * W + replica;
      const mirror_t& m = mirror[at];
      std::vector<int64_t> nh((size_t)Ntemps * W);
      ptm_check(ptm_get_array(eng, PTM_ARR_NHIST, nh.data()), "dumpChain");
      const int Ninit = Ninit_rows, Nhist = (int)nh[at];
      os << "#Ninit=" << Ninit << ", Nburn=" << Nburn << "\n";
      os << "#eval: log(posterior) log(likelihood) acceptance_ratio prop_type: ";
      for (int i = 0; i < dim; i++) os << (sp ? sp->get_name(i) : std::string("[unnamed]")) << " ";
      os << std::endl;
      if (Nburn + Ninit < 0) Nburn = -Ninit;
      const double invtemp = cur_beta(ichain, replica);
      for (int i = Nburn; i < Nhist; i += ievery) {
        int idx = Ninit + i;
        if (i >= 0) idx = Ninit + i / add_every_N;
        if (idx < 0 || (size_t)idx >= m.rows()) continue;
        const double t = m.beta[idx] * m.llike[idx];
        os << i << " " << m.lprior[idx] + t << " " << m.llike[idx] << " " << m.meta[3 * idx] / (double)m.meta[3 * idx + 1] << " " << m.meta[3 * idx + 2] << ": ";
        for (int j = 0; j < dim - 1; j++) os << m.x[(size_t)idx * dim + j] << " ";
        os << m.x[(size_t)idx * dim + dim - 1] << " " << invtemp << std::endl;
      }
      return;
    }
    if (hist_rows <= 0) { std::cout << "parallel_tempering_chains::dumpChain: call keep_history(rows) before initialize()" << std::endl; exit(1); }
    if (ichain >= history_rungs()) { std::cout << "parallel_tempering_chains::dumpChain: rung " << ichain << " keeps no history (keep_history(rows, " << history_rungs() << "))" << std::endl; exit(1); }
    const size_t HC = (size_t)history_rungs() * W, cap = hist_rows, at = (size_t)ichain * W + replica;
    read_history(at, "dumpChain");   // one read-back serves every rung / replica dumped at this step
    const std::vector<int32_t>& meta = hmeta;
    // (rows are numbered as the ring numbers them -- row 0 the start state --; with differential evolution on the device the header
    //  names the initial draws the device keeps beside the ring, as the reference's file does)
    const int Ninit = 1, Nhist = (int)hnhist[at];
    os << "#Ninit=" << (de_built ? Ninit_asked : Ninit) << ", Nburn=" << Nburn << "\n";
    os << "#eval: log(posterior) log(likelihood) acceptance_ratio prop_type: ";
    for (int i = 0; i < dim; i++) os << (sp ? sp->get_name(i) : std::string("[unnamed]")) << " ";
    os << std::endl;
    const int Nfront = de_built ? Ninit_asked - 1 : 0;              // initial draws in front of the ring's row 0 (kept by the host too: init_x)
    if (Nburn + Ninit + Nfront < 0) Nburn = -(Ninit + Nfront);
    const double invtemp = cur_beta(ichain, replica);   // the chain's CURRENT temperature on every row (chain.cc:1131)
    for (int i = Nburn; i < Nhist; i += ievery) {
      if (i + Ninit < 0) {   // one of the initial draws in front of the start state (MH_chain::initialize adds every one, chain.cc:846-876)
        const int k = Nfront + Ninit + i;
        if (k < 0 || k >= (int)init_x.size()) continue;
        const double t = (1 / temps[ichain]) * init_ll[k][at];
        os << i << " " << init_lp[k][at] + t << " " << init_ll[k][at] << " " << 1.0 << " " << -1 << ": ";
        for (int j = 0; j < dim - 1; j++) os << init_x[k][at * dim + j] << " ";
        os << init_x[k][at * dim + dim - 1] << " " << invtemp << std::endl;
        continue;
      }
      int idx = Ninit + i;                                          // chain.cc:1124-1125
      if (i >= 0) idx = Ninit + i / add_every_N;                    // get_state_idx, chain.cc:1041-1050 (Nzero = 0)
      const size_t o = (size_t)(idx % (int)cap) * HC + at;
      if (idx < 0 || meta[4 * o + 3] != idx) continue;              // not saved yet / overwritten in the ring
      const double lpo = hp[o] + hb[o] * hl[o];                     // chain.cc:928, at the temperature of that add_state
      os << i << " " << lpo << " " << hl[o] << " " << meta[4 * o] / (double)meta[4 * o + 1] << " " << meta[4 * o + 2] << ": ";
      for (int j = 0; j < dim - 1; j++) os << hx[o * dim + j] << " ";
      os << hx[o * dim + dim - 1];
      os << " " << invtemp;
      os << std::endl;
    }
  }
  // ---- effective sample size of the cold chain (chain::report_effective_samples, chain.cc:457-643; the sampler's
  // chain_ess_stop criterion, ptmcmc.cc:628-649): ess_estimator (above) on the cold chain's saved history, the first 20
  // parameters as features.  Returns (ess, useful chain length).  Needs the cold chain's whole saved history: the host mirror
  // (host-side proposals) or a history ring as long as the run (keep_history).
  bool cold_row(int step, std::vector<double>& out, int replica = 0) {   // state the cold chain saved for nominal step `step`
    const size_t at = (size_t)replica;
    if (step < 0) return false;
    if (host_mode) {
      const mirror_t& m = mirror[at];
      const size_t r = (size_t)Ninit_rows + step / add_every_N;
      if (r >= m.rows()) return false;
      out.assign(m.x.begin() + r * dim, m.x.begin() + (r + 1) * dim);
      return true;
    }
    if (hist_rows <= 0) return false;
    const size_t HC = (size_t)history_rungs() * W, cap = hist_rows;
    read_history(at, "report_effective_samples");
    const int idx = 1 + step / add_every_N;
    const size_t o = (size_t)(idx % (int)cap) * HC + at;
    if (hmeta[4 * o + 3] != idx) return false;
    out.assign(hx.begin() + o * dim, hx.begin() + (o + 1) * dim);
    return true;
  }
  int cold_steps(int replica = 0) {   // MH_chain::getStep() of the cold chain: its add_state calls
    std::vector<int64_t> nh((size_t)Ntemps * W);
    ptm_check(ptm_get_array(eng, PTM_ARR_NHIST, nh.data()), "report_effective_samples");
    return (int)nh[(size_t)replica];
  }
  std::pair<double, int> report_effective_samples(int imax = -1, int width = 40000, int every = 100, double esslimit = -1, bool reporting = true) {
    if (imax < 0 || imax > dim) imax = dim;
    if (imax > 20) imax = 20;   // (the reference's simplified interface looks at the first 20 parameters)
    const int nf = imax;
    ess_estimator est(cold_steps(), nf, [this](int step, std::vector<double>& row) { return cold_row(step, row); });
    const std::pair<double, int> r = est.report(width, every, esslimit, views.empty() ? 0 : views[0].size(), Ninit_rows);
    if (reporting)
      std::cout << "Over " << nf << " pars: ess=" << r.first << "  useful chain length is: " << r.second << " autocorrlen=" << (r.first > 0 ? r.second / r.first : 0.0) << std::endl;
    return r;
  }
  // swap_count / swap_accept_count (chain.hh:244-245)
  void swap_counts(std::vector<int64_t>& tries, std::vector<int64_t>& accepts) {
    std::vector<int64_t> t((size_t)W * (Ntemps > 1 ? Ntemps - 1 : 1)), a(t.size());
    ptm_check(ptm_get_swap_counts(eng, t.data(), a.data()), "swap_counts");
    tries.assign(t.begin(), t.begin() + (Ntemps > 1 ? Ntemps - 1 : 1));       // replica 0
    accepts.assign(a.begin(), a.begin() + (Ntemps > 1 ? Ntemps - 1 : 1));
  }
  std::string status() override {  // chain.cc:2053-2094 flavour
    refresh();
    std::vector<int32_t> nt((size_t)Ntemps * W), na((size_t)Ntemps * W);
    ptm_get_array(eng, PTM_ARR_NTRIES, nt.data());
    ptm_get_array(eng, PTM_ARR_NACCEPT, na.data());
    std::ostringstream s;
    for (int i = 0; i < Ntemps; i++) {   // replica 0
      const size_t c = (size_t)i * W;
      s << "T=" << 1 / cur_beta(i, 0) << ": lpost=" << lpost[c] << " llike=" << llike[c] << " acc=" << (double)na[c] / nt[c] << "\n";
    }
    return s.str();
  }
  // MH_chain::dumpChain row format of the current state (chain.cc:1112-1135): i lpost llike acc type: params invtemp
  void dumpCurrent(int ichain, std::ostream& os, int replica = 0) {
    refresh();
    std::vector<int32_t> nt((size_t)Ntemps * W), na((size_t)Ntemps * W), ty((size_t)Ntemps * W);
    ptm_get_array(eng, PTM_ARR_NTRIES, nt.data());
    ptm_get_array(eng, PTM_ARR_NACCEPT, na.data());
    ptm_get_array(eng, PTM_ARR_LAST_TYPE, ty.data());
    const size_t c = (size_t)ichain * W + replica;
    os << nstep << " " << lpost[c] << " " << llike[c] << " " << (double)na[c] / nt[c] << " " << ty[c] << ": ";
    for (int j = 0; j < dim; j++) os << X[c * dim + j] << " ";
    os << cur_beta(ichain, replica) << std::endl;
  }
};

// ---- bayesian.hh:795-821 / ptmcmc.hh: the driver ------------------------------------------------------------------------------
class bayes_sampler : public Optioned {   // bayesian.hh:795-821
 protected:
  std::string paramfile;
  bool have_paramfile;
  void processOptions() {
    *optValue("stateFile") >> paramfile;
    if (!(paramfile == "")) have_paramfile = true;
  }

 public:
  bayes_sampler() : have_paramfile(false) {}
  virtual bayes_sampler* clone() = 0;
  virtual int initialize() = 0;
  virtual int run(const std::string& base, int ic = 0) = 0;
  virtual int analyze(const std::string& base, int ic, int Nsigma, int Nbest, bayes_likelihood& like) = 0;
  virtual bool haveParfile() { return have_paramfile; }
  virtual state getState() = 0;
  void addOptions(Options& opt, const std::string& prefix = "") override {
    Optioned::addOptions(opt, prefix);
    addOption("stateFile", "File with initialization state parameters", "");
  }
  virtual void setup(bayes_likelihood& llike, const sampleable_probability_function& prior, int output_precision = 15) = 0;
};

// ptmcmc_sampler (ptmcmc.hh:12-97, ptmcmc.cc): same calls in the same order as a reference main()
// (exampleLISA.cc:699-822):
//     ptmcmc_sampler::Init(argc, argv);  Options opt;  ptmcmc_sampler mcmc;  bayes_sampler* s0 = &mcmc;
//     s0->addOptions(opt);  like->addOptions(opt);  opt.add(Option(...));  opt.parse(argc, argv);
//     ProbabilityDist::setSeed(seed);  mcmc.setup(*like, precision);  mcmc.select_proposal();
//     for (ic...) { bayes_sampler* s = s0->clone();  s->initialize();  s->run(base, ic);  delete s; }
// select_proposal() builds the reference's default recipe (ptmcmc.cc:15-183): differential evolution + six diagonal Gaussians
// a factor gauss_step_fac apart with doubling shares (+ prior draws, + the likelihood's own proposals).  With
// --gauss_draw_frac=1 the recipe is all Gaussian and runs fused on the device; with its default (0.2) the differential-
// evolution part makes it a host-side proposal (the engine's host-proposal step).
class ptmcmc_sampler : public bayes_sampler {
  Options own_opt;   // used when the program never called addOptions (the convenience calls set() / parse() below)
  chain* cc_base;
  std::unique_ptr<parallel_tempering_chains> cc;
  proposal_distribution* cprop;
  bool have_cc, have_cprop, own_cprop;
  bayes_likelihood* chain_llike;
  const sampleable_probability_function* chain_prior;
  bool have_setup;
  int chain_Nstep, chain_Ninit, chain_nburn, output_precision;
  double swap_rate, pt_evolve_rate, pt_evolve_lpost_cut, Tmax;
  int Nstep, Nskip, Nptc, Nevery, save_every, dump_n;
  double nburn_frac;
  bool parallel_tempering;
  int istep;
  bool restarting;
  std::string restart_dir;
  int checkp_at_step;
  double ess_stop, prop_adapt_rate, dpriormin;
  int nreplicas, replica_begin = 0, device = -1;

  // every flag read into a typed value (a flag without a value reads as T())
  template <class T> T flag(const std::string& name) { T v = T(); *optValue(name) >> v; return v; }
  void ensure_options() { if (!haveOptions()) addOptions(own_opt); }
  void processOptions() {   // the run-shaping flags (the reference reads them in ptmcmc_sampler::processOptions, ptmcmc.cc:430-473)
    ensure_options();
    bayes_sampler::processOptions();
    checkp_at_step = flag<int>("checkp_at_step");
    restart_dir = flag<std::string>("restart_dir");
    restarting = !restart_dir.empty();
    Nevery = flag<int>("nevery");
    save_every = flag<int>("save_every");
    Nstep = flag<int>("nsteps");
    Nskip = flag<int>("nskip");
    nburn_frac = flag<double>("burn_frac");
    Nptc = flag<int>("pt");
    parallel_tempering = Nptc > 1;
    pt_evolve_rate = flag<double>("pt_evolve_rate");
    pt_evolve_lpost_cut = flag<double>("pt_evolve_lpost_cut");
    swap_rate = flag<double>("pt_swap_rate");
    Tmax = flag<double>("pt_Tmax");
    dump_n = flag<int>("pt_dump_n");
    if (dump_n < 0 || dump_n > Nptc) dump_n = Nptc;   // "0 for all" is resolved where the files are opened
    if (Nptc == 0) dump_n = 1;
    prop_adapt_rate = flag<double>("prop_adapt_rate");
    ess_stop = flag<double>("chain_ess_stop");
    dpriormin = flag<double>("chain_dprior_min");
    nreplicas = optSet("replicas") ? flag<int>("replicas") : 1;
    replica_begin = optSet("replica_begin") ? flag<int>("replica_begin") : 0;
    device = optSet("device") ? flag<int>("device") : -1;
  }

 public:
  ptmcmc_sampler() : cc_base(nullptr), cprop(nullptr), have_cc(false), have_cprop(false), own_cprop(false), chain_llike(nullptr), chain_prior(nullptr),
                     have_setup(false), chain_Nstep(0), chain_Ninit(1), chain_nburn(0), output_precision(13), swap_rate(0.1), pt_evolve_rate(0.01),
                     pt_evolve_lpost_cut(-1), Tmax(1e9), Nstep(5000), Nskip(10), Nptc(0), Nevery(5000), save_every(1), dump_n(1), nburn_frac(0.5),
                     parallel_tempering(false), istep(0), restarting(false), checkp_at_step(-1), ess_stop(-1), prop_adapt_rate(0), dpriormin(-30), nreplicas(1) {}
  ~ptmcmc_sampler() { if (own_cprop) delete cprop; }
  // MPI start-up / shut-down in the reference (ptmcmc.cc:250-275); here the process already owns its GPU
  static void Init() { std::cout << "Running on the MI355X step engine (no MPI; ladders shard over GPUs through ptmcmc_amd.parallel)." << std::endl; }
  static void Init(int& argc, char* argv[]) { (void)argc; (void)argv; Init(); }
  static void setRNGseed(double seed) { ProbabilityDist::setSeed(seed); }
  static void Quit() { exit(0); }
  static bool static_reporting() { return true; }
  bool reporting() { return true; }

  // The sampler's flags: the NAMES and DEFAULTS are the reference's (ptmcmc.cc:375-427) -- command lines and scripts written for
  // it must parse and mean the same -- the descriptions are this build's.  "" as default: the flag has no value until given.
  // Flags of features outside this build's scope are declared too, so that such command lines parse; setup() refuses the ones
  // that would change the run (refuse_unbuilt).
  void addOptions(Options& opt, const std::string& prefix = "") override {
    bayes_sampler::addOptions(opt, prefix);
    struct decl { const char *name, *preset, *about; };
    static const char* const no_value = "<no default>";
    static const decl flags[] = {
        // run length and output
        {"nsteps", "5000", "Steps to run each chain for. [5000]"},
        {"save_every", "1", "Keep every n-th state in the chain history. [1]"},
        {"nevery", "5000", "Report and write chain-file rows every n steps. [5000]"},
        {"nskip", "10", "Of the saved states, write only every n-th to the chain files. [10]"},
        {"burn_frac", "0.5", "Leading fraction of the run treated as burn-in by the summaries. [0.5]"},
        // checkpointing
        {"checkp_at_step", "-1", "Write a checkpoint at this step and stop. [-1: never]"},
        {"checkp_at_time", "-1", "Write a checkpoint after this many hours of wall time and stop (not built here). [-1: never]"},
        {"checkp_on_sigterm_within", "0", "Look for a termination signal every n steps and checkpoint on it (not built here). [0: off]"},
        {"restart_dir", "", "Continue from the checkpoint in this directory."},
        // the temperature ladder
        {"pt", "0", "Number of temperatures of the parallel-tempering ladder. [0: no tempering]"},
        {"pt_Tmax", "1e9", "Temperature of the hottest rung. [1e9]"},
        {"pt_swap_rate", "0.10", "Exchange attempts per rung pair and step. [0.10]"},
        {"pt_evolve_rate", "0.01", "Every accepted exchange widens its temperature gap by this fraction (0: fixed ladder). [0.01]"},
        {"pt_evolve_lpost_cut", "-1", "While the ladder evolves, also widen gaps whose log-posteriors are out of order by more than this. [-1: off]"},
        {"pt_dump_n", "1", "Write chain files for the n coldest rungs (0: all of them). [1]"},
        {"pt_reboot_rate", "0", "Highest rate of restarting lagging hot chains (not built here). [0]"},
        {"pt_reboot_every", "0", "Steps between tests for lagging hot chains (not built here). [0]"},
        {"pt_reboot_grace", "0", "Steps a restarted chain is left alone (not built here). [0]"},
        {"pt_reboot_cut", "100", "Log-posterior deficit that marks a lagging chain (not built here). [100]"},
        {"pt_reboot_thermal", "0", "Temperature-dependent part of that deficit (not built here). [0]"},
        {"pt_reboot_blindly", "0", "Restart at random at this level even without a deficit (not built here). [0]"},
        {"pt_reboot_grad", no_value, "Make the grace period grow towards the cold rungs, with this mean (not built here)."},
        {"pt_stop_evid_err", "0", "Stop once the evidence estimate is consistent to this error (not built here). [0: off]"},
        // the default proposal recipe
        {"prop", "", "No longer used."},
        {"gauss_draw_frac", "0.20", "Share of Gaussian steps in the default proposal mixture. [0.20]"},
        {"gauss_1d_frac", "0.5", "Fraction of the Gaussian steps that move one randomly chosen parameter only. [0.5]"},
        {"gauss_step_fac", "2", "Ratio between the widths of successive Gaussian members. [2]"},
        {"gauss_temp_scaled", no_value, "Scale the Gaussian widths with the chain temperature."},
        {"prior_draw_frac", "0", "Share of independent draws from the prior in the mixture. [0]"},
        {"prior_draw_Tpow", "0", "Let the prior-draw share grow towards the hot rungs as 1 - T^-p. [0: same at every rung]"},
        {"cov_draw_frac", "0.50", "Share of Gaussian steps with the covariance of --covariance_file, if one is given. [0.50]"},
        {"covariance_file", "", "File holding a proposal covariance."},
        {"prop_adapt_rate", "0", "Rate (e.g. 1e-3) at which the Gaussian members' shares adapt to their acceptance. [0: fixed]"},
        {"prop_adapt_more", no_value, "Adapt the shares of the whole mixture, not of the Gaussian members alone."},
        {"sym_prop_frac", "0", "Share of state-space symmetry moves (not built here). [0]"},
        {"like_prop_frac", "0", "Share of the proposals the likelihood object brings along. [0]"},
        {"de_ni", "50", "Differential evolution: initial history rows per parameter. [50]"},
        {"de_eps", "1e-4", "Differential evolution: width of the small random jump. [1e-4]"},
        {"de_reduce_gamma", "4", "Differential evolution: divide the nominal jump scale by this. [4]"},
        {"de_g1_frac", "0.3", "Differential evolution: fraction of jumps made with unit scale. [0.3]"},
        {"de_mixing", no_value, "Differential evolution: draw history states from every rung of the ladder."},
        {"de_Tmix", "300", "Differential evolution: how readily other rungs' histories are used. [300]"},
        {"de_unlikely_alpha", "0", "Differential evolution: power with which improbable history states are passed over. [0: never]"},
        {"prop_test_index", "", "Index of a proposal to test in isolation (not built here)."},
        // chains
        {"chain_init_file", "", "Take the start states from this chain file instead of the prior (not built here)."},
        {"chain_ess_stop", "-1.0", "Stop as soon as the cold chain's effective sample size exceeds this. [-1: never]"},
        {"chain_ess_limit", "-1.0", "Largest effective sample size worth resolving; speeds the estimate up on long chains. [-1: none]"},
        {"chain_dprior_min", "-30", "Reject, without asking the likelihood, a proposal whose log-prior drops by more than this. [-30]"},
        // this build's additions.  Replicas: independent copies of the ladder side by side in one engine (the reference runs
        // its repeats one after the other, the caller's loop over clone() / initialize() / run()); over several GPUs one
        // process per GPU, each told which replicas of the population are its own
        {"replicas", "1", "Independent replicas of the ladder run side by side on the device (files <base>_c<w>_t<k>.dat). [1]"},
        {"replica_begin", "0", "Global index of this process's first replica (one process per GPU, disjoint ranges). [0]"},
        {"device", "-1", "GPU of this process. [-1: the current device, e.g. by HIP_VISIBLE_DEVICES]"},
    };
    for (size_t i = 0; i < sizeof flags / sizeof flags[0]; i++) addOption(flags[i].name, flags[i].about, flags[i].preset);
  }

  // ---- convenience for programs without an Options object of their own (examples/example_sampler.cc)
  void set(const std::string& name, const std::string& value) {
    ensure_options();
    std::string arg = "--" + name + "=" + value;
    std::vector<char> buf(arg.begin(), arg.end());
    buf.push_back(0);
    char* argv[2] = {nullptr, buf.data()};
    int argc = 2;
    if (own_opt.parse(argc, argv, false)) { std::cout << "ptmcmc_sampler::set: unknown option '" << name << "'" << std::endl; exit(1); }
  }
  bool parse(int argc, char* argv[]) {  // --name=value / --name; true when every flag was understood
    ensure_options();
    if (!own_opt.exists("seed")) {
      own_opt.add(Option("seed", "Seed of the random streams, a number in [0,1). [-1: the engine's fixed default key]", "-1"));
      own_opt.add(Option("outname", "Stem of the output file names. [mcmc_output]", "mcmc_output"));
      own_opt.add(Option("nchains", "Replicas of the ladder run side by side (alias of --replicas).", "1"));
    }
    int n = argc;
    const bool bad = own_opt.parse(n, argv, true);
    double seed = -1;
    std::istringstream(own_opt.value("seed")) >> seed;
    if (seed >= 0) ProbabilityDist::setSeed(seed);
    int nch = 1;
    std::istringstream(own_opt.value("nchains")) >> nch;
    if (nch > 1) set("replicas", own_opt.value("nchains"));
    return !bad;
  }
  double num(const std::string& n) { double v = 0; *optValue(n) >> v; return v; }

  // ptmcmc.cc:476-506
  void setup(int Ninit, bayes_likelihood& llike, const sampleable_probability_function& prior, proposal_distribution& prop, int output_precision_ = 15) {
    processOptions();
    chain_Nstep = Nstep; chain_Ninit = Ninit; chain_nburn = (int)(Nstep * nburn_frac); output_precision = output_precision_;
    cprop = &prop; own_cprop = false; chain_prior = &prior; chain_llike = &llike; have_setup = true; have_cprop = true;
  }
  void setup(bayes_likelihood& llike, const sampleable_probability_function& prior, int output_precision_ = 15) override {
    processOptions();
    chain_Nstep = Nstep; chain_nburn = (int)(Nstep * nburn_frac); output_precision = output_precision_;
    chain_prior = &prior; chain_llike = &llike; have_setup = true;
    refuse_unbuilt();
  }
  void setup(bayes_likelihood& llike, int output_precision_ = 15) { setup(llike, *llike.getObjectPrior(), output_precision_); }
  void refuse_unbuilt() {
    double v = 0; std::string sv;
    *optValue("pt_reboot_rate") >> v;
    if (v > 0) { std::cout << "ptmcmc_sampler: pt_reboot_rate > 0 (rebooting laggard chains) is not built in the GPU step engine." << std::endl; exit(1); }
    *optValue("pt_stop_evid_err") >> v;
    if (v > 0) { std::cout << "ptmcmc_sampler: pt_stop_evid_err > 0 (evidence integration) is not built in the GPU step engine." << std::endl; exit(1); }
    *optValue("chain_init_file") >> sv;
    if (!sv.empty()) { std::cout << "ptmcmc_sampler: chain_init_file is not built in the GPU step engine." << std::endl; exit(1); }
    if (Nptc < 2) { std::cout << "ptmcmc_sampler: this build drives parallel-tempering ladders: set --pt=N with N >= 2." << std::endl; exit(1); }
  }
  // a proposal of the caller's (the convenience path of examples/example_sampler.cc; the reference's deprecated setup(Ninit, ...))
  void select_proposal(proposal_distribution& p) { if (own_cprop) delete cprop; cprop = &p; own_cprop = false; have_cprop = true; }

  // ---- the default proposal (what ptmcmc_sampler::select_proposal() of the reference builds from the flags, ptmcmc.cc:15-183) -----
  // A mixture of, in this order:
  //   differential evolution   share 1 - gauss - cov - prior (not below 0): differential_evolution(snooker 0.1, de_g1_frac,
  //                            de_eps, ignore_frac 0, de_unlikely_alpha), reduce_gamma(de_reduce_gamma), optional rung mixing;
  //                            its history is seeded with de_ni initial draws per parameter
  //   prior draws              share prior_draw_frac (hot share 1, power prior_draw_Tpow), when asked for
  //   six Gaussian steps       share gauss_draw_frac split 2 : 4 : ... : 64; widths scale_i / 100 / f with
  //                            f = (2/s)^4 s, (2/s)^4 s^2, ..., s = gauss_step_fac (2: f = 2, 4, ..., 64); each moves a single
  //                            random parameter with probability gauss_1d_frac.  With prop_adapt_rate > 0 the six form a set
  //                            of their own whose inner shares adapt.
  // and, around that, the likelihood's own proposals with share like_prop_frac.
  struct recipe {
    double gauss, gauss_1d, gauss_ratio, prior, prior_power, like_share, de_unit_frac, de_noise, de_divisor, de_mix, de_alpha;
    int de_rows_per_dim;
    bool de_mixing, adapt_all, gauss_temp_scaled;
  };
  static double unit_interval(double v) { return v < 0 ? 0.0 : (v > 1 ? 1.0 : v); }
  recipe read_recipe() {
    recipe r;
    r.prior = unit_interval(flag<double>("prior_draw_frac"));
    r.prior_power = flag<double>("prior_draw_Tpow");
    r.gauss_1d = unit_interval(flag<double>("gauss_1d_frac"));
    r.gauss = unit_interval(flag<double>("gauss_draw_frac"));
    r.gauss_ratio = std::max(1.0, flag<double>("gauss_step_fac"));
    r.like_share = flag<double>("like_prop_frac");
    r.de_rows_per_dim = flag<int>("de_ni");
    r.de_noise = flag<double>("de_eps");
    r.de_divisor = flag<double>("de_reduce_gamma");
    r.de_unit_frac = flag<double>("de_g1_frac");
    r.de_mix = flag<double>("de_Tmix");
    r.de_alpha = flag<double>("de_unlikely_alpha");
    r.de_mixing = optSet("de_mixing");
    r.adapt_all = optSet("prop_adapt_more");
    r.gauss_temp_scaled = optSet("gauss_temp_scaled");   // (without effect in the reference as well: quirk Q4)
    // a covariance file is read by a function that is empty in the reference (ptmcmc.cc:761-764): its share is always 0 here
    if (!flag<std::string>("covariance_file").empty())
      std::cout << "ptmcmc_sampler::select_proposal: --covariance_file is not read (nor is it by the reference, whose reader is an empty stub); no covariance steps." << std::endl;
    if (r.prior + r.gauss > 1) r.gauss = 1.0;   // over-subscribed: the Gaussian share is renormalised against itself
    if (flag<double>("sym_prop_frac") > 0)
      std::cout << "ptmcmc_sampler::select_proposal: state-space symmetry moves (--sym_prop_frac) are outside this build's scope; none are added." << std::endl;
    return r;
  }
  // the six Gaussian members and their relative weights 2, 4, ..., 64 (sum 126)
  void gaussian_members(const recipe& r, const std::vector<double>& scales, std::vector<proposal_distribution*>& members, std::vector<double>& weights) {
    const int count = 6;
    const double total = std::pow(2.0, count + 1) - 2;
    double divisor = std::pow(2.0 / r.gauss_ratio, 4.0), weight = 1;
    for (int k = 0; k < count; k++) {
      divisor *= r.gauss_ratio;
      weight *= 2;
      std::vector<double> sigma(scales.size());
      for (size_t i = 0; i < sigma.size(); i++) sigma[i] = scales[i] / 100.0 / divisor;
      members.push_back(new gaussian_prop(sigma, r.gauss_1d, r.gauss_temp_scaled));
      weights.push_back(weight / total);
    }
  }
  void select_proposal() {
    if (!have_setup) { std::cout << "ptmcmc_sampler::select_proposal.  Must call setup() first!" << std::endl; exit(1); }
    const recipe r = read_recipe();
    std::vector<double> scales;
    chain_llike->getScales(scales);
    const int npar = chain_prior->get_space() ? chain_prior->get_space()->size() : chain_prior->getDim();
    std::vector<proposal_distribution*> members;
    std::vector<double> cold, hot;
    double thermal_power = 0;
    // A differential-evolution member with share 0 is never picked (no uniform lies below a threshold of 0) and draws nothing:
    // leaving it out gives the same chain, and an all-Gaussian recipe can then run on the device.
    const double de_share = std::max(0.0, 1 - r.gauss - r.prior);
    if (de_share > 0) {
      differential_evolution* de = new differential_evolution(0.1, r.de_unit_frac, r.de_noise, 0.0, r.de_alpha);
      de->reduce_gamma(r.de_divisor);
      if (r.de_mixing) de->support_mixing(true);
      de->mix_temperatures_more(r.de_mix);
      members.push_back(de); cold.push_back(de_share); hot.push_back(0.0);
      chain_Ninit = r.de_rows_per_dim * npar;
    } else chain_Ninit = 1;
    if (r.prior > 0) {
      members.push_back(new draw_from_dist(*chain_prior)); cold.push_back(r.prior); hot.push_back(1.0);
      thermal_power = r.prior_power;
    }
    std::vector<proposal_distribution*> gaussians;
    std::vector<double> gweights;
    gaussian_members(r, scales, gaussians, gweights);
    if (prop_adapt_rate > 0) {   // the Gaussian part as one member whose inner shares adapt
      members.push_back(new proposal_distribution_set(gaussians, gweights, prop_adapt_rate)); cold.push_back(r.gauss); hot.push_back(0.0);
    } else
      for (size_t k = 0; k < gaussians.size(); k++) { members.push_back(gaussians[k]); cold.push_back(gweights[k] * r.gauss); hot.push_back(0.0); }
    if (own_cprop) delete cprop;
    const double outer_rate = r.adapt_all ? prop_adapt_rate : 0.0;
    cprop = new proposal_distribution_set(members, cold, outer_rate, thermal_power, hot);
    own_cprop = true;
    std::cout << "ptmcmc_sampler::set_proposal: Tpow=" << thermal_power << ":" << std::endl;
    if (r.like_share > 0 && chain_llike->get_proposals().size() > 0) {
      std::cout << "Adding likelihood-based elements to proposal." << std::endl;
      std::vector<proposal_distribution*> theirs;
      for (auto q : chain_llike->get_proposals()) theirs.push_back(q->clone());   // (the likelihood keeps its own objects)
      std::vector<proposal_distribution*> both;
      both.push_back(cprop);
      both.push_back(new proposal_distribution_set(theirs, chain_llike->get_prop_shares(), outer_rate));
      std::vector<double> split;
      split.push_back(std::max(0.0, 1 - r.like_share));
      split.push_back(r.like_share);
      cprop = new proposal_distribution_set(both, split, prop_adapt_rate);
    }
    std::cout << "Proposal distribution is:\n" << cprop->show() << std::endl;
    have_cprop = true;
  }

  bayes_sampler* clone() override {   // ptmcmc.hh:57-76
    if (have_cc) { std::cout << "ptmcmc_sampler::clone(): Cannot clone after instantiating chain/prop." << std::endl; exit(1); }
    ptmcmc_sampler* s = new ptmcmc_sampler();
    s->copyOptioned(*this);
    if (!haveOptions()) { s->own_opt = own_opt; s->Optioned::addOptions(s->own_opt); }
    if (have_setup) s->setup(*chain_llike, *chain_prior, output_precision);
    if (have_cprop) {
      s->cprop = cprop->clone();
      s->own_cprop = true;
      s->have_cprop = true;
      s->chain_Ninit = chain_Ninit;
    }
    return s;
  }
  ptmcmc_sampler* clone_ptmcmc_sampler() { return dynamic_cast<ptmcmc_sampler*>(clone()); }
  state getState() override {
    if (!have_setup) { std::cout << "ptmcmc_sampler::getState.  Must call setup() before getState!" << std::endl; exit(1); }
    if (have_cc) return cc->getState();
    philox_random r;
    r.reseat(ProbabilityDist::engineSeed(), 0xFFFFFFFFu, 0);
    return chain_prior->drawSample(r);
  }

  // ptmcmc_sampler::checkpoint / restart (ptmcmc.cc:306-338): <path>/step_<istep>-cp/ptmcmc.cp + the ladder's own file
  void checkpoint(const std::string& path, int istep_) {
    std::ostringstream ss;
    ss << path << "/step_" << istep_ << "-cp/";
    const std::string dir = ss.str();
    std::cout << "Writing checkpoint files to dir:" << dir << std::endl;
    mkdir(dir.c_str(), 0777);
    std::ofstream os((dir + "ptmcmc.cp").c_str());
    os << istep_ << std::endl;
    cc->checkpoint(dir);
  }
  int restart(const std::string& path) {
    std::cout << "Restarting from checkpoint files in dir:" << path << std::endl;
    std::ifstream is((path + "/ptmcmc.cp").c_str());
    int istep_ = -1;
    is >> istep_;
    if (!is || istep_ < 0) { std::cout << "ptmcmc_sampler::restart: cannot read " << path << "/ptmcmc.cp" << std::endl; exit(1); }
    cc->restart(path + "/");
    return istep_;
  }

  int initialize() override {   // ptmcmc.cc:497-528
    if (!have_setup || !have_cprop) { std::cout << "ptmcmc_sampler::initialize.  Must call setup() and set proposal before initialization!" << std::endl; exit(1); }
    if (!parallel_tempering) { std::cout << "ptmcmc_sampler::initialize: this build drives parallel-tempering ladders: set --pt=N with N >= 2." << std::endl; exit(1); }
    int Ninit = chain_Ninit;
    if (restarting || Nstep <= 0) Ninit = 1;
    cc.reset(new parallel_tempering_chains(Nptc, Tmax, swap_rate, save_every, false, false, dpriormin));
    cc_base = cc.get();
    have_cc = true;
    // the chain files are written from the device's history ring, every "nevery" steps: it must hold what one such
    // interval saves (up to two add_state calls per step, every save_every-th saved)
    int dn = dump_n;
    if (dn > Nptc || dn <= 0) dn = Nptc;   // ptmcmc.cc:458
    // ... and with an effective-sample-size stop (chain_ess_stop) the cold chain's whole saved history
    int kind; double odf; std::vector<double> f;
    const int dim = chain_prior->getDim();
    bool host = !cprop->device_describe(dim, kind, f, odf) && !dynamic_cast<user_gaussian_prop*>(cprop);
    // A set with a differential evolution the device can draw (the default recipe): the device keeps EVERY rung's saved history of
    // the whole run -- up to two add_state calls per step, every save_every-th saved -- if that fits (16 GB here); else the host draws
    ptm_de_params deq;
    const long long de_rows = 2 + 2ll * std::max(Nevery, Nstep) / std::max(1, save_every);
    const double de_bytes = (double)de_rows * Nptc * std::max(1, nreplicas) * (8.0 * std::max(4, dim > 16 ? 32 : (dim > 8 ? 16 : (dim > 4 ? 8 : 4))) + 40.0);
    const bool de_dev = !host && cprop->device_describe_de(deq) && dim <= 128 && de_bytes < 16e9 && de_rows < (1ll << 30);
    if (!host && cprop->device_describe_de(deq) && !de_dev) host = true;
    if (de_dev) {
      if (restarting && Nstep > 0) Ninit = chain_Ninit;   // (the device's initial rows are drawn again, the same ones: they are not in the checkpoint)
      cc->keep_history((int)de_rows, Nptc);
      cc->use_device_de(true);
    } else {
      cc->keep_history(2 + 2 * (ess_stop > 0 ? std::max(Nevery, Nstep) : Nevery) / std::max(1, save_every), dn);
    }
    cc->set_replica_range(replica_begin, nreplicas, device);
    if (pt_evolve_rate > 0) cc->evolve_temps(pt_evolve_rate, pt_evolve_lpost_cut);   // ptmcmc.cc:512
    cc->use_host_proposals(host);
    cc->initialize(chain_llike, chain_prior, (host || de_dev) ? Ninit : 1, ProbabilityDist::nextLadderSeed());
    cc->set_proposal(*cprop);
    return 0;
  }

  // ptmcmc_sampler::run (ptmcmc.cc:530-679): step; every "nevery" steps append what the coldest pt_dump_n chains saved
  // since the last report to <base>_t<k>.dat (k = 0 the coldest), as dumpChain writes it
  int run(const std::string& base, int ic = 0) override {
    if (!have_cc && chain_Nstep > 0) { std::cout << "ptmcmc_sampler::run.  Must call initialize() before running!" << std::endl; exit(1); }
    if (ic > 0 && restarting) { std::cout << "ptmcmc_sampler::run: Can't restart except for single chain ic=0." << std::endl; exit(1); }
    int dn = dump_n;
    if (dn > cc->multiplicity() || dn <= 0) dn = cc->multiplicity();   // ptmcmc.cc:458
    cc->limit_history_reads(dn);   // (this loop dumps rungs 0 .. dn - 1 and estimates on the cold one: no other rung's ring is read back)
    const int nrep = cc->replicas(), every = std::max(1, Nevery), skip = std::max(1, Nskip);
    std::ios_base::openmode mode = std::ios::out;
    if (ic > 0 || restarting) mode = mode | std::ios::app;   // ptmcmc.cc:538
    std::vector<std::unique_ptr<std::ofstream> > out;
    for (int w = 0; w < nrep; w++)
      for (int ich = 0; ich < dn; ich++) {   // ptmcmc.cc:547-554; replica w > 0: <base>_c<w>_t<ich>.dat
        std::ostringstream ss;
        ss << base;
        if (w + cc->replica_begin() > 0) ss << "_c" << (w + cc->replica_begin());   // (the GLOBAL replica index)
        ss << "_t" << ich << ".dat";
        out.emplace_back(new std::ofstream(ss.str().c_str(), mode));
        out.back()->precision(output_precision);
      }
    std::cout << "\nRunning chain " << ic << " for up to " << chain_Nstep << " steps." << std::endl;
    chain_llike->reset();   // ptmcmc.cc:558
    int istep0 = 0;
    if (restarting) istep0 = restart(restart_dir);   // ptmcmc.cc:564
    for (istep = istep0; istep <= chain_Nstep; istep++) {   // ptmcmc.cc:565,599-607
      if (istep == checkp_at_step) {   // ptmcmc.cc:567,593-596: write the checkpoint and stop
        std::cout << "Checkpointing triggered." << std::endl;
        checkpoint(".", istep);
        return 0;
      }
      // up to the next step somebody looks at the chains: a report / dump (every `every` steps), the checkpoint, the end
      int last = istep + (every - istep % every) % every;
      if (last > chain_Nstep) last = chain_Nstep;
      if (checkp_at_step > istep && checkp_at_step <= last) last = checkp_at_step - 1;
      cc->step_n(last - istep + 1);
      istep = last;
      bool stop = false;
      if (0 == istep % every) {
        std::cout << "chain " << ic << " step " << istep << std::endl;
        std::cout << "   MaxPosterior=" << chain_llike->bestPost() << std::endl;
        for (int w = 0; w < nrep; w++)
          for (int ich = 0; ich < dn; ich++) cc->dumpChain(ich, *out[(size_t)w * dn + ich], istep - every + 1, skip, w);
        if (0 == istep % (every * 4)) {   // ptmcmc.cc:620-651
          std::cout << "Proposal report:\n" << cc->report_prop(1) << "\nacceptance report:\n" << cc->report_prop(0) << std::endl;
          if (ess_stop > 0) {   // ptmcmc.cc:628-649
            double esslimit = -1;
            *optValue("chain_ess_limit") >> esslimit;
            std::cout << "Effective sample size test" << std::endl;
            const std::pair<double, int> ess_len = cc->report_effective_samples(-1, save_every * 1000, save_every, esslimit);
            if (ess_len.first > ess_stop) {
              stop = true;
              std::cout << "ptmcmc_sampler::run: Stopping based on chain_ess_stop Effective Sample Size criterion." << std::endl;
            }
          }
        }
      }
      if (stop) break;
    }
    for (size_t k = 0; k < out.size(); k++) *out[k] << "\n" << std::endl;   // ptmcmc.cc:665
    std::cout << "Finished running chain " << ic << "." << std::endl;
    return 0;
  }
  // ptmcmc_sampler::analyze (ptmcmc.cc:681-755): the 1-sigma sample files of the reference need bayes_likelihood::write /
  // writeFine (signal / data modelling, out of this build's scope); what remains is its summary line
  int analyze(const std::string& base, int ic, int Nsigma, int Nbest, bayes_likelihood& like) override {
    (void)base; (void)Nsigma; (void)Nbest; (void)like;
    if (!have_cc) { std::cout << "ptmcmc_sampler::analyze.  Must call initialize() before analyze()!" << std::endl; exit(1); }
    std::cout << "chain " << ic << ": best_post " << chain_llike->bestPost() << ", state=" << chain_llike->bestState().get_string() << std::endl;
    return 0;
  }
  parallel_tempering_chains* chains() { return cc.get(); }
};

}  // namespace ptmgpu
#endif
