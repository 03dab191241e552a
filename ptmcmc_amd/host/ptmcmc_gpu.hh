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

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <valarray>
#include <vector>
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

class state {  // states.hh:147-234: a host value (parameters + space); validity is decided on the device
  const stateSpace* space;
  std::valarray<double> params;
  bool valid;

 public:
  state(const stateSpace* space = nullptr, int n = 0) : space(space), params(0.0, n), valid(space != nullptr) {}
  state(const stateSpace* sp, const std::valarray<double>& a) : space(sp), params(a), valid(sp != nullptr) {}
  state(const stateSpace* sp, const std::vector<double>& a) : space(sp), params(a.data(), a.size()), valid(sp != nullptr) {}
  int size() const { return params.size(); }
  double get_param(int i) const { return params[i]; }
  double get_param(const std::string& name) const { return params[space->requireIndex(name)]; }
  void set_param(int i, double v) { params[i] = v; }
  std::valarray<double> get_params() const { return params; }
  std::vector<double> get_params_vector() const { return std::vector<double>(std::begin(params), std::end(params)); }
  const stateSpace* getSpace() const { return space; }
  bool invalid() const { return !valid; }
  std::string get_string(int prec = -1) const {
    std::ostringstream s;
    if (prec > 0) s.precision(prec);
    for (int i = 0; i < size(); i++) s << (i ? ", " : "") << params[i];
    return s.str();
  }
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
  // engine description: per-dimension (type, center, halfwidth), types as mixed_dist_product::{uniform,...}
  virtual void describe(std::vector<int>& types, std::vector<double>& centers, std::vector<double>& halfwidths) const = 0;
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
  void describe(std::vector<int>& t, std::vector<double>& c, std::vector<double>& h) const override {
    t.assign(std::begin(types), std::end(types));
    c.assign(std::begin(centers), std::end(centers));
    h.assign(std::begin(halfwidths), std::end(halfwidths));
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

// A small persistent worker pool for the likelihood batches: starting and joining threads for every batch costs more than
// a cheap plug-in's whole batch.  run(n, chunk, f) calls f(k0, k1) over [0, n) in chunks, on the workers and the caller.
class eval_pool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  std::function<void(int, int)> job;
  std::atomic<int> next{0};
  int n = 0, chunk = 1, generation = 0, busy = 0;
  bool stop = false;
  void drain() {
    for (;;) {
      const int k0 = next.fetch_add(chunk);
      if (k0 >= n) return;
      job(k0, k0 + chunk < n ? k0 + chunk : n);
    }
  }
  void loop() {
    int seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_go.wait(lk, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
      }
      drain();
      std::lock_guard<std::mutex> lk(m);
      if (--busy == 0) cv_done.notify_one();
    }
  }

 public:
  ~eval_pool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_go.notify_all();
    for (auto& t : workers) t.join();
  }
  int size() const { return (int)workers.size(); }
  void resize(int nworkers) {
    while ((int)workers.size() < nworkers) workers.emplace_back([this] { loop(); });
  }
  void run(int n_, int chunk_, const std::function<void(int, int)>& f) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = f; n = n_; chunk = chunk_ < 1 ? 1 : chunk_; next = 0; busy = (int)workers.size(); ++generation;
    }
    cv_go.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return busy == 0; });
  }
};

// ---- bayesian.hh: the likelihood plug-in -----------------------------------------------------------------------------
class bayes_likelihood : public probability_function {  // bayesian.hh:307-581 (minimal interface)
 protected:
  stateSpace nativeSpace;
  std::shared_ptr<const sampleable_probability_function> nativePrior;
  double (*user_evaluate_log)(void* object, const state& s);
  void* user_object;
  bool evaluate_log_registered;
  double best_post;

 public:
  bool check_posterior;
  bayes_likelihood() : probability_function(nullptr), user_evaluate_log(nullptr), user_object(nullptr),
                       evaluate_log_registered(false), best_post(-INFINITY), check_posterior(true) {}
  void basic_setup(const stateSpace* sp, sampleable_probability_function* prior) {  // bayesian.hh:345-358
    nativeSpace = *sp;
    nativePrior.reset(prior);
    space = &nativeSpace;
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
  // host evaluation of the plug-in (bayesian.hh:553-581; the prior part of the posterior check happens on the device)
  double evaluate_log(state& s) override {
    if (!evaluate_log_registered) { std::cout << "bayes_component::panic!\nNo evaluate_log function is registered" << std::endl; exit(1); }
    double result = (*user_evaluate_log)(user_object, s);
    if (check_posterior && !std::isfinite(result) && !(result < 0)) result = -INFINITY;  // NaN/+inf -> -inf (bayesian.hh:569-575)
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
        state s(l->getObjectStateSpace(), std::valarray<double>(X + (size_t)k * dim, dim));
        out[k] = l->evaluate_log(s);
      }
    };
    int nt = l->eval_threads > 0 ? l->eval_threads : (int)std::thread::hardware_concurrency();
    if (nt > n / 8) nt = n / 8;      // at least 8 states per thread
    // starting and joining threads costs ~50 us: a batch that the measured cost per evaluation prices below ~4 such
    // units stays on this thread (eval_threads == 0 only; an explicit thread count is obeyed)
    if (l->eval_threads <= 0 && l->eval_ns >= 0 && l->eval_ns * n < 200e3) nt = 1;
    if (nt <= 1) {
      const auto t0 = std::chrono::steady_clock::now();
      work(0, n);
      const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
      if (n > 0) l->eval_ns = l->eval_ns < 0 ? ns / n : 0.8 * l->eval_ns + 0.2 * ns / n;
      return;
    }
    if (!l->pool) l->pool.reset(new eval_pool);
    l->pool->resize(nt - 1);   // the calling thread works too
    l->pool->run(n, (n + 4 * nt - 1) / (4 * nt), work);
  }

 private:
  int eval_threads = 0;   // 0: all hardware threads
  double eval_ns = 0;     // running estimate of one evaluation's cost (the first batch runs serially and measures it)
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

// ---- proposal_distribution.hh ---------------------------------------------------------------------------------------
class proposal_distribution {  // proposal_distribution.hh:38-88 (what the device path needs of it)
 public:
  virtual ~proposal_distribution() {}
  virtual proposal_distribution* clone() const = 0;
  virtual std::string show() { return "UnspecifiedProposal()"; }
  virtual bool device_describe(int dim, int& kind, std::vector<double>& factor, double& oneDfrac) const { return false; }
  // a set of Gaussian members that are scalar multiples of one factor: cumulative shares, scales, oneDfracs (else false)
  virtual bool device_describe_mixture(int dim, std::vector<double>& cum, std::vector<double>& scales, std::vector<double>& odfs) const { return false; }
};

class gaussian_prop : public proposal_distribution {  // proposal_distribution.hh:145-227
  bool identity_trans;
  std::vector<double> factor;  // DIAG: sigmas; else dense D x D row-major V*diag(sqrt(lambda))
  int ndim;
  double oneDfrac;

  // symmetric eigen-decomposition by cyclic Jacobi (the reference uses Eigen::SelfAdjointEigenSolver, hh:173-176);
  // eigenvalues ascending, factor = V * diag(sqrt(lambda)) so that offset = factor * z as in hh:207-213
  static void eigen_factor(std::vector<double> A, int n, std::vector<double>& F) {
    std::vector<double> V(n * n, 0.0);
    for (int i = 0; i < n; i++) V[i * n + i] = 1;
    for (int sweep = 0; sweep < 100; sweep++) {
      double off = 0;
      for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
      if (off < 1e-300) break;
      for (int p = 0; p < n; p++)
        for (int q = p + 1; q < n; q++) {
          if (std::fabs(A[p * n + q]) < 1e-300) continue;
          double theta = (A[q * n + q] - A[p * n + p]) / (2 * A[p * n + q]);
          double t = (theta >= 0 ? 1 : -1) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
          double c = 1 / std::sqrt(t * t + 1), s = t * c;
          for (int k = 0; k < n; k++) { double a = A[k * n + p], b = A[k * n + q]; A[k * n + p] = c * a - s * b; A[k * n + q] = s * a + c * b; }
          for (int k = 0; k < n; k++) { double a = A[p * n + k], b = A[q * n + k]; A[p * n + k] = c * a - s * b; A[q * n + k] = s * a + c * b; }
          for (int k = 0; k < n; k++) { double a = V[k * n + p], b = V[k * n + q]; V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b; }
        }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]]) std::swap(order[i], order[j]);
    F.assign(n * n, 0.0);
    for (int j = 0; j < n; j++) {
      double lam = A[order[j] * n + order[j]];
      double sg = std::sqrt(lam > 0 ? lam : 0.0);
      for (int i = 0; i < n; i++) F[i * n + j] = V[i * n + order[j]] * sg;
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
  bool device_describe(int dim, int& kind, std::vector<double>& f, double& odf) const override {
    if (dim != ndim) { std::cout << "gaussian_prop: dimension mismatch with the chain (" << ndim << " vs " << dim << ")" << std::endl; exit(1); }
    kind = identity_trans ? PTM_PROP_DIAG : PTM_PROP_DENSE;
    f = factor;
    odf = oneDfrac;
    return true;
  }
};

// proposal_distribution_set (proposal_distribution.hh:90-143, .cc:99-129): draws a member with probability share_i.
// On the device: members must be gaussian_props that are scalar multiples of the first one (the sampler's default
// Gaussian recipe, ptmcmc.cc:117-139, is exactly that) -- the rung keeps ONE factor and a table of scales.
class proposal_distribution_set : public proposal_distribution {
  std::vector<proposal_distribution*> proposals;   // owned
  std::vector<double> shares;

 public:
  proposal_distribution_set(const std::vector<proposal_distribution*>& props, const std::vector<double>& shares_) : shares(shares_) {
    if (props.size() != shares.size() || props.empty()) { std::cout << "proposal_distribution_set: need one share per proposal" << std::endl; exit(1); }
    for (auto p : props) proposals.push_back(p->clone());
  }
  ~proposal_distribution_set() { for (auto p : proposals) delete p; }
  proposal_distribution* clone() const override { return new proposal_distribution_set(proposals, shares); }
  std::string show() override {
    std::ostringstream ss;
    ss << "ProposalSet(";
    for (size_t i = 0; i < proposals.size(); i++) ss << shares[i] << " : " << proposals[i]->show() << (i + 1 < proposals.size() ? ", " : ")");
    return ss.str();
  }
  bool device_describe(int dim, int& kind, std::vector<double>& f, double& odf) const override {
    if (!proposals[0]->device_describe(dim, kind, f, odf)) return false;
    odf = 0;   // the members' oneDfracs live in the mixture table
    return true;
  }
  bool device_describe_mixture(int dim, std::vector<double>& cum, std::vector<double>& scales, std::vector<double>& odfs) const override {
    int kind0; double odf0; std::vector<double> f0;
    if (!proposals[0]->device_describe(dim, kind0, f0, odf0)) return false;
    double sum = 0;
    for (double sh : shares) sum += sh;
    double run = 0;
    cum.clear(); scales.clear(); odfs.clear();
    for (size_t i = 0; i < proposals.size(); i++) {
      int kind; double odf; std::vector<double> f;
      if (!proposals[i]->device_describe(dim, kind, f, odf) || kind != kind0 || f.size() != f0.size()) return false;
      double sc = 0;
      for (size_t k = 0; k < f.size(); k++) if (f0[k] != 0) { sc = f[k] / f0[k]; break; }
      for (size_t k = 0; k < f.size(); k++)
        if (std::fabs(f[k] - sc * f0[k]) > 1e-12 * (std::fabs(f[k]) + std::fabs(sc * f0[k])) + 1e-300) return false;   // not a multiple
      run += shares[i] / sum;                                  // bin_max (proposal_distribution.cc reset_bins)
      cum.push_back(i + 1 == proposals.size() ? 1.0 : run);
      scales.push_back(sc);
      odfs.push_back(odf);
    }
    return true;
  }
};

// ---- chain.hh ----------------------------------------------------------------------------------------------------------
class chain {  // chain.hh:34-141 (the part of the interface the driver uses)
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
  virtual double getMAPlpost() { return -1e200; }                    // chain.hh:116-117
  virtual state getMAPstate() { return getState(); }
  virtual std::string status() { return ""; }
};

class parallel_tempering_chains : public chain {  // chain.hh:214-330, chain.cc:1163-1571
  const int Ntemps, add_every_N;
  const double Tmax, swap_rate, dpriormin;
  ptm_engine* eng;
  const stateSpace* sp;
  int dim, nstep, hist_rows, hist_rungs = 0;   // hist_rungs: the coldest rungs whose saved states are kept (0: all)
  int W;   // independent replicas of the ladder run side by side (the reference runs its Nchain repeats one after the
           // other, ptmcmc.cc main loop): chain (rung i, replica w) sits at index i*W + w of every engine array
  std::vector<double> temps, X, llike, lpost;
  bool fresh, hist_fresh = false;
  // replica-exchange diagnostics of chain.cc:1346-1356,1448-1451,1495-1498 (directions / instances / ups / downs), kept
  // on the host by replaying each step's candidate log (replica 0); on after track_exchanges(true)
  bool tracking = false;
  std::vector<int> directions, instances;
  std::vector<long> ups, downs;
  std::vector<int32_t> log_pairs, log_acc;
  void replay_step() {   // the loop of chain.cc:1436-1537, bookkeeping part, in pick order
    const int ms = ptm_max_swaps_per_step(eng);
    log_pairs.resize((size_t)W * ms); log_acc.resize((size_t)W * ms);
    ptm_check(ptm_get_last_swaps(eng, log_pairs.data(), log_acc.data()), "replay_step");
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

  class rung_view : public chain {
    parallel_tempering_chains* p;
    int i, w;

   public:
    rung_view(parallel_tempering_chains* p, int i, int w = 0) : p(p), i(i), w(w) {}
    void step() override { std::cout << "rung_view::step: step the ladder, not a rung" << std::endl; exit(1); }
    size_t at() const { return (size_t)i * p->W + w; }
    state getState(int = -1, bool = false) override { p->refresh(); return state(p->sp, std::vector<double>(p->X.begin() + at() * p->dim, p->X.begin() + (at() + 1) * p->dim)); }
    double getLogPost(int = -1, bool = false) override { p->refresh(); return p->lpost[at()]; }
    double getLogLike(int = -1, bool = false) override { p->refresh(); return p->llike[at()]; }
    double invTemp() override { return p->cur_beta(i, w); }
    int getStep() override { return p->nstep; }
    double getMAPlpost() override { p->refresh_map(); return p->mlpost[at()]; }
    state getMAPstate() override { p->refresh_map(); return state(p->sp, std::vector<double>(p->mX.begin() + at() * p->dim, p->mX.begin() + (at() + 1) * p->dim)); }
  };
  std::vector<rung_view> views;

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
    if (ev_rate > 0) {
      betas.resize((size_t)W * Ntemps);
      ptm_check(ptm_get_invtemps(eng, betas.data()), "parallel_tempering_chains");
    }
    ptm_check(ptm_get_states(eng, X.data()), "parallel_tempering_chains");
    ptm_check(ptm_get_array(eng, PTM_ARR_LLIKE, llike.data()), "parallel_tempering_chains");
    ptm_check(ptm_get_array(eng, PTM_ARR_LPOST, lpost.data()), "parallel_tempering_chains");
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
  int history_rungs() const { return hist_rows > 0 ? (hist_rungs > 0 ? hist_rungs : Ntemps) : 0; }
  // Run `n` independent replicas of the ladder in one engine (call before initialize()).  Replica w uses the random
  // streams of walker w; every accessor below takes the replica as an optional last argument (default 0).  Multiples
  // of 64 fill whole wavefronts and take the fast kernels.
  void set_replicas(int n) { W = n < 1 ? 1 : n; }
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
  // Before or after initialize().  The posterior-ordering cut (lpost_cut >= 0) is not built.
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
  // PTchain.cp.  The reference's files hold its own generators and growing vectors; this one holds the engine's arrays
  // (states, llikes, MH_chain counters, step count = position of every random stream, exchange counters, evolving
  // temperatures, history ring, MAPs, the exchange diagnostics) -- the continued run is the uninterrupted run, bit for bit.
  void checkpoint(const std::string& path) {
    const std::string dir = path + "chain0-cp/";
    mkdir(dir.c_str(), 0777);
    std::ofstream os((dir + "PTchain.cp").c_str(), std::ios::binary);
    if (!os) { std::cout << "parallel_tempering_chains::checkpoint: cannot write " << dir << "PTchain.cp" << std::endl; exit(1); }
    const size_t N = (size_t)Ntemps * W, np = (size_t)W * (Ntemps > 1 ? Ntemps - 1 : 1);
    auto wr = [&](const void* ptr, size_t bytes) { os.write((const char*)ptr, (std::streamsize)bytes); };
    const char magic[8] = {'P', 'T', 'M', 'G', 'P', 'U', '0', '1'};
    int32_t hdr[8] = {Ntemps, W, dim, hist_rows + 65536 * history_rungs(), ev_rate > 0 ? 1 : 0, tracking ? 1 : 0, nstep, add_every_N};
    uint64_t estep = ptm_step_count(eng);
    wr(magic, 8); wr(hdr, sizeof hdr); wr(&estep, 8);
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
    if (hist_rows > 0) {
      const size_t n = (size_t)hist_rows * history_rungs() * W;
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
    if (!os) { std::cout << "parallel_tempering_chains::checkpoint: write failed" << std::endl; exit(1); }
  }
  // into a ladder set up exactly like the one that was saved (same sizes, seed, likelihood, prior, proposal, evolve_temps)
  void restart(const std::string& path) {
    const std::string fn = path + "chain0-cp/PTchain.cp";
    std::ifstream is(fn.c_str(), std::ios::binary);
    if (!is) { std::cout << "parallel_tempering_chains::restart: cannot read " << fn << std::endl; exit(1); }
    auto rd = [&](void* ptr, size_t bytes) { is.read((char*)ptr, (std::streamsize)bytes); };
    char magic[8]; int32_t hdr[8]; uint64_t estep;
    rd(magic, 8); rd(hdr, sizeof hdr); rd(&estep, 8);
    if (std::string(magic, 8) != "PTMGPU01" || hdr[0] != Ntemps || hdr[1] != W || hdr[2] != dim || hdr[3] != hist_rows + 65536 * history_rungs() ||
        hdr[4] != (ev_rate > 0 ? 1 : 0) || hdr[7] != add_every_N) {
      std::cout << "parallel_tempering_chains::restart: " << fn << " was written by a differently configured ladder" << std::endl;
      exit(1);
    }
    const size_t N = (size_t)Ntemps * W, np = (size_t)W * (Ntemps > 1 ? Ntemps - 1 : 1);
    std::vector<double> x(N * dim), ll(N);
    std::vector<int32_t> nt(N), na(N), ty(N);
    std::vector<int64_t> nh(N), st(np), sa(np);
    rd(x.data(), x.size() * 8); rd(ll.data(), N * 8); rd(nt.data(), N * 4); rd(na.data(), N * 4); rd(ty.data(), N * 4);
    rd(nh.data(), N * 8); rd(st.data(), np * 8); rd(sa.data(), np * 8);
    ptm_check(ptm_restore(eng, x.data(), ll.data(), nt.data(), na.data(), ty.data(), nh.data(), estep, st.data(), sa.data()), "restart");
    if (ev_rate > 0) {
      std::vector<double> b((size_t)W * Ntemps);
      rd(b.data(), b.size() * 8);
      ptm_check(ptm_set_invtemps(eng, b.data()), "restart");
    }
    if (hist_rows > 0) {
      const size_t n = (size_t)hist_rows * history_rungs() * W;
      std::vector<double> gx(n * dim), gl(n), gp(n), gb(n);
      std::vector<int32_t> gm(n * 4);
      rd(gx.data(), gx.size() * 8); rd(gl.data(), n * 8); rd(gp.data(), n * 8); rd(gm.data(), gm.size() * 4); rd(gb.data(), n * 8);
      ptm_check(ptm_set_history(eng, gx.data(), gl.data(), gp.data(), gm.data(), gb.data()), "restart");
    }
    {
      std::vector<double> mx(N * dim), mp(N), ml(N), mr(N);
      rd(mx.data(), mx.size() * 8); rd(mp.data(), N * 8); rd(ml.data(), N * 8); rd(mr.data(), N * 8);
      ptm_check(ptm_set_map(eng, mx.data(), mp.data(), ml.data(), mr.data()), "restart");
    }
    if (hdr[5]) {
      track_exchanges(true);
      std::vector<int64_t> u(Ntemps), d(Ntemps);
      rd(directions.data(), (size_t)Ntemps * 4); rd(instances.data(), (size_t)Ntemps * 4); rd(u.data(), (size_t)Ntemps * 8); rd(d.data(), (size_t)Ntemps * 8);
      ups.assign(u.begin(), u.end()); downs.assign(d.begin(), d.end());
    }
    if (!is) { std::cout << "parallel_tempering_chains::restart: " << fn << " is truncated" << std::endl; exit(1); }
    nstep = hdr[6];
    fresh = hist_fresh = map_fresh = false;
  }
  // chain.cc:1281-1365: n prior draws per rung; the device draws them (any prior type but the improper flat one)
  void initialize(bayes_likelihood* log_likelihood, const sampleable_probability_function* log_prior, int n = 1, uint64_t seed = 0x5EED0001ull,
                  const std::vector<double>* start_states = nullptr) {
    sp = log_prior->get_space();
    dim = log_prior->getDim();
    ptm_config cfg;
    cfg.struct_size = sizeof cfg;
    cfg.dim = dim; cfg.n_rungs = Ntemps; cfg.rung_begin = 0; cfg.rung_count = Ntemps; cfg.n_walkers = W; cfg.seed = seed;
    cfg.swap_rate = swap_rate; cfg.add_every_n = add_every_N; cfg.min_prior = dpriormin; cfg.device = -1; cfg.stream = nullptr;
    cfg.time_kernels = 0; cfg.swap_log_steps = 0; cfg.exchange_row_capacity = 0; cfg.history_rungs = history_rungs(); cfg.history_capacity = hist_rows; cfg.map_rungs = Ntemps;
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
    log_prior->describe(types, centers, halfwidths);
    ptm_check(ptm_set_prior(eng, types.data(), centers.data(), halfwidths.data()), "set_prior");
    if (!log_likelihood->describe_device_target(eng))
      ptm_check(ptm_set_target_callback(eng, &bayes_likelihood::batch_trampoline, log_likelihood), "set_target_callback");
    std::vector<double> beta(Ntemps);
    for (int i = 0; i < Ntemps; i++) beta[i] = 1 / temps[i];  // chain.cc:1340
    ptm_check(ptm_set_ladder(eng, beta.data()), "set_ladder");
    if (ev_rate > 0) ptm_check(ptm_set_evolve_temps(eng, ev_rate, ev_cut), "evolve_temps");
    X.assign((size_t)Ntemps * W * dim, 0.0); llike.assign((size_t)Ntemps * W, 0.0); lpost.assign((size_t)Ntemps * W, 0.0);
    if (start_states) {
      ptm_check(ptm_set_states(eng, start_states->data(), nullptr), "set_states");
    } else {
      int rc = ptm_init_from_prior(eng);
      if (rc == PTM_ERR_UNSUPPORTED) {
        std::cout << "parallel_tempering_chains::initialize: " << ptm_last_error() << "; pass start states" << std::endl;
        exit(1);
      }
      ptm_check(rc, "init_from_prior");
    }
    views.clear();
    for (int w = 0; w < W; w++)
      for (int i = 0; i < Ntemps; i++) views.push_back(rung_view(this, i, w));   // views[w*Ntemps + i]
    fresh = false;
  }
  // chain.cc:1367-1386: one clone per rung
  void set_proposal(proposal_distribution& proposal) {
    int kind = 0;
    double odf = 0;
    std::vector<double> f, all, odfs(Ntemps);
    for (int i = 0; i < Ntemps; i++) {
      props.push_back(proposal.clone());
      if (!props.back()->device_describe(dim, kind, f, odf)) {
        std::cout << "parallel_tempering_chains::set_proposal: " << proposal.show() << " has no device form (only gaussian_prop in this build)" << std::endl;
        exit(1);
      }
      all.insert(all.end(), f.begin(), f.end());
      odfs[i] = odf;
    }
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
  }
  // per-rung factors for proposals that differ by rung (what user_gaussian_prop's check_update achieves in the reference)
  void set_proposal_factors(int kind, const std::vector<double>& factors, const std::vector<double>& oneDfracs = std::vector<double>()) {
    ptm_check(ptm_set_proposals(eng, kind, factors.data(), oneDfracs.empty() ? nullptr : oneDfracs.data()), "set_proposals");
  }
  void step() override {
    ptm_check(ptm_step(eng, 1), "parallel_tempering_chains::step");
    nstep++;
    fresh = hist_fresh = map_fresh = false;
    if (tracking) replay_step();
  }
  void step(int n) {
    if (tracking) { for (int k = 0; k < n; k++) step(); return; }
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
  void dumpChain(int ichain, std::ostream& os, int Nburn = 0, int ievery = 1, int replica = 0) {
    if (hist_rows <= 0) { std::cout << "parallel_tempering_chains::dumpChain: call keep_history(rows) before initialize()" << std::endl; exit(1); }
    if (ichain >= history_rungs()) { std::cout << "parallel_tempering_chains::dumpChain: rung " << ichain << " keeps no history (keep_history(rows, " << history_rungs() << "))" << std::endl; exit(1); }
    const size_t HC = (size_t)history_rungs() * W, cap = hist_rows, at = (size_t)ichain * W + replica;
    if (!hist_fresh) {   // one read-back serves every rung / replica dumped at this step
      hx.resize(cap * HC * dim); hl.resize(cap * HC); hp.resize(cap * HC); hmeta.resize(cap * HC * 4); hnhist.resize((size_t)Ntemps * W); hb.resize(cap * HC);
      ptm_check(ptm_get_history(eng, hx.data(), hl.data(), hp.data(), hmeta.data()), "dumpChain");
      ptm_check(ptm_get_history_invtemps(eng, hb.data()), "dumpChain");   // the temperature each row was saved at
      ptm_check(ptm_get_array(eng, PTM_ARR_NHIST, hnhist.data()), "dumpChain");
      hist_fresh = true;
    }
    const std::vector<int32_t>& meta = hmeta;
    const int Ninit = 1, Nhist = (int)hnhist[at];
    os << "#Ninit=" << Ninit << ", Nburn=" << Nburn << "\n";
    os << "#eval: log(posterior) log(likelihood) acceptance_ratio prop_type: ";
    for (int i = 0; i < dim; i++) os << (sp ? sp->get_name(i) : std::string("[unnamed]")) << " ";
    os << std::endl;
    if (Nburn + Ninit < 0) Nburn = -Ninit;
    const double invtemp = cur_beta(ichain, replica);   // the chain's CURRENT temperature on every row (chain.cc:1131)
    for (int i = Nburn; i < Nhist; i += ievery) {
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

// ---- ptmcmc.hh: the driver, reduced to the run loop around cc->step() (ptmcmc.cc:530-679) ---------------------------------
class ptmcmc_sampler {
  std::map<std::string, std::string> opt;
  bayes_likelihood* chain_llike;
  const sampleable_probability_function* chain_prior;
  proposal_distribution* cprop;
  std::unique_ptr<parallel_tempering_chains> cc;

 public:
  ptmcmc_sampler() : chain_llike(nullptr), chain_prior(nullptr), cprop(nullptr) {
    // flag names and defaults of ptmcmc.cc:375-427 that shape this path
    opt["nsteps"] = "5000"; opt["save_every"] = "10"; opt["nevery"] = "1000"; opt["pt"] = "20"; opt["pt_swap_rate"] = "0.10";
    opt["pt_Tmax"] = "1e9"; opt["chain_dprior_min"] = "-30"; opt["seed"] = "-1"; opt["outname"] = "mcmc_output";
    opt["nskip"] = "10"; opt["pt_dump_n"] = "1"; opt["nchains"] = "1";
    opt["pt_evolve_rate"] = "0.01"; opt["pt_evolve_lpost_cut"] = "-1";   // ptmcmc.cc:389-390: the ladder evolves by default
    opt["checkp_at_step"] = "-1"; opt["restart_dir"] = "";                // ptmcmc.cc:377-379
  }
  // ptmcmc_sampler::checkpoint / restart (ptmcmc.cc:306-338): <path>/step_<istep>-cp/ptmcmc.cp + the ladder's own file
  void checkpoint(const std::string& path, int istep) {
    std::ostringstream ss;
    ss << path << "/step_" << istep << "-cp/";
    const std::string dir = ss.str();
    std::cout << "Writing checkpoint files to dir:" << dir << std::endl;
    mkdir(dir.c_str(), 0777);
    std::ofstream os((dir + "ptmcmc.cp").c_str());
    os << istep << std::endl;
    cc->checkpoint(dir);
  }
  int restart(const std::string& path) {
    std::cout << "Restarting from checkpoint files in dir:" << path << std::endl;
    std::ifstream is((path + "/ptmcmc.cp").c_str());
    int istep = -1;
    is >> istep;
    if (!is || istep < 0) { std::cout << "ptmcmc_sampler::restart: cannot read " << path << "/ptmcmc.cp" << std::endl; exit(1); }
    cc->restart(path + "/");
    return istep;
  }
  void set(const std::string& name, const std::string& value) { opt[name] = value; }
  bool parse(int argc, char* argv[]) {  // --name=value / --name (options.hh semantics)
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      if (a.compare(0, 2, "--")) return false;
      size_t eq = a.find('=');
      opt[a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2)] = eq == std::string::npos ? "true" : a.substr(eq + 1);
    }
    return true;
  }
  double num(const std::string& n) const { return atof(opt.at(n).c_str()); }
  void setup(bayes_likelihood& llike) { chain_llike = &llike; chain_prior = llike.getObjectPrior().get(); }
  void select_proposal(proposal_distribution& p) { cprop = &p; }
  int initialize() {
    if (!chain_llike || !cprop) { std::cout << "ptmcmc_sampler::initialize.  Must call setup() and set proposal before initialization!" << std::endl; exit(1); }
    cc.reset(new parallel_tempering_chains((int)num("pt"), num("pt_Tmax"), num("pt_swap_rate"), (int)num("save_every"), false, false, num("chain_dprior_min")));
    // the chain files are written from the device's history ring, every "nevery" steps: it must hold what one such
    // interval saves (up to two add_state calls per step, every save_every-th saved)
    {
      int dump_n = (int)num("pt_dump_n");
      if (dump_n > (int)num("pt") || dump_n <= 0) dump_n = (int)num("pt");   // ptmcmc.cc:458
      cc->keep_history(2 + 2 * (int)num("nevery") / std::max(1, (int)num("save_every")), dump_n);
    }
    cc->set_replicas((int)num("nchains"));   // the reference's Nchain repeats, all at once
    if (num("pt_evolve_rate") > 0) cc->evolve_temps(num("pt_evolve_rate"), num("pt_evolve_lpost_cut"));   // ptmcmc.cc:512
    uint64_t seed = num("seed") >= 0 ? (uint64_t)(num("seed") * 4294967296.0) : 0x5EED0001ull;
    cc->initialize(chain_llike, chain_prior, 1, seed);
    cc->set_proposal(*cprop);
    return 0;
  }
  // ptmcmc_sampler::run (ptmcmc.cc:530-679): step; every "nevery" steps append what the coldest pt_dump_n chains saved
  // since the last report to <base>_t<k>.dat (k = 0 the coldest), as dumpChain writes it
  int run(const std::string& base, int ic = 0) {
    (void)ic;
    const int Nstep = (int)num("nsteps"), Nevery = std::max(1, (int)num("nevery")), Nskip = std::max(1, (int)num("nskip"));
    int dump_n = (int)num("pt_dump_n");
    if (dump_n > cc->multiplicity() || dump_n <= 0) dump_n = cc->multiplicity();   // ptmcmc.cc:458
    const int nrep = cc->replicas();
    const int checkp_at_step = (int)num("checkp_at_step");
    const std::string restart_dir = opt.at("restart_dir");
    const bool restarting = !restart_dir.empty();
    std::vector<std::unique_ptr<std::ofstream> > out;
    for (int w = 0; w < nrep; w++)
      for (int ich = 0; ich < dump_n; ich++) {   // ptmcmc.cc:547-554; replica w > 0: <base>_c<w>_t<ich>.dat
        std::ostringstream ss;
        ss << base;
        if (w > 0) ss << "_c" << w;
        ss << "_t" << ich << ".dat";
        // a restarted run goes on writing where the first part stopped (ptmcmc.cc:538)
        out.emplace_back(new std::ofstream(ss.str().c_str(), restarting ? std::ios::out | std::ios::app : std::ios::out));
        out.back()->precision(13);
      }
    int istep0 = 0;
    if (restarting) istep0 = restart(restart_dir);   // ptmcmc.cc:564
    for (int istep = istep0; istep <= Nstep; istep++) {   // ptmcmc.cc:565,599-607
      if (istep == checkp_at_step) {   // ptmcmc.cc:567,593-596: write the checkpoint and stop
        std::cout << "Checkpointing triggered." << std::endl;
        checkpoint(".", istep);
        return 0;
      }
      cc->step();
      if (0 == istep % Nevery)
        for (int w = 0; w < nrep; w++)
          for (int ich = 0; ich < dump_n; ich++) cc->dumpChain(ich, *out[(size_t)w * dump_n + ich], istep - Nevery + 1, Nskip, w);
    }
    for (size_t k = 0; k < out.size(); k++) *out[k] << "\n" << std::endl;   // ptmcmc.cc:665
    return 0;
  }
  parallel_tempering_chains* chains() { return cc.get(); }
};

}  // namespace ptmgpu
#endif
