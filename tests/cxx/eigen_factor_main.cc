// Prints the proposal factor the facade's gaussian_prop(covariance) hands to the engine (ptmcmc_gpu.hh: V * diag(sqrt(lambda)),
// its own Jacobi eigen-decomposition) for a covariance read from stdin: "D" then D*D numbers; one row-major line out.
// Checked by tests/test_cxx_facade.py against the reference's gaussian_prop(covar) transform (tests/golden/eigen.json.gz).
#include <cstdio>
#include <iostream>
#include <vector>

#include "ptmcmc_gpu.hh"

int main() {
  int D;
  if (!(std::cin >> D)) return 2;
  std::vector<double> cov((size_t)D * D);
  for (auto& v : cov) std::cin >> v;
  ptmgpu::gaussian_prop gp(cov, D);
  int kind = -1;
  double odf = -1;
  std::vector<double> f;
  if (!gp.device_describe(D, kind, f, odf) || kind != PTM_PROP_DENSE) return 3;
  for (size_t i = 0; i < f.size(); i++) printf("%.17g%c", f[i], i + 1 == f.size() ? '\n' : ' ');
  return 0;
}
