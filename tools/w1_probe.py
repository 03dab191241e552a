import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem
pr = GaussianProblem(32, 1024, 1e9)
e = E.Engine(32, 1024, 1, add_every_n=100)
pr.configure(e, E.PROP_LOWER)
e.init_from_prior()
e.step(200); e.sync()
e.timer_start(); e.step(500); ms = e.timer_stop() / 500
print("W=1 step %.4f ms" % ms)
