# the sampler's default configuration over the population size: where the persistent ladder kernel ends (whole waves per rung) the
# lanes kernel takes over (PTM_FORCE_VALU=1: the general kernel, a lane per chain)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for W in 16 48 64 256 1024; do python tools/de_probe.py 12 64 $W 1000; done
for W in 64 256 1024; do PTM_FORCE_VALU=1 python tools/de_probe.py 12 64 $W 1000; done
for W in 16 64 512; do python tools/de_probe.py 32 128 $W 600; done
for W in 64 512; do PTM_FORCE_VALU=1 python tools/de_probe.py 32 128 $W 600; done
python tools/de_probe.py 6 20 1 4000
python tools/de_probe.py 6 20 64 2000
