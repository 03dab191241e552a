# small problems (up to 8 dimensions): the fused small-ladder kernel (rungs x padded dimensions <= 256) against the persistent ladder
# kernel's builds for 4 and 8 padded dimensions (PTM_FUSED=0), and ladders beyond the fused kernel's reach against two launches (PTM_LADDER=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "6 20 1" "6 20 16" "3 8 64" "2 8 64"; do
  python tools/de_probe.py $cfg 4000
  PTM_FUSED=0 python tools/de_probe.py $cfg 4000
done
for cfg in "6 64 1" "6 128 2" "3 200 1"; do
  python tools/de_probe.py $cfg 3000
  PTM_LADDER=0 python tools/de_probe.py $cfg 3000
done
