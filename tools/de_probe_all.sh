cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "12 64 1" "17 128 1" "32 128 1" "32 1024 1" "12 64 8"; do
  python tools/de_probe.py $cfg
  PTM_LADDER=0 python tools/de_probe.py $cfg
  PTM_LADDER=0 PTM_FORCE_VALU=1 python tools/de_probe.py $cfg 1000
done
