cd $GRAFT_REPO_ROOT
for r in 0 1 5 7 13; do echo "ROT=$r"; RDMI_ROT=$r python scripts/gpu_stamps.py 2>&1 | grep -E "total cycles|CONV rows=4 |CONV rows=16 |CONV rows=81 "; done
