cd $GRAFT_REPO_ROOT
for r in 0 1 2 3; do echo "UDBG=$r"; RDMI_UDBG=$r python scripts/gpu_stamps.py 2>&1 | grep -E "total cycles|^ *(4|9|44|67|101) |^GN|^GATHER"; done
