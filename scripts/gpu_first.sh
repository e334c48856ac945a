set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
