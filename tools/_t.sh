cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -5
for ns in 28 10000; do echo "Ns=$ns: $(PROF_NS=$ns python tools/prof_adam.py 2>&1 | tail -1)"; done
