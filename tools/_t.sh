cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -5
