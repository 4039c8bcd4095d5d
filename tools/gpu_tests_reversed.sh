#!/bin/bash
# The GPU suite with the test files in reverse order (shared workspaces: order dependence).
set -e
cd $GRAFT_REPO_ROOT
FILES=$(ls tests/test_*.py | sort -r | tr '\n' ' ')
timeout -k 10 900 python -u -m pytest $FILES -x -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | tail -5
