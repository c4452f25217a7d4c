#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | tr '\n' ' '; echo
done
