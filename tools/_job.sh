cd $GRAFT_REPO_ROOT
( time python bench.py ) 2>&1 | tail -5 | cut -c1-1800
