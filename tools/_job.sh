#!/bin/bash
# scratch: repeat the GPU suite to look for flakiness
cd /root/repo
for i in 1 2 3 4 5; do
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -1
done
