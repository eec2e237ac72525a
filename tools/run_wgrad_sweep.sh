#!/bin/bash
# one process per setting (the switches are read once)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python3 tools/sweep_linear_wgrad.py --torch 2>&1 | tail -1
python3 tools/sweep_linear_wgrad.py 2>&1 | tail -1
for tile in 64 128; do for blocks in 512 1024 1536 2048; do for rows in 64 256 512; do
  NRX_WGRAD_TILE=$tile NRX_WGRAD_BLOCKS=$blocks NRX_WGRAD_MIN_ROWS=$rows python3 tools/sweep_linear_wgrad.py 2>&1 | tail -1
done; done; done
