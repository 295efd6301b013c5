#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for flags in "-DTPG_NO_T3 -DTPG_NO_T4 -DTPG_NO_T5" "-DTPG_NO_T4 -DTPG_NO_T5" "-DTPG_NO_T3 -DTPG_NO_T5" "-DTPG_NO_T3 -DTPG_NO_T4" ""; do
  touch orthogonalsphericalshellgrids.jl_amd/csrc/tpg_grid.hip
  make -C orthogonalsphericalshellgrids.jl_amd/csrc GRID_FLAGS="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
  echo "[$flags]"; bash tools/pmc_cells.sh ab 2>&1 | tail -1
done
