#!/bin/bash
# A/B of compile-time variants of the cell pass (run on the GPU box): rebuilds libmyfm_hip.so with extra hipcc flags, profiles config 5
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  touch myfm_amd/csrc/mfm_cell.hpp
  MYFM_AMD_HIPCC_FLAGS="$v" python -c "
import importlib.util
spec=importlib.util.spec_from_file_location('b','myfm_amd/_build.py');b=importlib.util.module_from_spec(spec);spec.loader.exec_module(b);b.build_all()" > /dev/null 2>&1
  bash scripts/prof_cfg.sh var_$tag --config 5 --scale 1.0 --steps 2 --warmup 1 > /dev/null 2>&1
  echo "== $v"; grep "k_cell_pass" gpurun_out/prof_var_$tag.txt | cut -c1-110
done
