#!/bin/bash
# Dev: every dense shape of a Performer layer on the four-wave ring mainloop at each instantiated tile height (SA_PP_DBG bits 24-27 force one), next to the
# two-buffer kernels and hipBLASLt (tools/bench_dense_tiles.py)
for h in 0 1 2 3 4; do
  echo "== forced height index $h (0 = the dispatcher's choice)"
  SA_PP_DBG=$((h << 24)) python tools/bench_dense_tiles.py ${1:-8400} 2>&1 | grep -v amdgpu.ids
done
