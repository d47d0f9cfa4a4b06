#!/bin/bash
# the torch-free C host's first whitening call on a fresh box, with the library's own read-ahead of the solver files
O=gpurun_out/r02u; mkdir -p $O
python - <<'PY'
import numpy as np
rng = np.random.default_rng(9)
open("/tmp/edges.tsv", "w").write("\n".join(" ".join(f"n{int(v)}" for v in rng.integers(0, 60, rng.integers(2, 6))) for _ in range(400)) + "\n")
PY
( time timeout 170 examples/embed_file --whiten "complex::reflexive::n" 8 4 /tmp/o.tsv /tmp/edges.tsv ) > $O/chost.log 2>&1; tail -6 $O/chost.log; wc -l /tmp/o.tsv
( time timeout 60 examples/embed_file --whiten "complex::reflexive::n" 8 4 /tmp/o2.tsv /tmp/edges.tsv ) > $O/chost2.log 2>&1; tail -4 $O/chost2.log
