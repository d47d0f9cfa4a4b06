#!/bin/bash
# Round 3, GPU call 1: the new tests, the whole -m gpu suite, and A/B probes of the new kernels.
set -u
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
python -c 'import oracle; oracle.build()' > $O/oracle_build.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_whiten.py tests/test_gpu_spmm.py tests/test_gpu_variants.py -m gpu -q -x --no-header -p no:cacheprovider ) > $O/pytest_new.log 2>&1
tail -15 $O/pytest_new.log
( time timeout 900 python -m pytest tests/test_gpu_parity_at_scale.py -m gpu -q --no-header -p no:cacheprovider ) > $O/pytest_scale.log 2>&1
tail -15 $O/pytest_scale.log
for v in default "CLEORA_PROJECT=f32 CLEORA_GRAM=f64"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = default ]; then timeout 300 python scripts/r03_probe.py kernels > $O/kernels_$tag.json 2> $O/kernels_$tag.err
  else env $v timeout 300 python scripts/r03_probe.py kernels > $O/kernels_$tag.json 2> $O/kernels_$tag.err; fi
  cat $O/kernels_$tag.json; tail -3 $O/kernels_$tag.err
done
for v in default "CLEORA_SPMM_WAITS=compiler" "CLEORA_PROJECT=f32 CLEORA_GRAM=f64"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = default ]; then timeout 400 python scripts/r03_probe.py loop > $O/loop_$tag.json 2> $O/loop_$tag.err
  else env $v timeout 400 python scripts/r03_probe.py loop > $O/loop_$tag.json 2> $O/loop_$tag.err; fi
  cat $O/loop_$tag.json; tail -3 $O/loop_$tag.err
done
( time timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --deselect tests/test_gpu_parity_at_scale.py ) > $O/pytest_all.log 2>&1
tail -25 $O/pytest_all.log
