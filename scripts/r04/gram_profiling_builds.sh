set -u
R=$(pwd); O=$R/gpurun_out/r04t; mkdir -p $O; export TMPDIR=/tmp
for lib in - scripts/r04/probe_libs/libcleora_hip_p3.so scripts/r04/probe_libs/libcleora_hip_p6.so; do
  timeout 100 python scripts/r04/gram_probe.py $lib 2>>$O/err.log | tee -a $O/profiling_builds.jsonl
done
timeout 180 python scripts/r04/kernel_probe.py 10000000 256 2>>$O/err.log | cut -c1-330
timeout 180 python scripts/r04/kernel_probe.py 1500007 256 2>>$O/err.log | cut -c1-330
