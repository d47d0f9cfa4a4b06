"""cleora_amd — MI355X-native implementation of pycleora's Markov-propagation hot path.

  cleora_amd.pycleora.SparseMatrix   drop-in for pycleora.pycleora.SparseMatrix (src/lib.rs)
  cleora_amd.embed                   device-resident embed()/whitening loop (pycleora/__init__.py:51-164)
  cleora_amd.sharded                 row-partitioned multi-GPU propagation (one process per GPU)
  cleora_amd._hip                    ctypes binding of libcleora_hip.so (include/cleora_hip.h)
"""
__version__ = "0.1.0"
