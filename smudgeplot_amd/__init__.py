"""smudgeplot_amd -- MI355X-native engine for `smudgeplot hetmers` (FastK .ktab -> .smu).

Only the hot path of KamilSJaron/smudgeplot lives here:
  csrc/    HIP kernels + C ABI (libsmg_hetmers.so) and the plain-C drop-in `hetmers` executable
  engine   ctypes binding of include/smg_hetmers.h
  cli      mirror of `smudgeplot hetmers` argument handling (src/smudgeplot/cli.py:140-174,348-366)
  sharded  one-process-per-GPU driver (prefix shards, RCCL via torch.distributed)
  ktab     FastK table reader/writer, synth: synthetic conditioned tables (tests / bench)
"""

__version__ = "0.1.0"
