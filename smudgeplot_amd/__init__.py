"""smudgeplot_amd -- MI355X-native engine for `smudgeplot hetmers` (FastK .ktab -> .smu).

Only the hot path of KamilSJaron/smudgeplot lives here:
  csrc/    HIP kernels + C ABI (libsmg_hetmers.so) and the plain-C drop-in `hetmers` executable
  engine   ctypes binding of include/smg_hetmers.h
  cli      mirror of the `smudgeplot hetmers` / `smudgeplot extract` tasks (src/smudgeplot/cli.py:140-174,
           210-232, 348-382): `python -m smudgeplot_amd hetmers ...`
  sharded  one-process-per-GPU driver (prefix shards, RCCL via torch.distributed)
  ktab     FastK table reader/writer, synth: synthetic conditioned tables (tests / bench)
"""

__version__ = "0.1.0"
