"""Randomised GPU-vs-oracle parity stress (run by hand on the GPU box: python tests/stress_gpu.py [seed] [families]).
Random families over group sizes, lengths, indel rates/lengths, protein/DNA, ls 1/3, tgapf 1/0.5, weighted or not;
every division of every family is aligned by the product and compared bit for bit with the CPU oracle.  Kernel paths
can be forced with the G2G_* environment variables listed in DESIGN.md section 4."""
