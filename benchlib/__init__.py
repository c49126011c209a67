"""Measurement plumbing of bench.py: importable and unit-tested on CPU.  bench.py keeps the
argument parsing, the timed region, the CPU-baseline legs (the only code that may touch oracle/) and the print."""
