"""starkware.cairo.bootloaders.hash_program names (cairo-lang, imported at program_hash_test_utils.py:3)."""
from starkperp.program_hash import compute_program_hash_chain  # noqa: F401
