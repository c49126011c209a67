"""starkware.cairo.bootloaders.program_hash_test_utils names, served by starkperp.program_hash."""
from starkperp.program_hash import program_hash_test_main, run_generate_hash_test  # noqa: F401
