"""starkware.crypto.signature.math_utils names, served by starkperp.math_utils."""
from starkperp.math_utils import (  # noqa: F401
    ECPoint, div_mod, ec_add, ec_double, ec_mult, ec_neg, is_quad_residue, pi_as_string, sqrt_mod,
)
